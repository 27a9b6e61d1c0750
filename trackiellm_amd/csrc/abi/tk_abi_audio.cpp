/*
 * tk_abi_audio.cpp — tk_asr_whisper_* and tk_vad_silero_* on the HIP audio engine; host-side buffering and state
 * machines restated from src/audio/tk_asr_whisper.c:282-344 and src/sensors/tk_vad_silero.c:283-390,488-600.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <memory>
#include <string>
#include <vector>

#include "../audio/tk_audio_engine.h"
#include "../audio/tk_vad_graph.h"
#include "../audio/tk_whisper_ggml.h"
#include "tk/tk_audio.h"
#include "tk/tk_mi355x_ext.h"

#define TK_ASR_MAX_BUFFER (16000 * 30) /* MAX_AUDIO_BUFFER_SIZE: 30 s at 16 kHz */

static tk_error_code_t afail(tk_error_code_t code, const std::string& why) {
    tk_error_set_detail("%s", why.c_str());
    return code;
}

static uint64_t seed_of(const std::string& p, uint64_t dflt) {
    size_t k = p.find("seed=");
    return k == std::string::npos ? dflt : strtoull(p.c_str() + k + 5, nullptr, 10);
}

#include <sys/stat.h>
/* a registry key names a FILE, not a path: a path rewritten while an older handle is alive (size, modification time or inode differ) is another model */
static std::string file_identity(const std::string& path) {
    struct stat st;
    if (path.compare(0, 12, "synthetic://") == 0 || stat(path.c_str(), &st) != 0) return path;
    return path + "|" + std::to_string((long long)st.st_size) + "|" + std::to_string((long long)st.st_mtim.tv_sec) + "." + std::to_string((long long)st.st_mtim.tv_nsec) + "|" +
           std::to_string((unsigned long long)st.st_ino);
}

/* ---- per-model-file registry: every tk_asr_whisper_context_t opened on the same checkpoint / device shares ONE set of weights
 * and ONE batched engine.  The reference's call is one utterance per handle (src/audio/tk_asr_whisper.c:282-344); K cortex handles that each own
 * an engine at batch 1 ran K encoders side by side.  A call enqueues its utterance, a scheduler thread coalesces the waiting ones (same decoder
 * prompt and step count; shorter utterances are padded with the zeros the 30 s window would hold anyway) into one batched transcribe() — every
 * caller gets what it would get alone, the engine's arithmetic is per utterance (tests/test_audio_gpu.py: batch == singles). */
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#define TK_ASR_SHARED_MAX_BATCH 32

struct AsrReq {
    const int16_t* pcm = nullptr;
    int n_samples = 0, n_steps = 0;
    std::vector<int32_t> prompt;
    bool policy = false;         /* a temperature / log-probability decode (one utterance per job: its draws are keyed by the batch row) */
    bool ref = false;            /* the decode under the reference's whisper.cpp parameters (logit filters, timestamps): TkAsr::transcribe_ref */
    bool fast = false;           /* tk_mi355x_asr_set_fast_contraction: requests of one kind share a job */
    float temperature = 0.0f;
    uint64_t seed = 0;
    int32_t* tokens = nullptr;   /* [n_steps] */
    float* logprobs = nullptr;   /* [n_steps], policy / ref only */
    int32_t result_len = 0, status = 0; /* ref only */
    bool ok = false, done = false;
    std::string err;
};

struct SharedAsr {
    std::string key;
    TkWhisperModel model;
    std::vector<std::string> vocab; /* token id -> bytes, from a ggml checkpoint */
    std::vector<uint8_t> suppress;  /* [n_vocab] 1 = never sampled under the reference's parameters (build_suppress); empty = the vocabulary has no timestamp tokens */
    int32_t token_beg = 0, token_eot = 0;
    std::unique_ptr<TkAsr> eng;
    bool scheduled = false;         /* false: a private context (tk_mi355x_asr_create): calls go straight to the engine */
    std::mutex mu;
    std::condition_variable cv_req, cv_done;
    std::deque<AsrReq*> q;
    std::thread th;
    bool stop = false;
    int handles = 0;
    uint64_t n_batches = 0, n_utts = 0, widest = 0;

    ~SharedAsr() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_req.notify_all();
        if (th.joinable()) th.join();
    }
    std::mutex eng_mu; /* held while a job runs on `eng` and while `eng` is replaced */
    /* called when a handle joins (under the registry lock): the engine is sized for the handles there are — 1, 2, 4 ... MAX frames / utterances per
     * job — at CREATE time, so that nothing allocates or frees device memory while jobs and other streams' graph captures run */
    bool grow_for_handles(std::string* err) {
        int want = 1;
        while (want < handles && want < TK_ASR_SHARED_MAX_BATCH) want *= 2;
        std::lock_guard<std::mutex> lk(eng_mu);
        return ensure_engine(want, err);
    }
    bool ensure_engine(int want, std::string* err) {
        if (eng && eng->max_batch >= want) return true;
        std::unique_ptr<TkAsr> ne(new TkAsr());
        if (!ne->init(&model, want)) { *err = ne->error; return false; }
        eng.swap(ne);
        return true;
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_req.wait(lk, [&] { return stop || !q.empty(); });
            if (stop) break;
            if (handles > 1 && !q.front()->policy) { /* an encoder pass is ~10 ms: waiting a millisecond for the other handles' callers pays */
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
                while ((int)q.size() < (handles < TK_ASR_SHARED_MAX_BATCH ? handles : TK_ASR_SHARED_MAX_BATCH) && !stop)
                    if (cv_req.wait_until(lk, until) == std::cv_status::timeout) break;
                if (stop) break;
            }
            std::vector<AsrReq*> job;
            AsrReq* first = q.front();
            if (first->policy) { job.push_back(first); q.pop_front(); }
            else
                for (auto it = q.begin(); it != q.end() && (int)job.size() < TK_ASR_SHARED_MAX_BATCH;) {
                    AsrReq* r = *it;
                    if (!r->policy && r->ref == first->ref && r->fast == first->fast && r->n_steps == first->n_steps && r->prompt == first->prompt) { job.push_back(r); it = q.erase(it); }
                    else ++it;
                }
            int cap = 1;
            while (cap < (int)job.size() || (cap < handles && cap < TK_ASR_SHARED_MAX_BATCH)) cap *= 2;
            lk.unlock();
            std::string err;
            std::unique_lock<std::mutex> el(eng_mu);
            bool ok = ensure_engine(cap, &err); /* a no-op after grow_for_handles(): the engine already takes this many */
            if (ok) {
                eng->fast = first->fast;
                const int B = (int)job.size();
                int n_max = 0;
                for (AsrReq* r : job) n_max = r->n_samples > n_max ? r->n_samples : n_max;
                if (first->ref) {
                    /* B utterances (one when the decode draws: policy), each with its own length = its own seek_end */
                    std::vector<int16_t> pcm((size_t)B * (n_max > 0 ? n_max : 1), 0);
                    std::vector<int32_t> lens(B), toks((size_t)B * first->n_steps), rl(B), stt(B);
                    std::vector<float> lps((size_t)B * first->n_steps);
                    for (int b = 0; b < B; ++b) {
                        lens[b] = job[b]->n_samples;
                        if (job[b]->n_samples > 0) memcpy(&pcm[(size_t)b * n_max], job[b]->pcm, (size_t)job[b]->n_samples * sizeof(int16_t));
                    }
                    ok = eng->transcribe_ref(B, pcm.data(), n_max, lens.data(), first->prompt.data(), (int)first->prompt.size(), first->n_steps, first->temperature, first->seed,
                                             suppress.data(), token_beg, token_eot, toks.data(), lps.data(), rl.data(), stt.data());
                    for (int b = 0; ok && b < B; ++b) {
                        memcpy(job[b]->tokens, &toks[(size_t)b * first->n_steps], (size_t)first->n_steps * sizeof(int32_t));
                        if (job[b]->logprobs) memcpy(job[b]->logprobs, &lps[(size_t)b * first->n_steps], (size_t)first->n_steps * sizeof(float));
                        job[b]->result_len = rl[b];
                        job[b]->status = stt[b];
                    }
                } else if (first->policy) {
                    ok = eng->transcribe_policy(1, first->pcm, first->n_samples, first->prompt.data(), (int)first->prompt.size(), first->n_steps, first->temperature, first->seed,
                                                first->tokens, first->logprobs);
                } else if (B == 1) {
                    ok = eng->transcribe(1, first->pcm, first->n_samples, first->prompt.data(), (int)first->prompt.size(), first->n_steps, first->tokens, nullptr, nullptr, nullptr);
                } else {
                    std::vector<int16_t> pcm((size_t)B * (n_max > 0 ? n_max : 1), 0); /* zero padding = the silence the 30 s window holds behind a shorter utterance */
                    for (int b = 0; b < B; ++b)
                        if (job[b]->n_samples > 0) memcpy(&pcm[(size_t)b * n_max], job[b]->pcm, (size_t)job[b]->n_samples * sizeof(int16_t));
                    std::vector<int32_t> toks((size_t)B * first->n_steps);
                    ok = eng->transcribe(B, pcm.data(), n_max, first->prompt.data(), (int)first->prompt.size(), first->n_steps, toks.data(), nullptr, nullptr, nullptr);
                    if (ok)
                        for (int b = 0; b < B; ++b) memcpy(job[b]->tokens, &toks[(size_t)b * first->n_steps], (size_t)first->n_steps * sizeof(int32_t));
                }
                if (!ok) err = eng->error;
            }
            el.unlock();
            lk.lock();
            n_batches++;
            n_utts += job.size();
            if (job.size() > widest) widest = job.size();
            for (AsrReq* r : job) { r->ok = ok; if (!ok) r->err = err; r->done = true; }
            cv_done.notify_all();
        }
        for (AsrReq* r : q) { r->ok = false; r->err = "the ASR context was destroyed"; r->done = true; }
        q.clear();
        cv_done.notify_all();
    }
    void submit(AsrReq* r) {
        std::unique_lock<std::mutex> lk(mu);
        q.push_back(r);
        cv_req.notify_all();
        cv_done.wait(lk, [&] { return r->done; });
    }
};

static std::mutex g_asr_mu;
static std::map<std::string, std::weak_ptr<SharedAsr>> g_asr_registry;

struct tk_asr_whisper_context_s {
    std::shared_ptr<SharedAsr> sh; /* weights, vocabulary and the engine: shared by every context of the same file (or private: tk_mi355x_asr_create) */
    std::vector<int16_t> buffer;
    size_t buffer_size = 0;
    bool has_partial = false;
    std::string last_text, language;
    int decode_steps = 16;
    int32_t sot = 0, nots = 0, eot = 0;
    bool multilingual = false, translate = false;
    /* whisper.cpp's decoding policy as the reference's wrapper arms it (src/audio/tk_asr_whisper.c:126-138); off unless
     * tk_mi355x_asr_set_decode_policy switched it on (the default path is the plain greedy decode the parity tests pin) */
    bool forced_greedy = false; /* tk_mi355x_asr_set_reference_decode(ctx, 0): the round-1..5 decode — <|notimestamps|> prompt, bare arg max, decode_steps tokens */
    bool fast = false; /* tk_mi355x_asr_set_fast_contraction */
    bool policy_on = false;
    float temperature_inc = 0.2f, entropy_thold = 2.4f, logprob_thold = -1.0f;
    uint64_t policy_seed = 0;
    float last_temperature = 0.0f, last_avg_logprob = 0.0f;
    int last_attempts = 0;
    ~tk_asr_whisper_context_s() {
        if (sh) { std::lock_guard<std::mutex> lk(sh->mu); sh->handles--; }
    }
};

/* Whisper's language table in token order (<|en|> = sot + 1, <|zh|> = sot + 2, ...): the published tokenizer order, which
 * whisper.cpp's g_lang follows (the reference hands `language` to whisper.cpp, src/audio/tk_asr_whisper.c:252,386) */
static const char* const k_whisper_langs[] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi", "he", "uk", "el", "ms", "cs",
    "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk",
    "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu",
    "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"};

static int whisper_lang_id(const std::string& code) {
    for (int i = 0; i < (int)(sizeof k_whisper_langs / sizeof k_whisper_langs[0]); ++i)
        if (code == k_whisper_langs[i]) return i;
    return -1;
}

/* decoder prompt, as whisper.cpp's whisper_full builds it: English-only vocabularies start from <|startoftranscript|> alone;
 * multilingual ones add the language and task tokens; <|notimestamps|> closes both (the reference's wrapper concatenates segment
 * texts only, tk_asr_whisper.c:160-181).  Returns false for a language the vocabulary has no token for. */
static bool asr_prompt(const tk_asr_whisper_context_s* c, std::vector<int32_t>* out, std::string* why, bool timestamps = false) {
    out->clear();
    out->push_back(c->sot);
    if (c->multilingual) {
        const int n_lang = c->sh->model.hp.n_vocab - 51865 + 99; /* 99 languages at 51865 tokens, 100 (+ yue) at 51866 */
        int id = whisper_lang_id(c->language.empty() || c->language == "auto" ? std::string("en") : c->language); /* no language detection pass: "auto" decodes as English */
        if (id < 0 || id >= n_lang) { *why = "unknown language \"" + c->language + "\""; return false; }
        out->push_back(c->sot + 1 + id);
        out->push_back(c->sot + 1 + n_lang + (c->translate ? 0 : 1)); /* <|translate|>, <|transcribe|> follow the language block */
    }
    if (!timestamps) out->push_back(c->nots); /* the reference leaves no_timestamps at its default (off): its decodes start without this token */
    return true;
}

/* whisper.cpp's non_speech_tokens (whisper_process_logits, suppress_non_speech_tokens — the reference sets it, tk_asr_whisper.c:100): each string
 * and its " "-prefixed form is masked when the vocabulary holds it; " -" and " '" are masked too (hyphens and quotes only inside words) */
static const char* const k_non_speech_tokens[] = {
    "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
    "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]",
    "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac",
    "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};

/* the tokens whisper_process_logits masks at every step under the reference's parameters: <|notimestamps|>, <|startoftranscript|>, <|nospeech|>,
 * <|startoflm|> (tinydiarize off), <|translate|>, <|transcribe|>, <|startofprev|>, the language tokens, the non-speech tokens.  Only vocabularies
 * with whisper's special-token layout (>= 51864 tokens) have timestamp tokens: smaller (test) vocabularies leave the table empty and decode the
 * forced-greedy way.  Called under sh->mu or before the context is published. */
static void build_suppress(SharedAsr* sh, const tk_asr_whisper_context_s* c) {
    const int v = sh->model.hp.n_vocab;
    if (!sh->suppress.empty() || v < 51864) return;
    const int32_t nots = c->nots, beg = nots + 1;
    if (beg >= v) return;
    sh->suppress.assign((size_t)v, 0);
    auto mask = [&](int32_t id) { if (id >= 0 && id < v) sh->suppress[(size_t)id] = 1; };
    /* <|translate|> .. <|notimestamps|> are the six tokens below the first timestamp: translate, transcribe, startoflm, startofprev, nospeech, notimestamps */
    for (int32_t id = nots - 5; id <= nots; ++id) mask(id);
    mask(c->sot);
    for (int32_t id = c->sot + 1; id < nots - 5; ++id) mask(id); /* the language tokens */
    if (!sh->vocab.empty()) {
        std::map<std::string, int32_t> ids;
        for (size_t i = 0; i < sh->vocab.size() && (int)i < c->eot; ++i) ids.emplace(sh->vocab[i], (int32_t)i); /* text tokens only; first id of a string */
        for (const char* t : k_non_speech_tokens)
            for (const std::string& form : {std::string(t), std::string(" ") + t}) {
                auto it = ids.find(form);
                if (it != ids.end()) mask(it->second);
            }
        for (const char* t : {" -", " '"}) {
            auto it = ids.find(t);
            if (it != ids.end()) mask(it->second);
        }
    }
    sh->token_beg = beg;
    sh->token_eot = c->eot;
}

static bool have_gpu() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

static tk_error_code_t asr_load(SharedAsr* sh, const TkWhisperHP& hp, const std::string& path, uint64_t seed, int device) {
    const bool synthetic = path.empty() || path.compare(0, 12, "synthetic://") == 0;
    if (!synthetic && TkWhisperGgml::is_ggml(path.c_str())) {
        /* the reference's checkpoint format (whisper.cpp ggml .bin): geometry, filter bank, vocabulary and weights come from the file */
        TkWhisperGgml g;
        if (!g.open(path.c_str())) return afail(TK_ERROR_MODEL_LOAD_FAILED, g.error);
        if (!sh->model.init(g.hp, device)) return afail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
        if (!sh->model.load_ggml(g)) return afail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
        sh->vocab.swap(g.vocab);
    } else {
        if (!sh->model.init(hp, device)) return afail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
        if (synthetic) {
            if (!sh->model.fill_synthetic(seed)) return afail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
        } else if (!sh->model.load_file(path.c_str())) {
            return afail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
        }
    }
    return TK_SUCCESS;
}

/* shared = the reference's entry (tk_asr_whisper_create): find-or-load by (file, device), calls coalesced by the scheduler; otherwise a private
 * context whose engine takes `max_batch` utterances per direct call (tk_mi355x_asr_create: the bench's batched perception stream, the tests) */
static tk_error_code_t asr_new(tk_asr_whisper_context_t** out, const TkWhisperHP& hp, const std::string& path, uint64_t seed, int device, int max_batch, bool shared) {
    if (!have_gpu()) return afail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible (the MI355X path has no CPU fallback)");
    std::unique_ptr<tk_asr_whisper_context_s> c(new tk_asr_whisper_context_s());
    std::shared_ptr<SharedAsr> sh;
    std::string err;
    if (shared) {
        const std::string key = file_identity(path) + "|dev" + std::to_string(device);
        std::lock_guard<std::mutex> lk(g_asr_mu);
        sh = g_asr_registry[key].lock();
        if (!sh) {
            sh.reset(new SharedAsr());
            sh->key = key;
            sh->scheduled = true;
            tk_error_code_t rc = asr_load(sh.get(), hp, path, seed, device);
            if (rc != TK_SUCCESS) return rc;
            if (!sh->ensure_engine(1, &err)) return afail(TK_ERROR_GPU_MEMORY, err);
            SharedAsr* raw = sh.get();
            sh->th = std::thread([raw] { raw->run(); });
            g_asr_registry[key] = sh;
        }
    } else {
        sh.reset(new SharedAsr());
        tk_error_code_t rc = asr_load(sh.get(), hp, path, seed, device);
        if (rc != TK_SUCCESS) return rc;
        if (!sh->ensure_engine(max_batch, &err)) return afail(TK_ERROR_GPU_MEMORY, err);
    }
    { std::lock_guard<std::mutex> hl(sh->mu); sh->handles++; }
    c->sh = sh;
    if (shared && !sh->grow_for_handles(&err)) return afail(TK_ERROR_GPU_MEMORY, err); /* (c's destructor takes the handle back) */
    c->buffer.assign(TK_ASR_MAX_BUFFER, 0);
    const int v = sh->model.hp.n_vocab;
    const int ml = v >= 51865 ? 1 : 0; /* multilingual vocabularies have one more text token: the special ids move up by one */
    c->sot = 50257 + ml < v - 3 ? 50257 + ml : v - 3;
    c->nots = 50362 + ml < v - 1 ? 50362 + ml : v - 1;
    c->eot = 50256 + ml < v - 4 ? 50256 + ml : v - 4;
    c->multilingual = ml != 0;
    if (ml) { /* multilingual vocabulary (whisper.cpp's whisper_vocab): eot 50257, sot 50258, one token per language, then translate,
               * transcribe, startoflm, startofprev, nospeech, notimestamps */
        const int n_lang = v - 51865 + 99;
        c->eot = 50257; c->sot = 50258;
        c->nots = c->sot + 1 + n_lang + 5;
    }
    { std::lock_guard<std::mutex> hl(sh->mu); build_suppress(sh.get(), c.get()); }
    *out = c.release();
    return TK_SUCCESS;
}

/* one utterance under the reference's whisper.cpp parameters (TkAsr::transcribe_ref): tokens [n_steps], log-probabilities, the length of the text's
 * token run and the row's final status */
static tk_error_code_t asr_run_ref(tk_asr_whisper_context_s* c, const int16_t* pcm, int n_samples, const std::vector<int32_t>& prompt, int n_steps, float temperature,
                                   uint64_t seed, int32_t* tokens_out, float* logprobs_out, int32_t* result_len, int32_t* status) {
    SharedAsr* sh = c->sh.get();
    const TkWhisperHP& h = sh->model.hp;
    if (sh->suppress.empty()) return afail(TK_ERROR_INFERENCE_FAILED, "this vocabulary has no timestamp tokens: the reference-parameter decode does not apply");
    if (n_samples < 0 || n_samples > h.n_samples()) return afail(TK_ERROR_INFERENCE_FAILED, "audio longer than the model window");
    if (prompt.empty() || (int)prompt.size() + n_steps > h.n_text_ctx) return afail(TK_ERROR_INFERENCE_FAILED, "prompt + steps exceed the text context");
    if (sh->scheduled) {
        AsrReq r;
        r.pcm = pcm; r.n_samples = n_samples; r.n_steps = n_steps; r.prompt = prompt; r.ref = true; r.fast = c->fast; r.policy = temperature > 0.0f; r.temperature = temperature; r.seed = seed;
        r.tokens = tokens_out; r.logprobs = logprobs_out;
        sh->submit(&r);
        if (!r.ok) return afail(TK_ERROR_INFERENCE_FAILED, r.err);
        if (result_len) *result_len = r.result_len;
        if (status) *status = r.status;
        return TK_SUCCESS;
    }
    std::lock_guard<std::mutex> lk(sh->eng_mu);
    const int32_t len = n_samples;
    int32_t rl = 0, stt = 0;
    sh->eng->fast = c->fast;
    if (!sh->eng->transcribe_ref(1, pcm, n_samples, &len, prompt.data(), (int)prompt.size(), n_steps, temperature, seed, sh->suppress.data(), sh->token_beg, sh->token_eot,
                                 tokens_out, logprobs_out, &rl, &stt))
        return afail(TK_ERROR_INFERENCE_FAILED, sh->eng->error);
    if (result_len) *result_len = rl;
    if (status) *status = stt;
    return TK_SUCCESS;
}

/* one decode of `batch` utterances for this context: through the scheduler (shared contexts, one utterance per call) or straight on the engine */
static tk_error_code_t asr_run(tk_asr_whisper_context_s* c, int batch, const int16_t* pcm, int n_samples, const std::vector<int32_t>& prompt, int n_steps, bool policy,
                               float temperature, uint64_t seed, int32_t* tokens_out, float* logprobs_out, std::vector<float>* mel, std::vector<float>* enc,
                               std::vector<float>* lg) {
    SharedAsr* sh = c->sh.get();
    if (sh->scheduled && batch == 1 && !mel && !enc && !lg) {
        AsrReq r;
        r.pcm = pcm; r.n_samples = n_samples; r.n_steps = n_steps; r.prompt = prompt; r.fast = c->fast; r.policy = policy; r.temperature = temperature; r.seed = seed;
        r.tokens = tokens_out; r.logprobs = logprobs_out;
        const TkWhisperHP& h = sh->model.hp;
        if (n_samples < 0 || n_samples > h.n_samples()) return afail(TK_ERROR_INFERENCE_FAILED, "audio longer than the model window");
        if (prompt.empty() || (int)prompt.size() + n_steps > h.n_text_ctx) return afail(TK_ERROR_INFERENCE_FAILED, "prompt + steps exceed the text context");
        sh->submit(&r);
        return r.ok ? TK_SUCCESS : afail(TK_ERROR_INFERENCE_FAILED, r.err);
    }
    /* the test / bench entry points on a shared context (copies of the mel, batches): the engine itself, one caller at a time */
    std::lock_guard<std::mutex> lk(sh->eng_mu);
    std::string err;
    if (!sh->ensure_engine(batch, &err)) return afail(TK_ERROR_GPU_MEMORY, err);
    sh->eng->fast = c->fast;
    const bool ok = policy ? sh->eng->transcribe_policy(batch, pcm, n_samples, prompt.data(), (int)prompt.size(), n_steps, temperature, seed, tokens_out, logprobs_out)
                           : sh->eng->transcribe(batch, pcm, n_samples, prompt.data(), (int)prompt.size(), n_steps, tokens_out, mel, enc, lg);
    return ok ? TK_SUCCESS : afail(TK_ERROR_INFERENCE_FAILED, sh->eng->error);
}

extern "C" {

tk_error_code_t tk_asr_whisper_create(tk_asr_whisper_context_t** out_context, const tk_asr_whisper_config_t* config) {
    if (!out_context || !config || !config->model_path || !config->model_path->path_str) return TK_ERROR_INVALID_ARGUMENT;
    if (config->sample_rate != 16000) return afail(TK_ERROR_INVALID_ARGUMENT, "Whisper needs 16 kHz audio");
    const std::string path = config->model_path->path_str;
    tk_error_code_t rc = asr_new(out_context, tk_whisper_tiny_en(), path, seed_of(path, 6), tk_mi355x_get_default_device(), 1, true);
    if (rc == TK_SUCCESS && config->language) (*out_context)->language = config->language;
    if (rc == TK_SUCCESS) (*out_context)->translate = config->translate_to_en;
    /* TK_MI355X_ASR_POLICY=1: a reference host that cannot call tk_mi355x_asr_set_decode_policy gets whisper.cpp's temperature fallback armed
     * with the numbers process_with_whisper sets (tk_asr_whisper.c:126-138) from create on */
    const char* pol = getenv("TK_MI355X_ASR_POLICY");
    if (rc == TK_SUCCESS && pol && pol[0] == '1') (*out_context)->policy_on = true;
    return rc;
}

tk_error_code_t tk_mi355x_asr_create(tk_asr_whisper_context_t** out, const tk_mi355x_whisper_hparams_t* hp, uint64_t seed, int device, int max_batch) {
    if (!out || !hp) return TK_ERROR_INVALID_ARGUMENT;
    TkWhisperHP h{hp->n_mels, hp->n_audio_ctx, hp->n_audio_state, hp->n_audio_head, hp->n_audio_layer,
                  hp->n_text_ctx, hp->n_text_state, hp->n_text_head, hp->n_text_layer, hp->n_vocab};
    return asr_new(out, h, "", seed, device, max_batch, false);
}

tk_error_code_t tk_mi355x_asr_get_hparams(tk_asr_whisper_context_t* ctx, tk_mi355x_whisper_hparams_t* out) {
    if (!ctx || !out) return TK_ERROR_INVALID_ARGUMENT;
    const TkWhisperHP& h = ctx->sh->model.hp;
    *out = tk_mi355x_whisper_hparams_t{h.n_mels, h.n_audio_ctx, h.n_audio_state, h.n_audio_head, h.n_audio_layer,
                                       h.n_text_ctx, h.n_text_state, h.n_text_head, h.n_text_layer, h.n_vocab};
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_whisper_ggml_probe(const char* path, tk_mi355x_whisper_hparams_t* out, int32_t* n_tokens, int32_t* n_tensors) {
    if (!path || !out) return TK_ERROR_INVALID_ARGUMENT;
    TkWhisperGgml g;
    if (!g.open(path)) return afail(TK_ERROR_MODEL_LOAD_FAILED, g.error);
    const TkWhisperHP& h = g.hp;
    *out = tk_mi355x_whisper_hparams_t{h.n_mels, h.n_audio_ctx, h.n_audio_state, h.n_audio_head, h.n_audio_layer,
                                       h.n_text_ctx, h.n_text_state, h.n_text_head, h.n_text_layer, h.n_vocab};
    if (n_tokens) *n_tokens = (int32_t)g.vocab.size();
    if (n_tensors) *n_tensors = (int32_t)g.tensors.size();
    /* every tensor the graph needs must be there with the right element count */
    const TkWhManifest man = tk_whisper_manifest(h);
    for (int i = 0; i < (int)man.t.size(); ++i) {
        if (i == man.hann || i == man.dft || i == man.melw) continue;
        bool ok = false;
        for (const auto& t : g.tensors)
            if (t.name == man.t[i].name) { ok = t.count == man.t[i].rows * man.t[i].cols; break; }
        if (!ok) return afail(TK_ERROR_MODEL_VERIFICATION_FAILED, "tensor " + man.t[i].name + " is missing or has the wrong size");
    }
    return TK_SUCCESS;
}

void tk_asr_whisper_destroy(tk_asr_whisper_context_t** context) {
    if (!context || !*context) return;
    delete *context;
    *context = nullptr;
}

void tk_mi355x_asr_share_stats(const tk_asr_whisper_context_t* c, uint64_t* handles, uint64_t* batches, uint64_t* utterances, uint64_t* widest) {
    if (!c || !c->sh) return;
    std::lock_guard<std::mutex> lk(c->sh->mu);
    if (handles) *handles = (uint64_t)c->sh->handles;
    if (batches) *batches = c->sh->n_batches;
    if (utterances) *utterances = c->sh->n_utts;
    if (widest) *widest = c->sh->widest;
}

tk_error_code_t tk_mi355x_asr_set_fast_contraction(tk_asr_whisper_context_t* c, int on) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    c->fast = on != 0;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_asr_set_reference_decode(tk_asr_whisper_context_t* c, int enable) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    c->forced_greedy = enable == 0;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_asr_transcribe_ref(tk_asr_whisper_context_t* c, const int16_t* pcm, int n_samples, int n_steps, float temperature, uint64_t seed, int32_t* tokens_out,
                                             float* logprobs_out, int32_t* result_len, int32_t* status) {
    if (!c || (!pcm && n_samples > 0) || !tokens_out || n_steps <= 0) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<int32_t> prompt;
    std::string why;
    if (!asr_prompt(c, &prompt, &why, true)) return afail(TK_ERROR_INFERENCE_FAILED, why);
    return asr_run_ref(c, pcm, n_samples, prompt, n_steps, temperature, seed, tokens_out, logprobs_out, result_len, status);
}

int32_t tk_mi355x_asr_suppress_table(tk_asr_whisper_context_t* c, uint8_t* out, int32_t cap, int32_t* token_beg, int32_t* token_eot) {
    if (!c) return -1;
    std::lock_guard<std::mutex> lk(c->sh->mu);
    const int32_t n = (int32_t)c->sh->suppress.size();
    if (out) for (int32_t i = 0; i < n && i < cap; ++i) out[i] = c->sh->suppress[(size_t)i];
    if (token_beg) *token_beg = c->sh->token_beg;
    if (token_eot) *token_eot = c->sh->token_eot;
    return n;
}

void tk_mi355x_asr_set_decode_steps(tk_asr_whisper_context_t* ctx, int n_steps) {
    if (ctx && n_steps > 0) ctx->decode_steps = n_steps;
}

tk_error_code_t tk_mi355x_asr_transcribe_tokens(tk_asr_whisper_context_t* c, int batch, const int16_t* pcm, int n_samples, int n_steps,
                                                int32_t* tokens_out, float* mel_out, float* enc_out, float* logits_out) {
    if (!c || (!pcm && n_samples > 0) || !tokens_out || n_steps <= 0) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<int32_t> prompt;
    std::string why;
    if (!asr_prompt(c, &prompt, &why)) return afail(TK_ERROR_INFERENCE_FAILED, why); /* whisper_full fails the same way at decode time */
    std::vector<float> mel, enc, lg;
    tk_error_code_t rc = asr_run(c, batch, pcm, n_samples, prompt, n_steps, false, 0.0f, 0, tokens_out, nullptr, mel_out ? &mel : nullptr, enc_out ? &enc : nullptr,
                                 logits_out ? &lg : nullptr);
    if (rc != TK_SUCCESS) return rc;
    if (mel_out) memcpy(mel_out, mel.data(), mel.size() * 4);
    if (enc_out) memcpy(enc_out, enc.data(), enc.size() * 4);
    if (logits_out) memcpy(logits_out, lg.data(), lg.size() * 4);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_asr_transcribe_policy(tk_asr_whisper_context_t* c, int batch, const int16_t* pcm, int n_samples, int n_steps, float temperature,
                                                uint64_t seed, int32_t* tokens_out, float* logprobs_out) {
    if (!c || (!pcm && n_samples > 0) || !tokens_out || n_steps <= 0) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<int32_t> prompt;
    std::string why;
    if (!asr_prompt(c, &prompt, &why)) return afail(TK_ERROR_INFERENCE_FAILED, why);
    return asr_run(c, batch, pcm, n_samples, prompt, n_steps, true, temperature, seed, tokens_out, logprobs_out, nullptr, nullptr, nullptr);
}

tk_error_code_t tk_mi355x_asr_set_decode_policy(tk_asr_whisper_context_t* c, int enable, float temperature_inc, float entropy_thold, float logprob_thold, uint64_t seed) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    c->policy_on = enable != 0;
    c->temperature_inc = temperature_inc; c->entropy_thold = entropy_thold; c->logprob_thold = logprob_thold; c->policy_seed = seed;
    return TK_SUCCESS;
}

void tk_mi355x_asr_last_decode(const tk_asr_whisper_context_t* c, float* temperature, float* avg_logprob, int32_t* attempts) {
    if (!c) return;
    if (temperature) *temperature = c->last_temperature;
    if (avg_logprob) *avg_logprob = c->last_avg_logprob;
    if (attempts) *attempts = c->last_attempts;
}

/* whisper.cpp's acceptance test of one decode (whisper_full_with_state: the fallback loop over temperatures): the sequence up to and including
 * the end-of-text token (or all n_steps), its mean log-probability, and — over its last 32 tokens, when it has more than 32 — the entropy of the
 * token histogram; failed = repetitive (entropy below the threshold) or improbable (mean log-probability below the threshold) */
static bool asr_decode_failed(const tk_asr_whisper_context_s* c, const int32_t* toks, const float* lp, int n_steps, float* avg_out) {
    int len = n_steps;
    for (int i = 0; i < n_steps; ++i)
        if (toks[i] == c->eot) { len = i + 1; break; }
    double sum = 0.0;
    for (int i = 0; i < len; ++i) sum += (double)lp[i];
    const float avg = (float)(sum / (double)len);
    *avg_out = avg;
    bool failed = avg < c->logprob_thold;
    if (len > 32) {
        int cnt[32];
        int32_t ids[32];
        int nd = 0;
        for (int i = len - 32; i < len; ++i) {
            int k = 0;
            for (; k < nd; ++k) if (ids[k] == toks[i]) break;
            if (k == nd) { ids[nd] = toks[i]; cnt[nd] = 0; ++nd; }
            cnt[k]++;
        }
        double ent = 0.0;
        for (int k = 0; k < nd; ++k) { const double pr = cnt[k] / 32.0; ent -= pr * log(pr); }
        if (ent < (double)c->entropy_thold) failed = true;
    }
    return failed;
}

static std::string piece_of(const tk_asr_whisper_context_s* c, int32_t id) {
    const std::vector<std::string>& vocab = c->sh->vocab;
    if (!vocab.empty()) return id >= 0 && id < (int)vocab.size() ? vocab[(size_t)id] : std::string(); /* specials render as nothing */
    /* no GPT-2 BPE vocabulary ships with synthetic / TKWHSP1 weights: ids are rendered symbolically */
    char b[24];
    snprintf(b, sizeof b, " w%d", id);
    return b;
}

tk_error_code_t tk_asr_whisper_process_audio(tk_asr_whisper_context_t* c, const int16_t* audio_data, size_t frame_count, bool is_final,
                                             tk_asr_whisper_result_t** out_result) {
    if (!c || !audio_data || !out_result) return TK_ERROR_INVALID_ARGUMENT;
    *out_result = NULL;
    if (frame_count > TK_ASR_MAX_BUFFER) return afail(TK_ERROR_BUFFER_TOO_SMALL, "chunk longer than the 30 s buffer");
    if (c->buffer_size + frame_count > TK_ASR_MAX_BUFFER) c->buffer_size = 0; /* reference: warn + reset */
    memcpy(c->buffer.data() + c->buffer_size, audio_data, frame_count * sizeof(int16_t));
    c->buffer_size += frame_count;
    tk_asr_whisper_result_t* r = (tk_asr_whisper_result_t*)calloc(1, sizeof(tk_asr_whisper_result_t));
    if (!r) return TK_ERROR_OUT_OF_MEMORY;
    if (c->buffer_size < 16000 && !is_final) { *out_result = r; return TK_SUCCESS; } /* not enough audio yet: empty result */
    std::vector<int32_t> toks(c->decode_steps);
    tk_error_code_t rc = TK_SUCCESS;
    int text_tokens = c->decode_steps; /* tokens that make up the text */
    const bool ref_decode = !c->sh->suppress.empty() && !c->forced_greedy;
    if (ref_decode) {
        /* whisper_full as the reference configures it (tk_asr_whisper.c:89-110): timestamps on, logit filters, the decode's own bookkeeping decides
         * where the text ends (csrc/nn/tk_nn_kernels.h: TkWhFilter).  whisper.cpp returns no segment for less than a second of audio ("input is too
         * short").  With the decoding policy armed, final results fall back through temperatures 0, inc, 2 inc, ... <= 1 while the decode fails —
         * went back in time / ended without a timestamp, or whisper.cpp's log-probability / entropy test; partial results decode once. */
        std::vector<int32_t> prompt;
        std::string why;
        if (!asr_prompt(c, &prompt, &why, true)) { free(r); return afail(TK_ERROR_INFERENCE_FAILED, why); }
        std::vector<float> lp(c->decode_steps);
        text_tokens = 0;
        if (c->buffer_size / 160 >= 100) {
            const float inc = (c->policy_on && is_final) ? c->temperature_inc : -1.0f;
            int attempts = 0;
            for (float t = 0.0f;; t += inc) {
                int32_t rl = 0, stt = 0;
                rc = asr_run_ref(c, c->buffer.data(), (int)c->buffer_size, prompt, c->decode_steps, t, c->policy_seed + (uint64_t)attempts, toks.data(), lp.data(), &rl, &stt);
                if (rc != TK_SUCCESS) break;
                attempts++;
                text_tokens = rl;
                bool failed = stt == 2;
                if (c->policy_on) { /* tk_mi355x_asr_last_decode reports policy decodes only */
                    c->last_attempts = attempts;
                    c->last_temperature = t;
                    if (rl > 0) failed = asr_decode_failed(c, toks.data(), lp.data(), rl, &c->last_avg_logprob) || failed;
                }
                if (!failed || !(inc > 0.0f) || t + inc > 1.0f + 1e-6f) break;
            }
        }
    } else if (c->policy_on) {
        /* the forced decode with the reference's fallback thresholds (tk_asr_whisper.c:126-138): final results fall back through temperatures 0, inc,
         * 2 inc, ... <= 1 while the decode fails whisper.cpp's test; partial results decode once (temperature_inc = -1 there) */
        std::vector<float> lp(c->decode_steps);
        const float inc = is_final ? c->temperature_inc : -1.0f;
        c->last_attempts = 0;
        for (float t = 0.0f;; t += inc) {
            rc = tk_mi355x_asr_transcribe_policy(c, 1, c->buffer.data(), (int)c->buffer_size, c->decode_steps, t, c->policy_seed + (uint64_t)c->last_attempts, toks.data(), lp.data());
            if (rc != TK_SUCCESS) break;
            c->last_attempts++;
            c->last_temperature = t;
            const bool failed = asr_decode_failed(c, toks.data(), lp.data(), c->decode_steps, &c->last_avg_logprob);
            if (!failed || !(inc > 0.0f) || t + inc > 1.0f + 1e-6f) break;
        }
    } else {
        rc = tk_mi355x_asr_transcribe_tokens(c, 1, c->buffer.data(), (int)c->buffer_size, c->decode_steps, toks.data(), nullptr, nullptr, nullptr);
    }
    if (rc != TK_SUCCESS) { free(r); return rc; }
    std::string text;
    for (int i = 0; i < text_tokens && i < (int)toks.size(); ++i) {
        const int32_t t = toks[(size_t)i];
        if (ref_decode) { if (t >= c->eot) continue; } /* segment texts hold the text tokens only: timestamps and eot render as nothing (whisper_full: id < token_eot) */
        else if (t == c->eot) break;
        text += piece_of(c, t);
    }
    r->text = (char*)calloc(text.size() + 1, 1);
    if (!r->text) { free(r); return TK_ERROR_OUT_OF_MEMORY; }
    memcpy(r->text, text.data(), text.size());
    r->text_length = text.size();
    r->is_partial = !is_final;
    r->confidence = 0.9f; /* the reference's placeholder constant (tk_asr_whisper.c:189) */
    if (is_final) { c->buffer_size = 0; c->has_partial = false; c->last_text = text; }
    else c->has_partial = true;
    *out_result = r;
    return TK_SUCCESS;
}

void tk_asr_whisper_free_result(tk_asr_whisper_result_t** result) {
    if (!result || !*result) return;
    free((*result)->text);
    free(*result);
    *result = NULL;
}

tk_error_code_t tk_asr_whisper_reset(tk_asr_whisper_context_t* c) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    c->buffer_size = 0;
    c->has_partial = false;
    c->last_text.clear();
    return TK_SUCCESS;
}

tk_error_code_t tk_asr_whisper_set_language(tk_asr_whisper_context_t* c, const char* language) {
    if (!c || !language) return TK_ERROR_INVALID_ARGUMENT;
    c->language = language; /* like the reference (tk_asr_whisper.c:377-396) any string is stored; a language the vocabulary lacks fails the next decode */
    return TK_SUCCESS;
}

int tk_mi355x_asr_prompt_tokens(tk_asr_whisper_context_t* c, int32_t* out, int cap) {
    if (!c || !out || cap <= 0) return -1;
    std::vector<int32_t> p;
    std::string why;
    if (!asr_prompt(c, &p, &why)) { tk_error_set_detail("%s", why.c_str()); return -1; }
    for (size_t i = 0; i < p.size() && (int)i < cap; ++i) out[i] = p[i];
    return (int)p.size();
}

/* ------------------------------------------------------------------ VAD ------------------ */

#define TK_VAD_MAX_BUFFER (16000 * 30)

struct tk_vad_silero_context_s {
    tk_vad_silero_config_t config;
    TkVadModel model;                  /* synthetic://vad: the stand-in MLP */
    std::unique_ptr<TkVadGraph> graph; /* an .onnx model_path: the graph itself, node by node (csrc/audio/tk_vad_graph.h) */
    tk_vad_silero_state_t state;
    float last_probability = 0.0f, time_since_last_event_ms = 0.0f;
    bool triggered_speech_start = false;
    std::vector<float> audio;
    size_t audio_size = 0;
    uint32_t sample_rate = 16000;
    size_t window = 480, step = 160;
};

static void vad_update(tk_vad_silero_context_s* c, float probability, float dt_ms) {
    c->last_probability = probability;
    c->state.speech_probability = probability;
    c->time_since_last_event_ms += dt_ms;
    if (probability >= c->config.threshold) {
        c->state.speech_duration_ms += dt_ms;
        c->state.silence_duration_ms = 0.0f;
        if (!c->state.is_speech_active && c->state.speech_duration_ms >= c->config.min_speech_duration_ms && !c->triggered_speech_start) {
            c->state.is_speech_active = true;
            c->triggered_speech_start = true;
            c->time_since_last_event_ms = 0.0f;
        }
    } else {
        c->state.silence_duration_ms += dt_ms;
        c->state.speech_duration_ms = 0.0f;
        if (c->state.is_speech_active && c->state.silence_duration_ms >= c->config.min_silence_duration_ms) {
            c->state.is_speech_active = false;
            c->triggered_speech_start = false;
            c->time_since_last_event_ms = 0.0f;
        }
    }
}

tk_error_code_t tk_vad_silero_create(tk_vad_silero_context_t** out_context, const tk_vad_silero_config_t* config) {
    if (!out_context || !config || !config->model_path) return TK_ERROR_INVALID_ARGUMENT;
    if (config->sample_rate != 8000 && config->sample_rate != 16000 && config->sample_rate != 48000) return afail(TK_ERROR_INVALID_ARGUMENT, "unsupported sample rate");
    *out_context = NULL;
    if (!have_gpu()) return afail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible (the MI355X path has no CPU fallback)");
    std::unique_ptr<tk_vad_silero_context_s> c(new tk_vad_silero_context_s());
    c->config = *config;
    c->config.model_path = NULL;
    c->sample_rate = config->sample_rate;
    if (c->config.threshold <= 0.0f) c->config.threshold = 0.5f;
    if (c->config.min_silence_duration_ms <= 0.0f) c->config.min_silence_duration_ms = 300.0f;
    if (c->config.min_speech_duration_ms <= 0.0f) c->config.min_speech_duration_ms = 250.0f;
    if (c->config.speech_pad_ms < 0.0f) c->config.speech_pad_ms = 30.0f;
    memset(&c->state, 0, sizeof c->state);
    c->audio.assign(TK_VAD_MAX_BUFFER, 0.0f);
    c->window = (size_t)c->sample_rate * 30 / 1000;
    c->step = (size_t)c->sample_rate * 10 / 1000;
    const std::string path = config->model_path->path_str ? config->model_path->path_str : "";
    if (path.compare(0, 12, "synthetic://") == 0) {
        if (!c->model.init(tk_mi355x_get_default_device(), (int)c->window, 64) || !c->model.fill_synthetic(seed_of(path, 7))) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
    } else {
        /* the reference's OrtCreateSession on config->model_path (src/sensors/tk_vad_silero.c:110-150) */
        c->graph.reset(new TkVadGraph());
        if (!c->graph->load(path.c_str(), tk_mi355x_get_default_device(), (int)c->window, (int)c->sample_rate)) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->graph->error);
    }
    *out_context = c.release();
    return TK_SUCCESS;
}

void tk_vad_silero_destroy(tk_vad_silero_context_t** context) {
    if (!context || !*context) return;
    delete *context;
    *context = nullptr;
}

tk_error_code_t tk_mi355x_vad_probabilities(tk_vad_silero_context_t* c, const float* windows, int n, float* out) {
    if (!c || !windows || !out || n < 0) return TK_ERROR_INVALID_ARGUMENT;
    if (c->graph) { if (!c->graph->infer(windows, n, out)) return afail(TK_ERROR_INFERENCE_FAILED, c->graph->error); return TK_SUCCESS; }
    if (!c->model.infer(windows, n, out)) return afail(TK_ERROR_INFERENCE_FAILED, c->model.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_vad_silero_process_audio(tk_vad_silero_context_t* c, const int16_t* audio_data, size_t frame_count, float* out_probability) {
    if (!c || !audio_data || !out_probability) return TK_ERROR_INVALID_ARGUMENT;
    *out_probability = 0.0f;
    /* stateless single-window probability: the model window is fixed, shorter input is zero padded, longer truncated */
    std::vector<float> w(c->window, 0.0f);
    for (size_t i = 0; i < frame_count && i < c->window; ++i) w[i] = (float)audio_data[i] / 32768.0f;
    /* a recurrent graph answers this stand-alone query from a cleared state and leaves a cleared state behind (the reference never feeds
     * state at all: one input, src/sensors/tk_vad_silero.c:225-245) */
    if (c->graph && c->graph->stateful() && !c->graph->reset()) return afail(TK_ERROR_INFERENCE_FAILED, c->graph->error);
    tk_error_code_t rc = tk_mi355x_vad_probabilities(c, w.data(), 1, out_probability);
    if (rc == TK_SUCCESS && c->graph && c->graph->stateful() && !c->graph->reset()) return afail(TK_ERROR_INFERENCE_FAILED, c->graph->error);
    return rc;
}

tk_error_code_t tk_mi355x_vad_onnx_probe(const char* path, int32_t* n_nodes, int32_t* n_initialisers, int32_t* n_state_inputs) {
    if (!path) return TK_ERROR_INVALID_ARGUMENT;
    TkOnnxGraph g;
    if (!g.load(path)) return afail(access(path, 0) == 0 ? TK_ERROR_FILE_CORRUPT : TK_ERROR_FILE_NOT_FOUND, g.error);
    std::string err;
    if (!TkVadGraph::check_supported(g, &err)) return afail(TK_ERROR_MODEL_VERIFICATION_FAILED, err);
    int st = 0, fl = 0;
    for (const auto& vi : g.inputs) { if (vi.elem_type == 1 || vi.elem_type == 0) { if (fl++) ++st; } }
    if (n_nodes) *n_nodes = (int32_t)g.nodes.size();
    if (n_initialisers) *n_initialisers = (int32_t)g.init.size();
    if (n_state_inputs) *n_state_inputs = st;
    return TK_SUCCESS;
}

int tk_mi355x_vad_step(tk_vad_silero_context_t* c, float probability) {
    if (!c) return -1;
    const bool before = c->state.is_speech_active;
    vad_update(c, probability, (float)(c->window * 1000) / (float)c->sample_rate);
    if (!before && c->state.is_speech_active) return 0;
    if (before && !c->state.is_speech_active) return 1;
    return -1;
}

tk_error_code_t tk_vad_silero_process_audio_with_events(tk_vad_silero_context_t* c, const int16_t* audio_data, size_t frame_count,
                                                        tk_vad_silero_event_callback_t callback, void* user_data) {
    if (!c || !audio_data) return TK_ERROR_INVALID_ARGUMENT;
    if (frame_count > TK_VAD_MAX_BUFFER) return afail(TK_ERROR_BUFFER_TOO_SMALL, "chunk longer than the 30 s buffer");
    if (c->audio_size + frame_count > TK_VAD_MAX_BUFFER) c->audio_size = 0; /* reference: warn + reset */
    for (size_t i = 0; i < frame_count; ++i) c->audio[c->audio_size + i] = (float)audio_data[i] / 32768.0f;
    c->audio_size += frame_count;
    if (c->audio_size < c->window) return TK_SUCCESS;
    /* every complete window in the buffer, hop = step: one batched GPU call, then the state machine in order */
    size_t nwin = 0;
    for (size_t p = 0; p + c->window <= c->audio_size; p += c->step) ++nwin;
    std::vector<float> wins(nwin * c->window), prob(nwin);
    for (size_t k = 0; k < nwin; ++k) memcpy(&wins[k * c->window], &c->audio[k * c->step], c->window * sizeof(float));
    tk_error_code_t rc = tk_mi355x_vad_probabilities(c, wins.data(), (int)nwin, prob.data());
    if (rc != TK_SUCCESS) return rc;
    const float dt_ms = (float)(c->window * 1000) / (float)c->sample_rate; /* window, not hop: as the reference (:355) */
    bool last = c->state.is_speech_active;
    for (size_t k = 0; k < nwin; ++k) {
        vad_update(c, prob[k], dt_ms);
        if (callback) {
            if (!last && c->state.is_speech_active) callback(TK_VAD_EVENT_SPEECH_STARTED, user_data);
            else if (last && !c->state.is_speech_active) callback(TK_VAD_EVENT_SPEECH_ENDED, user_data);
        }
        last = c->state.is_speech_active;
    }
    const size_t processed = nwin * c->step;
    if (processed > 0 && c->audio_size > processed) {
        memmove(c->audio.data(), c->audio.data() + processed, (c->audio_size - processed) * sizeof(float));
        c->audio_size -= processed;
    } else if (processed > 0) {
        c->audio_size = 0;
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_vad_silero_get_state(tk_vad_silero_context_t* c, tk_vad_silero_state_t* out_state) {
    if (!c || !out_state) return TK_ERROR_INVALID_ARGUMENT;
    *out_state = c->state;
    return TK_SUCCESS;
}

tk_error_code_t tk_vad_silero_reset(tk_vad_silero_context_t* c) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    memset(&c->state, 0, sizeof c->state);
    c->last_probability = 0.0f;
    c->time_since_last_event_ms = 0.0f;
    c->triggered_speech_start = false;
    c->audio_size = 0;
    if (c->graph && !c->graph->reset()) return afail(TK_ERROR_INFERENCE_FAILED, c->graph->error);
    return TK_SUCCESS;
}

tk_error_code_t tk_vad_silero_set_threshold(tk_vad_silero_context_t* c, float threshold) {
    if (!c || threshold < 0.0f || threshold > 1.0f) return TK_ERROR_INVALID_ARGUMENT;
    c->config.threshold = threshold;
    return TK_SUCCESS;
}

} /* extern "C" */
