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

struct tk_asr_whisper_context_s {
    TkWhisperModel model;
    TkAsr asr;
    std::vector<int16_t> buffer;
    size_t buffer_size = 0;
    bool has_partial = false;
    std::string last_text, language;
    int decode_steps = 16;
    int32_t sot = 0, nots = 0, eot = 0;
    bool multilingual = false, translate = false;
    std::vector<std::string> vocab; /* token id -> bytes, from a ggml checkpoint */
    /* whisper.cpp's decoding policy as the reference's wrapper arms it (src/audio/tk_asr_whisper.c:126-138); off unless
     * tk_mi355x_asr_set_decode_policy switched it on (the default path is the plain greedy decode the parity tests pin) */
    bool policy_on = false;
    float temperature_inc = 0.2f, entropy_thold = 2.4f, logprob_thold = -1.0f;
    uint64_t policy_seed = 0;
    float last_temperature = 0.0f, last_avg_logprob = 0.0f;
    int last_attempts = 0;
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
static bool asr_prompt(const tk_asr_whisper_context_s* c, std::vector<int32_t>* out, std::string* why) {
    out->clear();
    out->push_back(c->sot);
    if (c->multilingual) {
        const int n_lang = c->model.hp.n_vocab - 51865 + 99; /* 99 languages at 51865 tokens, 100 (+ yue) at 51866 */
        int id = whisper_lang_id(c->language.empty() || c->language == "auto" ? std::string("en") : c->language); /* no language detection pass: "auto" decodes as English */
        if (id < 0 || id >= n_lang) { *why = "unknown language \"" + c->language + "\""; return false; }
        out->push_back(c->sot + 1 + id);
        out->push_back(c->sot + 1 + n_lang + (c->translate ? 0 : 1)); /* <|translate|>, <|transcribe|> follow the language block */
    }
    out->push_back(c->nots);
    return true;
}

static bool have_gpu() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

static tk_error_code_t asr_new(tk_asr_whisper_context_t** out, const TkWhisperHP& hp, const std::string& path, uint64_t seed, int device, int max_batch) {
    if (!have_gpu()) return afail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible (the MI355X path has no CPU fallback)");
    std::unique_ptr<tk_asr_whisper_context_s> c(new tk_asr_whisper_context_s());
    const bool synthetic = path.empty() || path.compare(0, 12, "synthetic://") == 0;
    if (!synthetic && TkWhisperGgml::is_ggml(path.c_str())) {
        /* the reference's checkpoint format (whisper.cpp ggml .bin): geometry, filter bank, vocabulary and weights come from the file */
        TkWhisperGgml g;
        if (!g.open(path.c_str())) return afail(TK_ERROR_MODEL_LOAD_FAILED, g.error);
        if (!c->model.init(g.hp, device)) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
        if (!c->model.load_ggml(g)) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
        c->vocab.swap(g.vocab);
    } else {
        if (!c->model.init(hp, device)) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
        if (synthetic) {
            if (!c->model.fill_synthetic(seed)) return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
        } else if (!c->model.load_file(path.c_str())) {
            return afail(TK_ERROR_MODEL_LOAD_FAILED, c->model.error);
        }
    }
    if (!c->asr.init(&c->model, max_batch)) return afail(TK_ERROR_GPU_MEMORY, c->asr.error);
    c->buffer.assign(TK_ASR_MAX_BUFFER, 0);
    const int v = c->model.hp.n_vocab;
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
    *out = c.release();
    return TK_SUCCESS;
}

extern "C" {

tk_error_code_t tk_asr_whisper_create(tk_asr_whisper_context_t** out_context, const tk_asr_whisper_config_t* config) {
    if (!out_context || !config || !config->model_path || !config->model_path->path_str) return TK_ERROR_INVALID_ARGUMENT;
    if (config->sample_rate != 16000) return afail(TK_ERROR_INVALID_ARGUMENT, "Whisper needs 16 kHz audio");
    const std::string path = config->model_path->path_str;
    tk_error_code_t rc = asr_new(out_context, tk_whisper_tiny_en(), path, seed_of(path, 6), tk_mi355x_get_default_device(), 1);
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
    return asr_new(out, h, "", seed, device, max_batch);
}

tk_error_code_t tk_mi355x_asr_get_hparams(tk_asr_whisper_context_t* ctx, tk_mi355x_whisper_hparams_t* out) {
    if (!ctx || !out) return TK_ERROR_INVALID_ARGUMENT;
    const TkWhisperHP& h = ctx->model.hp;
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
    if (!c->asr.transcribe(batch, pcm, n_samples, prompt.data(), (int)prompt.size(), n_steps, tokens_out, mel_out ? &mel : nullptr, enc_out ? &enc : nullptr,
                           logits_out ? &lg : nullptr))
        return afail(TK_ERROR_INFERENCE_FAILED, c->asr.error);
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
    if (!c->asr.transcribe_policy(batch, pcm, n_samples, prompt.data(), (int)prompt.size(), n_steps, temperature, seed, tokens_out, logprobs_out))
        return afail(TK_ERROR_INFERENCE_FAILED, c->asr.error);
    return TK_SUCCESS;
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
    if (!c->vocab.empty()) return id >= 0 && id < (int)c->vocab.size() ? c->vocab[(size_t)id] : std::string(); /* specials render as nothing */
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
    if (c->policy_on) {
        /* the reference's parameters (tk_asr_whisper.c:126-138): final results fall back through temperatures 0, inc, 2 inc, ... <= 1 while the decode
         * fails whisper.cpp's test; partial results decode once (temperature_inc = -1 there) */
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
    for (int32_t t : toks) {
        if (t == c->eot) break;
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
