/*
 * tk_abi_llm.cpp — tk_model_loader_* / tk_llm_runner_* (reference surface) and the
 * tk_mi355x_llm_* extension entry points on top of the HIP engine.
 *
 * Behaviour mirrored from the reference:
 *   loader: find-or-load by path, handle is an opaque pointer (src/ai_models/tk_model_loader.c:918-1083,
 *           tk_model_loader_private.h:43-51)
 *   runner: prepare = tokenise(add_bos) -> clear KV -> prefill whole prompt (tk_runner_streaming.c:20-34);
 *           next_token = sample -> EOS? NULL -> decode one token -> piece valid until next call (:57-85);
 *           add_tool_response formats "[TOOL_RESULT] name: \"%s\", output: %s [/TOOL_RESULT]" and decodes it
 *           at n_past (tk_runner_helpers.c:78-126); reset_context clears the KV state (:128-138).
 * There is no CPU fallback: every entry fails with a GPU error when no gfx950 device is usable.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../llm/tk_gguf.h"
#include "../llm/tk_grammar.h"
#include "../llm/tk_llm_batcher.h"
#include "../llm/tk_llm_engine.h"
#include "../llm/tk_lora.h"
#include "../llm/tk_llm_pipe.h"
#include "../llm/tk_tokenizer.h"
#include "../common/tk_ggml_blocks.h"
#include "tk/tk_mi355x_ext.h"
#include "tk/tk_model_runner.h"

struct tk_mi355x_llm_model_s {
    TkLlmModel model;
    TkTokenizer tok;
    int context_length = 4096;
    std::string path;
    std::unique_ptr<TkLoraAdapter> lora; /* the adapter merged into this model's matrices at load (tk_lora.h); part of the registry key */
    int refcount = 1;
    /* continuous batching behind tk_llm_runner_*: the runners created on this model share decode sessions (declared after `model`:
     * destroyed before it) */
    std::mutex batch_mu;
    std::vector<std::unique_ptr<TkLlmBatcher>> batchers;
    int runner_slots = 0; /* sequences per shared session; 0 = $TK_MI355X_RUNNER_SLOTS or 16 */
};

struct tk_mi355x_llm_session_s {
    TkLlmSession session;
};

static tk_error_code_t fail(tk_error_code_t code, const std::string& why) {
    tk_error_set_detail("%s", why.c_str());
    return code;
}

static TkLlmHParams to_hp(const tk_mi355x_llm_hparams_t& h) {
    TkLlmHParams o{};
    o.n_layer = h.n_layer; o.d_model = h.d_model; o.n_head = h.n_head; o.n_kv_head = h.n_kv_head; o.head_dim = h.head_dim;
    o.d_ff = h.d_ff; o.vocab = h.vocab; o.rms_eps = h.rms_eps; o.rope_theta = h.rope_theta;
    o.ks_qkv = h.ks_qkv; o.ks_o = h.ks_o; o.ks_gateup = h.ks_gateup; o.ks_down = h.ks_down; o.ks_out = h.ks_out;
    return o;
}

/* K-split plan: at least one 64-row workgroup per CU (256) where the matrix allows it */
static void default_plan(tk_mi355x_llm_hparams_t* h) {
    auto pick = [](int64_t rows, int64_t K) {
        if (rows < 64 || K < 256) return 1; /* geometry the model rejects later (TkLlmModel::init): no plan to make */
        int nb = (int)(K / 256);
        int want = (int)((256 + rows / 64 - 1) / (rows / 64));
        int ks = 1;
        /* prefer K-ranges of a multiple of 4 blocks: the GEMV keeps 4 weight tiles in flight per wave */
        for (int c = 1; c <= nb && c <= 8; ++c) /* at most 8 K-ranges: the consumers keep every partial slab of a row in flight at once */
            if (nb % c == 0 && ((nb / c) % 4 == 0 || nb < 4 * want)) { ks = c; if (c >= want) break; }
        if (K / ks > 4096) { /* LDS image of the K-range must fit: 16 B per k */
            for (int c = ks; c <= nb; ++c) if (nb % c == 0 && K / c <= 4096) { ks = c; break; }
        }
        return ks;
    };
    const int64_t qd = (int64_t)h->n_head * h->head_dim, kvd = (int64_t)h->n_kv_head * h->head_dim;
    h->ks_qkv = pick(qd + 2 * kvd, h->d_model);
    h->ks_o = pick(h->d_model, qd);
    h->ks_gateup = pick(2 * (int64_t)h->d_ff, h->d_model);
    h->ks_down = pick(h->d_model, h->d_ff);
    h->ks_out = 1;
}

extern "C" {

tk_error_code_t tk_mi355x_llm_model_create(tk_mi355x_llm_model_t** out, const tk_mi355x_llm_hparams_t* hp, int device) {
    if (!out || !hp) return TK_ERROR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TK_ERROR_INVALID_ARGUMENT, "device ordinal out of range");
    tk_mi355x_llm_hparams_t h = *hp;
    if (h.ks_qkv <= 0 || h.ks_o <= 0 || h.ks_gateup <= 0 || h.ks_down <= 0) default_plan(&h);
    std::unique_ptr<tk_mi355x_llm_model_s> m(new tk_mi355x_llm_model_s());
    if (!m->model.init(to_hp(h), device)) return fail(TK_ERROR_MODEL_LOAD_FAILED, m->model.error);
    m->tok.init_bytes(h.vocab);
    *out = m.release();
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_model_fill_synthetic(tk_mi355x_llm_model_t* m, uint64_t seed) {
    if (!m) return TK_ERROR_INVALID_ARGUMENT;
    if (!m->model.fill_synthetic(seed)) return fail(TK_ERROR_GPU_ROCM_ERROR, m->model.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_model_fill_synthetic_f16(tk_mi355x_llm_model_t* m, uint64_t seed) {
    if (!m) return TK_ERROR_INVALID_ARGUMENT;
    if (!m->model.fill_synthetic(seed, true)) return fail(TK_ERROR_GPU_ROCM_ERROR, m->model.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_model_set_lora(tk_mi355x_llm_model_t* m, const char* adapter_path) {
    if (!m) return TK_ERROR_INVALID_ARGUMENT;
    if (!adapter_path || !*adapter_path) { m->model.lora = nullptr; m->lora.reset(); return TK_SUCCESS; }
    std::unique_ptr<TkLoraAdapter> ad(new TkLoraAdapter());
    if (!ad->load(adapter_path)) return fail(TK_ERROR_MODEL_LOAD_FAILED, "LoRA adapter: " + ad->error);
    const TkLlmHParams& hp = m->model.hp;
    for (const auto& t : ad->tensors) {
        int64_t rows = 0, cols = 0;
        if (t.layer >= hp.n_layer) return fail(TK_ERROR_MODEL_LOAD_FAILED, "LoRA adapter names a layer this model does not have");
        m->model.shape(t.layer, t.which, &rows, &cols);
        if (rows != t.n_out || cols != t.k_in) return fail(TK_ERROR_MODEL_LOAD_FAILED, "LoRA adapter does not fit this model (factor shapes against the base matrix)");
    }
    m->lora = std::move(ad);
    m->model.lora = m->lora.get();
    return TK_SUCCESS;
}

int tk_mi355x_llm_model_lora_merged(const tk_mi355x_llm_model_t* m) { return m ? m->model.lora_merged : 0; }

tk_error_code_t tk_mi355x_lora_probe(const char* path, int32_t* rank, float* alpha, int32_t* n_tensors) {
    if (!path) return TK_ERROR_INVALID_ARGUMENT;
    TkLoraAdapter ad;
    if (!ad.load(path)) return fail(TK_ERROR_FILE_CORRUPT, "LoRA adapter: " + ad.error);
    if (rank) *rank = ad.r;
    if (alpha) *alpha = ad.alpha;
    if (n_tensors) *n_tensors = (int32_t)ad.tensors.size();
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_model_set_tensor(tk_mi355x_llm_model_t* m, int layer, int which, int type, const void* data, size_t nbytes) {
    if (!m || !data) return TK_ERROR_INVALID_ARGUMENT;
    if (!m->model.set_tensor(layer, which, type, data, nbytes)) return fail(TK_ERROR_INVALID_INPUT_TENSOR, m->model.error);
    return TK_SUCCESS;
}

void tk_mi355x_llm_model_get_hparams(const tk_mi355x_llm_model_t* m, tk_mi355x_llm_hparams_t* out) {
    if (!m || !out) return;
    const TkLlmHParams& h = m->model.hp;
    *out = tk_mi355x_llm_hparams_t{h.n_layer, h.d_model, h.n_head, h.n_kv_head, h.head_dim, h.d_ff, h.vocab, h.rms_eps, h.rope_theta,
                                   h.ks_qkv, h.ks_o, h.ks_gateup, h.ks_down, h.ks_out};
}

uint64_t tk_mi355x_llm_model_weight_bytes(const tk_mi355x_llm_model_t* m) {
    if (!m) return 0;
    uint64_t b = m->model.output.bytes;
    for (const auto& L : m->model.layers) b += L.q.bytes + L.k.bytes + L.v.bytes + L.o.bytes + L.gate.bytes + L.up.bytes + L.down.bytes;
    return b;
}

/* Weights are immutable and 4.3 GB: a model file resident on a device is shared by every loader of the process (the reference caches
 * per loader, src/ai_models/tk_model_loader.c:918-1083; K cortex handles — each with its own loader, tk_cortex_main.c:779-925 — would
 * otherwise hold K copies and could not share decode passes).  force_reload loads a private copy.
 * Lifetime: refcount = the creator / the loaders that hold the handle + every live tk_llm_runner_t created on it (a runner keeps raw
 * pointers into the model's shared decode sessions, so the model must outlive it whatever order the host destroys things in). */
static std::mutex g_models_mu;
static std::vector<tk_mi355x_llm_model_t*> g_models;

static void release_model(tk_mi355x_llm_model_t* m) { /* g_models_mu held */
    if (--m->refcount > 0) return;
    for (size_t i = 0; i < g_models.size(); ++i)
        if (g_models[i] == m) { g_models.erase(g_models.begin() + i); break; }
    delete m;
}

void tk_mi355x_llm_model_destroy(tk_mi355x_llm_model_t** m) {
    if (!m || !*m) return;
    {
        std::lock_guard<std::mutex> gl(g_models_mu);
        release_model(*m); /* freed once the last runner created on it is gone too */
    }
    *m = nullptr;
}

/* GGUF metadata only (no GPU): lets the loader report geometry errors before touching the device */
tk_error_code_t tk_mi355x_gguf_probe(const char* path, tk_mi355x_llm_hparams_t* out, int32_t* n_vocab_tokens) {
    if (!path || !out) return TK_ERROR_INVALID_ARGUMENT;
    TkGgufFile f;
    if (!f.open(path)) return fail(access(path, 0) == 0 ? TK_ERROR_FILE_CORRUPT : TK_ERROR_FILE_NOT_FOUND, f.error);
    auto it = f.str.find("general.architecture");
    std::string arch = it == f.str.end() ? "llama" : it->second;
    tk_mi355x_llm_hparams_t h{};
    h.n_layer = (int)f.get(arch + ".block_count", 0);
    h.d_model = (int)f.get(arch + ".embedding_length", 0);
    h.d_ff = (int)f.get(arch + ".feed_forward_length", 0);
    h.n_head = (int)f.get(arch + ".attention.head_count", 0);
    h.n_kv_head = (int)f.get(arch + ".attention.head_count_kv", h.n_head);
    h.head_dim = (int)f.get(arch + ".rope.dimension_count", h.n_head ? h.d_model / h.n_head : 0);
    h.rms_eps = (float)f.get(arch + ".attention.layer_norm_rms_epsilon", 1e-5);
    h.rope_theta = (float)f.get(arch + ".rope.freq_base", 10000.0);
    const TkGgufTensor* te = f.find("token_embd.weight");
    h.vocab = te && te->dims.size() == 2 ? (int)te->dims[1] : (int)f.tokens.size();
    if (h.n_layer <= 0 || h.d_model <= 0 || h.n_head <= 0 || h.vocab <= 0 || h.d_ff <= 0 || h.n_kv_head <= 0 || h.head_dim <= 0)
        return fail(TK_ERROR_MODEL_LOAD_FAILED, "GGUF lacks llama hyper-parameters");
    default_plan(&h);
    *out = h;
    if (n_vocab_tokens) *n_vocab_tokens = (int32_t)f.tokens.size();
    return TK_SUCCESS;
}

/* tokenises with the vocabulary stored in a GGUF file (CPU only; byte tokens when the file has no vocabulary) */
int tk_mi355x_gguf_tokenize(const char* path, const char* text, int add_bos, int32_t* ids, int cap) {
    if (!path || !text || !ids || cap <= 0) return -1;
    TkGgufFile f;
    if (!f.open(path)) { tk_error_set_detail("%s", f.error.c_str()); return -1; }
    TkTokenizer t;
    if (!f.tokens.empty()) t.init_spm(f.tokens, f.scores, f.token_type, (int)f.get("tokenizer.ggml.bos_token_id", 1), (int)f.get("tokenizer.ggml.eos_token_id", 2));
    else t.init_bytes(1 << 30);
    std::vector<int32_t> v = t.encode(text, add_bos != 0);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) ids[i] = v[i];
    return (int)v.size();
}

tk_error_code_t tk_mi355x_llm_model_load_gguf(tk_mi355x_llm_model_t** out, const char* path, int device) {
    return tk_mi355x_llm_model_load_gguf_lora(out, path, nullptr, device);
}

tk_error_code_t tk_mi355x_llm_model_load_gguf_lora(tk_mi355x_llm_model_t** out, const char* path, const char* lora_path, int device) {
    if (!out || !path) return TK_ERROR_INVALID_ARGUMENT;
    tk_mi355x_llm_hparams_t h{};
    tk_error_code_t rc = tk_mi355x_gguf_probe(path, &h, nullptr);
    if (rc != TK_SUCCESS) return rc;
    TkGgufFile f;
    if (!f.open(path)) return fail(TK_ERROR_FILE_CORRUPT, f.error);
    tk_mi355x_llm_model_t* m = nullptr;
    rc = tk_mi355x_llm_model_create(&m, &h, device);
    if (rc != TK_SUCCESS) return rc;
    if (lora_path && *lora_path) {
        rc = tk_mi355x_llm_model_set_lora(m, lora_path);
        if (rc != TK_SUCCESS) { delete m; return rc; } /* private to this call, like the failure path below */
    }
    auto put = [&](int layer, int which, const std::string& name, const char* alt) -> bool {
        const TkGgufTensor* t = f.find(name);
        if (!t && alt) t = f.find(alt);
        if (!t) { tk_error_set_detail("GGUF tensor missing: %s", name.c_str()); return false; }
        if (!t->data) { tk_error_set_detail("GGUF tensor %s has unsupported type %u (supported: F32, Q4_K, Q6_K)", name.c_str(), t->type); return false; }
        if (!m->model.set_tensor(layer, which, (int)t->type, t->data, t->nbytes)) { tk_error_set_detail("%s: %s", name.c_str(), m->model.error.c_str()); return false; }
        return true;
    };
    bool ok = put(-1, TK_T_TOKEN_EMBD, "token_embd.weight", nullptr) && put(-1, TK_T_OUT_NORM, "output_norm.weight", nullptr) &&
              put(-1, TK_T_OUTPUT, "output.weight", "token_embd.weight");
    static const char* names[TK_L_COUNT] = {"attn_norm", "attn_q", "attn_k", "attn_v", "attn_output", "ffn_norm", "ffn_gate", "ffn_up", "ffn_down"};
    for (int l = 0; l < h.n_layer && ok; ++l)
        for (int w = 0; w < TK_L_COUNT && ok; ++w) {
            char nm[96];
            snprintf(nm, sizeof nm, "blk.%d.%s.weight", l, names[w]);
            ok = put(l, w, nm, nullptr);
        }
    /* `m` is private to this call (never registered, no runner holds it): freed directly.  NOT tk_mi355x_llm_model_destroy — that takes
     * g_models_mu, which tk_model_loader_load_model holds across this call (a GGUF with a missing tensor hung the loader) */
    if (!ok) { delete m; return TK_ERROR_MODEL_LOAD_FAILED; }
    if (m->lora) { m->model.lora = nullptr; m->lora->drop_factors(); } /* every matrix is installed: nothing reads the factors any more */
    if (!f.tokens.empty()) m->tok.init_spm(f.tokens, f.scores, f.token_type, (int)f.get("tokenizer.ggml.bos_token_id", 1), (int)f.get("tokenizer.ggml.eos_token_id", 2));
    auto it = f.str.find("general.architecture");
    m->context_length = (int)f.get((it == f.str.end() ? std::string("llama") : it->second) + ".context_length", 4096);
    m->path = path;
    *out = m;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_session_create(tk_mi355x_llm_session_t** out, tk_mi355x_llm_model_t* m, int max_seq, int max_ctx) {
    if (!out || !m) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_ptr<tk_mi355x_llm_session_s> s(new tk_mi355x_llm_session_s());
    if (!s->session.init(&m->model, max_seq, max_ctx)) return fail(TK_ERROR_GPU_MEMORY, s->session.error);
    *out = s.release();
    return TK_SUCCESS;
}

void tk_mi355x_llm_session_destroy(tk_mi355x_llm_session_t** s) {
    if (!s || !*s) return;
    delete *s;
    *s = nullptr;
}

tk_error_code_t tk_mi355x_llm_forward(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok,
                                      float* logits, int32_t* argmax) {
    if (!s || !seq || !pos || !tok) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.forward(nrows, seq, pos, tok, logits, argmax, true)) return fail(TK_ERROR_INFERENCE_FAILED, s->session.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_forward_sampled(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok,
                                              const tk_mi355x_sampling_t* sampling, float* logits, int32_t* ids) {
    if (!s || !seq || !pos || !tok || !sampling || nrows < 1 || nrows > TK_MAX_ROWS) return TK_ERROR_INVALID_ARGUMENT;
    TkSampleRow rows[TK_MAX_ROWS] = {};
    for (int r = 0; r < nrows; ++r) {
        const tk_mi355x_sampling_t& p = sampling[r];
        /* the checks tk_mi355x_llm_runner_set_sampling makes, per row */
        if (!(p.temperature >= 0.0f) || p.top_k < 0 || p.top_k > TK_SAMPLE_MAX_K || !(p.top_p > 0.0f) || p.top_p > 1.0f || !(p.min_p >= 0.0f) || p.min_p > 1.0f)
            return fail(TK_ERROR_INVALID_ARGUMENT, "sampling parameters of row " + std::to_string(r) + " are out of range (temperature >= 0, 0 <= top_k <= 64, 0 < top_p <= 1, 0 <= min_p <= 1)");
        rows[r].temp = p.temperature; rows[r].top_p = p.top_p; rows[r].min_p = p.min_p; rows[r].top_k = p.top_k; rows[r].seed = p.seed; rows[r].counter = p.counter;
    }
    if (!s->session.forward(nrows, seq, pos, tok, logits, ids, true, nullptr, rows)) return fail(TK_ERROR_INFERENCE_FAILED, s->session.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_session_kv_write(tk_mi355x_llm_session_t* s, int layer, int seq, int pos0, int n_pos, const uint16_t* k, const uint16_t* v) {
    if (!s || !k || !v) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.kv_write(layer, seq, pos0, n_pos, k, v)) return fail(TK_ERROR_INVALID_ARGUMENT, s->session.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_session_kv_read(tk_mi355x_llm_session_t* s, int layer, int seq, int pos0, int n_pos, uint16_t* k, uint16_t* v) {
    if (!s || !k || !v) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.kv_read(layer, seq, pos0, n_pos, k, v)) return fail(TK_ERROR_INVALID_ARGUMENT, s->session.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_prefill(tk_mi355x_llm_session_t* s, int nseq, int n_prompt, const int32_t* tokens, int32_t* first_tokens) {
    if (!s || !tokens) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.prefill(nseq, n_prompt, tokens, first_tokens)) return fail(TK_ERROR_INFERENCE_FAILED, s->session.error);
    return TK_SUCCESS;
}

int tk_mi355x_llm_max_rows(void) { return TK_MAX_ROWS; }

tk_error_code_t tk_mi355x_llm_forward_stage(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok,
                                            const float* x_in, float* x_out, int x_on_host, int layer0, int layer1, int head, int32_t* argmax) {
    if (!s || !seq || !pos) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.forward_stage(nrows, seq, pos, tok, x_in, x_out, x_on_host != 0, layer0, layer1, head != 0, argmax))
        return fail(TK_ERROR_INFERENCE_FAILED, s->session.error);
    return TK_SUCCESS;
}

/* ---- layer-sharded pipeline with the hand-off inside the library (csrc/llm/tk_llm_pipe.h) ---- */
struct tk_mi355x_pipe_s { TkLlmPipe pipe; };
static_assert(sizeof(tk_mi355x_pipe_handle_t) == sizeof(TkPipeHandle), "public and internal pipe handles must have one layout");

tk_error_code_t tk_mi355x_pipe_create(tk_mi355x_pipe_t** out, tk_mi355x_llm_session_t* s, int stage, int n_stages, int layer0, int layer1, int payload_f16,
                                      tk_mi355x_pipe_handle_t* my_handle) {
    if (!out || !s) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_ptr<tk_mi355x_pipe_s> p(new tk_mi355x_pipe_s());
    if (!p->pipe.init(&s->session, stage, n_stages, layer0, layer1, payload_f16 != 0, (TkPipeHandle*)my_handle)) return fail(TK_ERROR_INVALID_ARGUMENT, p->pipe.error);
    *out = p.release();
    return TK_SUCCESS;
}

void tk_mi355x_pipe_destroy(tk_mi355x_pipe_t** p) {
    if (!p || !*p) return;
    delete *p;
    *p = nullptr;
}

tk_error_code_t tk_mi355x_pipe_connect(tk_mi355x_pipe_t* p, const tk_mi355x_pipe_handle_t* next, const tk_mi355x_pipe_handle_t* prev) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.connect((const TkPipeHandle*)next, (const TkPipeHandle*)prev)) return fail(TK_ERROR_GPU_ROCM_ERROR, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_connect_local(tk_mi355x_pipe_t* p, tk_mi355x_pipe_t* next, tk_mi355x_pipe_t* prev) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.connect_local(next ? &next->pipe : nullptr, prev ? &prev->pipe : nullptr)) return fail(TK_ERROR_GPU_ROCM_ERROR, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_rccl_unique_id(uint8_t out[128]) {
    if (!out) return TK_ERROR_INVALID_ARGUMENT;
    std::string err;
    if (!tk_pipe_rccl_unique_id(out, &err)) return fail(TK_ERROR_GPU_ROCM_ERROR, err);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_connect_rccl(tk_mi355x_pipe_t* p, const uint8_t unique_id[128]) {
    if (!p || !unique_id) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.connect_rccl(unique_id)) return fail(TK_ERROR_GPU_DEVICE_NOT_FOUND, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_pass(tk_mi355x_pipe_t* p, int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, int head) {
    if (!p || !seq || !pos) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.pass(nrows, seq, pos, tok, head != 0)) return fail(TK_ERROR_INFERENCE_FAILED, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_decode(tk_mi355x_pipe_t* p, int nrows, int n_steps) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.decode(nrows, n_steps)) return fail(TK_ERROR_INFERENCE_FAILED, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_pipe_sync(tk_mi355x_pipe_t* p, int32_t* out_tokens, int n_steps) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    if (!p->pipe.sync(out_tokens, n_steps)) return fail(TK_ERROR_TIMEOUT, p->pipe.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_decode(tk_mi355x_llm_session_t* s, int nrows, int n_steps, int32_t* out_tokens, float* ms_per_step) {
    if (!s) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.decode(nrows, n_steps, out_tokens)) return fail(TK_ERROR_INFERENCE_FAILED, s->session.error);
    if (ms_per_step) *ms_per_step = s->session.last_step_ms;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_time_gemv(tk_mi355x_llm_session_t* s, int layer, int which, int nrows, int iters, float* avg_ms,
                                        double* algorithmic_bytes) {
    if (!s || !avg_ms || !algorithmic_bytes || iters <= 0) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.time_gemv(layer, which, nrows, iters, avg_ms, algorithmic_bytes)) return fail(TK_ERROR_GPU_ROCM_ERROR, s->session.error);
    return TK_SUCCESS;
}

int tk_mi355x_device_cu_count(int device) {
    int n = 0;
    return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess ? n : -1;
}

tk_error_code_t tk_mi355x_attention_plan(int device, int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, int fused, int32_t out[4]) {
    if (!out || nrows < 1 || nrows > TK_MAX_ROWS || n_head < 1 || n_kv_head < 1 || n_head % n_kv_head || head_dim < 1 || max_ctx < 1) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return fail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no such HIP device");
    const TkAttentionPlan pl = tk_attention_plan(nrows, n_head, n_kv_head, head_dim, max_ctx, fused != 0);
    out[0] = pl.kernel; out[1] = pl.gq; out[2] = pl.chunk; out[3] = pl.slots;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_attention_plan_at(int device, int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, int fused, int top_position, int32_t out[4]) {
    const tk_error_code_t rc = tk_mi355x_attention_plan(device, nrows, n_head, n_kv_head, head_dim, max_ctx, fused, out);
    if (rc != TK_SUCCESS) return rc;
    if (top_position < 0 || top_position >= max_ctx) return TK_ERROR_INVALID_ARGUMENT;
    /* the session's per-pass choices (TkLlmSession::choose_attention_top) on top of the launcher's */
    if (fused) {
        if (max_ctx > TK_LONG_ATT_MIN_POS && tk_attention_long_applies(nrows, n_head, n_kv_head, head_dim) && top_position >= tk_long_att_min_pos(nrows)) {
            out[0] = 3; out[1] = n_head / n_kv_head; out[2] = 64; out[3] = 1;
        }
    } else if (out[0] == 2 && top_position < 128) {
        out[0] = 0;
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_time_attention(tk_mi355x_llm_session_t* s, int nrows, int ctx, int iters, float* avg_ms, double* kv_bytes) {
    if (!s || !avg_ms || !kv_bytes) return TK_ERROR_INVALID_ARGUMENT;
    if (!s->session.time_attention(nrows, ctx, iters, avg_ms, kv_bytes)) return fail(TK_ERROR_GPU_ROCM_ERROR, s->session.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_llm_model_set_runner_slots(void* model_handle, int slots) {
    if (!model_handle || slots < 1 || slots > TK_MAX_ROWS) return TK_ERROR_INVALID_ARGUMENT;
    tk_mi355x_llm_model_t* m = (tk_mi355x_llm_model_t*)model_handle;
    std::lock_guard<std::mutex> lk(m->batch_mu);
    m->runner_slots = slots;
    return TK_SUCCESS;
}

void tk_mi355x_llm_model_batch_stats(void* model_handle, uint64_t* passes, uint64_t* rows, int32_t* max_rows_in_a_pass) {
    uint64_t p = 0, r = 0;
    int mx = 0;
    if (model_handle) {
        tk_mi355x_llm_model_t* m = (tk_mi355x_llm_model_t*)model_handle;
        std::lock_guard<std::mutex> lk(m->batch_mu);
        for (auto& b : m->batchers) {
            uint64_t bp, br;
            int bm;
            b->stats(&bp, &br, &bm);
            p += bp; r += br; mx = bm > mx ? bm : mx;
        }
    }
    if (passes) *passes = p;
    if (rows) *rows = r;
    if (max_rows_in_a_pass) *max_rows_in_a_pass = mx;
}

uint64_t tk_mi355x_llm_model_run_ahead_wasted(void* model_handle) {
    uint64_t w = 0;
    if (model_handle) {
        tk_mi355x_llm_model_t* m = (tk_mi355x_llm_model_t*)model_handle;
        std::lock_guard<std::mutex> lk(m->batch_mu);
        for (auto& b : m->batchers) {
            uint64_t bw = 0;
            b->stats(nullptr, nullptr, nullptr, &bw);
            w += bw;
        }
    }
    return w;
}

/* grammar engine entry points (no GPU involved): used by the CPU tests and by hosts that want the tool-call text */
tk_error_code_t tk_mi355x_grammar_check(const char* gbnf, const char* text, int32_t* n_accepted, int32_t* complete) {
    if (!text || !n_accepted || !complete) return TK_ERROR_INVALID_ARGUMENT;
    TkGrammar g;
    std::string err;
    if (!g.parse(gbnf ? gbnf : TK_DEFAULT_TOOL_CALL_GBNF, &err)) return fail(TK_ERROR_CONFIG_PARSE_FAILED, "GBNF: " + err);
    TkGrammarState st;
    st.init(&g);
    int32_t n = 0;
    for (const char* c = text; *c; ++c, ++n)
        if (!st.accept((uint8_t)*c)) break;
    *n_accepted = n;
    *complete = st.complete() ? 1 : 0;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_grammar_next_bytes(const char* gbnf, const char* prefix, uint8_t allowed[256], int32_t* complete) {
    if (!prefix || !allowed || !complete) return TK_ERROR_INVALID_ARGUMENT;
    TkGrammar g;
    std::string err;
    if (!g.parse(gbnf ? gbnf : TK_DEFAULT_TOOL_CALL_GBNF, &err)) return fail(TK_ERROR_CONFIG_PARSE_FAILED, "GBNF: " + err);
    TkGrammarState st;
    st.init(&g);
    if (!st.accept(std::string(prefix))) return fail(TK_ERROR_INVALID_ARGUMENT, "prefix is not in the grammar");
    for (int b = 0; b < 256; ++b) {
        TkGrammarState t = st;
        allowed[b] = t.accept((uint8_t)b) ? 1 : 0;
    }
    *complete = st.complete() ? 1 : 0;
    return TK_SUCCESS;
}

/* ------------------------------------------------------------------ reference surface ------ */

struct tk_model_loader_s {
    std::mutex mu;
    std::vector<tk_mi355x_llm_model_t*> held; /* one entry per successful load_model: the references this loader owns */
    uint32_t max_models = 4;
};

static size_t distinct_held(const tk_model_loader_s* l) {
    size_t n = 0;
    for (size_t i = 0; i < l->held.size(); ++i) {
        bool seen = false;
        for (size_t j = 0; j < i; ++j) seen = seen || l->held[j] == l->held[i];
        n += seen ? 0 : 1;
    }
    return n;
}

tk_error_code_t tk_model_loader_create(tk_model_loader_t** out_loader, const tk_model_loader_config_t* config) {
    if (!out_loader || !config) return TK_ERROR_INVALID_ARGUMENT;
    tk_model_loader_s* l = new tk_model_loader_s();
    l->max_models = config->max_models ? config->max_models : 4;
    *out_loader = l;
    return TK_SUCCESS;
}

void tk_model_loader_destroy(tk_model_loader_t** loader) {
    if (!loader || !*loader) return;
    {
        std::lock_guard<std::mutex> gl(g_models_mu);
        for (auto* m : (*loader)->held) release_model(m);
    }
    delete *loader;
    *loader = nullptr;
}

static bool parse_synthetic(const std::string& p, std::string* name, uint64_t* seed) {
    const std::string pre = "synthetic://";
    if (p.compare(0, pre.size(), pre) != 0) return false;
    std::string rest = p.substr(pre.size());
    *seed = 4;
    size_t q = rest.find('?');
    *name = rest.substr(0, q);
    if (q != std::string::npos) {
        size_t s = rest.find("seed=", q);
        if (s != std::string::npos) *seed = strtoull(rest.c_str() + s + 5, nullptr, 10);
    }
    return true;
}

tk_error_code_t tk_model_loader_load_model(tk_model_loader_t* loader, const tk_model_load_params_t* params, void** out_model_handle) {
    if (!loader || !params || !out_model_handle || !params->model_path || !params->model_path->path_str) return TK_ERROR_INVALID_ARGUMENT;
    if (params->model_type == TK_MODEL_FORMAT_ONNX) return fail(TK_ERROR_NOT_IMPLEMENTED, "ONNX graphs are not interpreted: the detector/ASR/VAD streams have dedicated entry points");
    std::lock_guard<std::mutex> lk(loader->mu);
    std::lock_guard<std::mutex> gl(g_models_mu); /* also serialises loads: two threads asking for the same file get one copy */
    const std::string lora = params->lora_adapter ? params->lora_adapter : "";
    /* the registry's key: a model with an adapter merged in (tk_model_loader.c:259-270) is another model than the file alone */
    const std::string path = std::string(params->model_path->path_str) + (lora.empty() ? "" : "\n+lora=" + lora);
    const std::string file = params->model_path->path_str;
    const int device = tk_mi355x_get_default_device();
    if (!params->force_reload)
        for (auto* m : g_models)
            if (m->path == path && m->model.device == device) {
                bool mine = false;
                for (auto* h : loader->held) mine = mine || h == m;
                if (!mine && distinct_held(loader) >= loader->max_models) return fail(TK_ERROR_OUT_OF_MEMORY, "model cache full (max_models)");
                m->refcount++;
                loader->held.push_back(m);
                *out_model_handle = m;
                return TK_SUCCESS;
            }
    if (distinct_held(loader) >= loader->max_models) return fail(TK_ERROR_OUT_OF_MEMORY, "model cache full (max_models)");
    tk_mi355x_llm_model_t* m = nullptr;
    std::string name;
    uint64_t seed;
    tk_error_code_t rc;
    if (parse_synthetic(file, &name, &seed)) {
        tk_mi355x_llm_hparams_t h{};
        const bool f16 = name.size() > 4 && name.compare(name.size() - 4, 4, "-f16") == 0; /* the fp16 checkpoint recipe (BASELINE configs[4]) */
        if (f16) name.resize(name.size() - 4);
        if (name == "mistral-7b") h = tk_mi355x_llm_hparams_t{32, 4096, 32, 8, 128, 14336, 32000, 1e-5f, 10000.0f, 0, 0, 0, 0, 1};
        else if (name == "tiny") h = tk_mi355x_llm_hparams_t{2, 256, 8, 2, 64, 512, 512, 1e-5f, 10000.0f, 0, 0, 0, 0, 1};
        else return fail(TK_ERROR_FILE_NOT_FOUND, "unknown synthetic model: " + name);
        rc = tk_mi355x_llm_model_create(&m, &h, device);
        if (rc == TK_SUCCESS && !lora.empty()) rc = tk_mi355x_llm_model_set_lora(m, lora.c_str());
        if (rc == TK_SUCCESS) rc = f16 ? tk_mi355x_llm_model_fill_synthetic_f16(m, seed) : tk_mi355x_llm_model_fill_synthetic(m, seed);
        if (rc != TK_SUCCESS) { if (m) release_model(m); return rc; } /* g_models_mu is held here */
        if (m->lora) { m->model.lora = nullptr; m->lora->drop_factors(); }
        m->path = path;
    } else {
        rc = tk_mi355x_llm_model_load_gguf_lora(&m, file.c_str(), lora.c_str(), device);
        if (rc != TK_SUCCESS) return rc;
        m->path = path;
    }
    if (!params->force_reload) g_models.push_back(m); /* a force-reloaded copy stays private to its handle */
    loader->held.push_back(m);
    *out_model_handle = m;
    return TK_SUCCESS;
}

tk_error_code_t tk_model_loader_unload_model(tk_model_loader_t* loader, void** model_handle) {
    if (!loader || !model_handle || !*model_handle) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(loader->mu);
    std::lock_guard<std::mutex> gl(g_models_mu);
    for (size_t i = 0; i < loader->held.size(); ++i)
        if (loader->held[i] == *model_handle) {
            loader->held.erase(loader->held.begin() + i);
            release_model((tk_mi355x_llm_model_t*)*model_handle);
            *model_handle = nullptr;
            return TK_SUCCESS;
        }
    return fail(TK_ERROR_INVALID_ARGUMENT, "handle was not produced by this loader");
}

struct tk_llm_runner_s {
    tk_mi355x_llm_model_t* model = nullptr;
    TkLlmBatcher* batcher = nullptr; /* shared with the other runners of the model */
    int slot = -1;                   /* this runner's sequence in the shared KV cache */
    int n_ctx = 0;
    int n_past = 0;
    int32_t pending = -1; /* token sampled from the last logits, not yet decoded */
    /* sampling: greedy unless tk_mi355x_llm_runner_set_sampling gave a temperature; the generator is keyed by (config.random_seed, number of
     * tokens this runner has sampled since its creation), so a runner's ids do not depend on which other runners share its passes */
    TkSampleRow samp{};
    const TkSampleRow* next_samp() { return samp.temp > 0.0f ? &samp : nullptr; }
    bool is_processing = false;
    std::string piece;
    std::string system_prompt;
    /* tool-call grammar (reference: tk_runner_lifecycle.c:59 loads it at create, tk_runner_streaming.c:44-48 arms it per generation) */
    TkGrammar grammar;
    bool has_grammar = false;
    bool grammar_on = false;
    TkGrammarState gstate;
    TkTokenTrie trie;
    bool trie_built = false;
    std::vector<uint32_t> mask;
    std::string tool_call_text; /* what the grammar has accepted so far: the tool call the host reads after the sentinel */
    /* allowed-token bits for the next sample, or nullptr when sampling is unconstrained */
    const uint32_t* next_mask() {
        if (!grammar_on) return nullptr;
        if (!trie_built) {
            std::vector<std::string> pieces((size_t)model->model.hp.vocab);
            for (int i = 0; i < (int)pieces.size(); ++i) pieces[i] = (i == model->tok.eos) ? std::string() : model->tok.piece(i);
            trie.build(pieces);
            trie_built = true;
        }
        gstate.mask(trie, model->tok.eos, &mask);
        return mask.data();
    }
};

/* the reference reads the grammar from a cwd-relative path; fall back to the built-in text of the same language */
static std::string load_tool_grammar_text() {
    const char* env = getenv("TK_TOOL_GRAMMAR");
    const char* paths[2] = {env, "src/ai_models/grammars/tool_call.gbnf"};
    for (const char* path : paths) {
        if (!path) continue;
        FILE* f = fopen(path, "rb");
        if (!f) continue;
        std::string text;
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
        fclose(f);
        return text;
    }
    return TK_DEFAULT_TOOL_CALL_GBNF;
}

const char* tk_mi355x_llm_runner_tool_call_text(tk_llm_runner_t* runner) { return runner ? runner->tool_call_text.c_str() : NULL; }

tk_error_code_t tk_mi355x_llm_runner_set_sampling(tk_llm_runner_t* runner, float temperature, int32_t top_k, float top_p, float min_p) {
    if (!runner || !(temperature >= 0.0f) || top_k < 0 || !(top_p > 0.0f) || top_p > 1.0f || !(min_p >= 0.0f) || min_p > 1.0f) return TK_ERROR_INVALID_ARGUMENT;
    /* the device sampler draws from at most TK_SAMPLE_MAX_K = 64 candidates: llama.cpp's default top_k (40) fits; a larger top_k is refused rather than
     * clamped silently, and top_k = 0 ("the whole vocabulary" in llama.cpp) means the 64 largest here (include/tk/tk_mi355x_ext.h) */
    if (top_k > TK_SAMPLE_MAX_K) return fail(TK_ERROR_INVALID_ARGUMENT, "top_k above 64: the sampler keeps at most 64 candidates");
    if (temperature > 0.0f && runner->model->model.hp.vocab > 65536) return fail(TK_ERROR_NOT_IMPLEMENTED, "stochastic sampling supports vocabularies of at most 65536 tokens");
    runner->samp.temp = temperature; runner->samp.top_k = top_k; runner->samp.top_p = top_p; runner->samp.min_p = min_p;
    return TK_SUCCESS;
}

tk_error_code_t tk_llm_runner_create(tk_llm_runner_t** out_runner, void* model_handle, const tk_llm_config_t* config) {
    if (!out_runner || !model_handle || !config) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_ptr<tk_llm_runner_s> r(new tk_llm_runner_s());
    r->model = (tk_mi355x_llm_model_t*)model_handle;
    r->n_ctx = config->context_size ? (int)config->context_size : 4096;
    if (config->system_prompt) r->system_prompt = config->system_prompt;
    r->samp.seed = config->random_seed;
    {   /* a sequence slot in a shared session of this context size; a new session when every slot is taken */
        std::lock_guard<std::mutex> lk(r->model->batch_mu);
        for (auto& b : r->model->batchers)
            if (b->n_ctx() == r->n_ctx && (r->slot = b->acquire_slot()) >= 0) { r->batcher = b.get(); break; }
        if (!r->batcher) {
            int slots = r->model->runner_slots;
            if (slots <= 0) { const char* e = getenv("TK_MI355X_RUNNER_SLOTS"); slots = e ? atoi(e) : 0; }
            if (slots <= 0) slots = 16;
            std::unique_ptr<TkLlmBatcher> b(new TkLlmBatcher());
            std::string err;
            if (!b->init(&r->model->model, slots, r->n_ctx, r->model->tok.eos, &err)) return fail(TK_ERROR_GPU_MEMORY, err);
            r->slot = b->acquire_slot();
            r->batcher = b.get();
            r->model->batchers.push_back(std::move(b));
        }
    }
    std::string gerr;
    r->has_grammar = r->grammar.parse(load_tool_grammar_text(), &gerr);
    if (!r->has_grammar) tk_error_set_detail("tool-call grammar rejected (%s): tool grammar disabled", gerr.c_str()); /* the reference logs and goes on */
    {
        std::lock_guard<std::mutex> gl(g_models_mu);
        r->model->refcount++; /* released in tk_llm_runner_destroy: unloading the model under a live runner cannot free its sessions */
    }
    *out_runner = r.release();
    return TK_SUCCESS;
}

void tk_llm_runner_destroy(tk_llm_runner_t** runner) {
    if (!runner || !*runner) return;
    if ((*runner)->batcher) (*runner)->batcher->release_slot((*runner)->slot); /* the shared session lives as long as the model */
    {
        std::lock_guard<std::mutex> gl(g_models_mu);
        release_model((*runner)->model);
    }
    delete *runner;
    *runner = nullptr;
}

/* feed `toks` at positions n_past.. ; the last one is sampled.  The rows join whatever passes the model's scheduler is forming. */
static tk_error_code_t feed(tk_llm_runner_s* r, const std::vector<int32_t>& toks) {
    if (toks.empty()) return TK_SUCCESS;
    if (r->n_past + (int)toks.size() >= r->n_ctx) return fail(TK_ERROR_INFERENCE_FAILED, "prompt exceeds the context window");
    int32_t am = -1;
    std::string err;
    if (!r->batcher->submit(r->slot, r->n_past, toks.data(), (int)toks.size(), r->next_mask(), &am, &err, r->next_samp())) return fail(TK_ERROR_INFERENCE_FAILED, err);
    if (r->samp.temp > 0.0f) r->samp.counter++;
    r->n_past += (int)toks.size();
    r->pending = am;
    return TK_SUCCESS;
}

tk_error_code_t tk_llm_runner_prepare_generation(tk_llm_runner_t* runner, const char* prompt, bool use_tool_grammar) {
    if (!runner || !prompt) return TK_ERROR_INVALID_ARGUMENT;
    /* greedy sampling constrained by the tool-call grammar: the arg max runs over the tokens the grammar allows (mask applied on
     * the device), the reference's llama_sampling_set_grammar(sctx, grammar or NULL) (tk_runner_streaming.c:44-48) */
    runner->grammar_on = use_tool_grammar && runner->has_grammar;
    if (runner->grammar_on) runner->gstate.init(&runner->grammar);
    runner->tool_call_text.clear();
    std::vector<int32_t> toks = runner->model->tok.encode(prompt, true);
    runner->n_past = 0; /* llama_kv_cache_clear: positions restart, stale cache rows are never attended */
    runner->pending = -1;
    tk_error_code_t rc = feed(runner, toks);
    if (rc != TK_SUCCESS) return rc;
    runner->is_processing = true;
    return TK_SUCCESS;
}

const char* tk_llm_runner_generate_next_token(tk_llm_runner_t* runner) {
    if (!runner || !runner->is_processing) return NULL;
    const int32_t id = runner->pending;
    if (id < 0 || id == runner->model->tok.eos) { runner->is_processing = false; return NULL; }
    if (runner->grammar_on) {
        /* llama_sampling_accept: advance the grammar by the sampled token; then the reference's "grammar rule completed" test
         * (tk_runner_streaming.c:69-75): the token is NOT decoded, generation pauses, the sentinel address is returned */
        const std::string pc = runner->model->tok.piece(id);
        if (!runner->gstate.accept(pc)) { /* cannot happen with the mask applied */
            tk_error_set_detail("sampled token %d is outside the grammar", id);
            runner->is_processing = false;
            return NULL;
        }
        runner->tool_call_text += pc;
        if (runner->gstate.complete()) {
            runner->is_processing = false;
            runner->grammar_on = false; /* the tool response and what follows it are free text */
            return (const char*)1;
        }
    }
    if (runner->n_past + 1 >= runner->n_ctx) { runner->is_processing = false; return NULL; }
    int32_t t = id, am = -1;
    std::string err;
    if (!runner->batcher->submit(runner->slot, runner->n_past, &t, 1, runner->next_mask(), &am, &err, runner->next_samp())) {
        tk_error_set_detail("%s", err.c_str());
        runner->is_processing = false;
        return NULL;
    }
    if (runner->samp.temp > 0.0f) runner->samp.counter++;
    runner->n_past++;
    runner->pending = am;
    runner->piece = runner->model->tok.piece(id);
    return runner->piece.c_str();
}

tk_error_code_t tk_llm_runner_add_tool_response(tk_llm_runner_t* runner, const char* tool_name, const char* tool_output) {
    if (!runner || !tool_name || !tool_output) return TK_ERROR_INVALID_ARGUMENT;
    std::string text = std::string("[TOOL_RESULT] name: \"") + tool_name + "\", output: " + tool_output + " [/TOOL_RESULT]";
    std::vector<int32_t> toks = runner->model->tok.encode(text, false);
    tk_error_code_t rc = feed(runner, toks);
    if (rc != TK_SUCCESS) return rc;
    runner->is_processing = true;
    return TK_SUCCESS;
}

tk_error_code_t tk_llm_runner_reset_context(tk_llm_runner_t* runner) {
    if (!runner) return TK_ERROR_INVALID_ARGUMENT;
    runner->n_past = 0;
    runner->pending = -1;
    runner->is_processing = false;
    return TK_SUCCESS;
}

void tk_llm_result_destroy(tk_llm_result_t** result) {
    if (!result || !*result) return;
    tk_llm_result_t* r = *result;
    if (r->type == TK_LLM_RESULT_TYPE_TEXT_RESPONSE) free(r->data.text_response);
    else if (r->type == TK_LLM_RESULT_TYPE_TOOL_CALL) { free(r->data.tool_call.name); free(r->data.tool_call.arguments_json); }
    free(r);
    *result = NULL;
}

} /* extern "C" */
