/*
 * tk_llm_engine.hip — host side of the LLM stream: weight residency, KV cache, pass
 * scheduling and hipGraph capture.  See tk_llm_engine.h for the reference call sites replaced.
 */
#include "tk_llm_engine.h"

#include <chrono>
#include "tk_lora.h"

#include <math.h>
#include <algorithm>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../common/tk_ggml_blocks.h"
#include "../nn/tk_gemm_tiled.h"
#include "../nn/tk_nn_kernels.h"

#define HIPQ(expr)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            char b__[256];                                                                      \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            error = b__;                                                                        \
            return false;                                                                       \
        }                                                                                       \
    } while (0)

/* ------------------------------------------------------------------ model ------------------ */

static int use_more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; }

int TkLlmModel::recipe_type(const TkLlmHParams& hp, int layer, int which) {
    if (layer < 0) return which == TK_T_OUTPUT ? TK_TYPE_Q6_K : (which == TK_T_TOKEN_EMBD ? TK_TYPE_Q4_K : TK_TYPE_F32);
    if (which == TK_L_ATTN_NORM || which == TK_L_FFN_NORM) return TK_TYPE_F32;
    if ((which == TK_L_V || which == TK_L_DOWN) && use_more_bits(layer, hp.n_layer)) return TK_TYPE_Q6_K;
    return TK_TYPE_Q4_K;
}

void TkLlmModel::shape(int layer, int which, int64_t* rows, int64_t* cols) const {
    const int64_t d = hp.d_model, kv = (int64_t)hp.n_kv_head * hp.head_dim, qd = (int64_t)hp.n_head * hp.head_dim;
    if (layer < 0) {
        if (which == TK_T_OUT_NORM) { *rows = 1; *cols = d; }
        else { *rows = hp.vocab; *cols = d; }
        return;
    }
    switch (which) {
        case TK_L_ATTN_NORM: case TK_L_FFN_NORM: *rows = 1; *cols = d; break;
        case TK_L_Q: *rows = qd; *cols = d; break;
        case TK_L_K: case TK_L_V: *rows = kv; *cols = d; break;
        case TK_L_O: *rows = d; *cols = qd; break;
        case TK_L_GATE: case TK_L_UP: *rows = hp.d_ff; *cols = d; break;
        default: *rows = d; *cols = hp.d_ff; break;
    }
}

TkDevTensor* TkLlmModel::slot(int layer, int which) {
    if (layer < 0) return which == TK_T_TOKEN_EMBD ? &token_embd : which == TK_T_OUT_NORM ? &out_norm : which == TK_T_OUTPUT ? &output : nullptr;
    if (layer >= (int)layers.size() || which < 0 || which >= TK_L_COUNT) return nullptr;
    TkLlmLayer& L = layers[layer];
    TkDevTensor* t[TK_L_COUNT] = {&L.attn_norm, &L.q, &L.k, &L.v, &L.o, &L.ffn_norm, &L.gate, &L.up, &L.down};
    return t[which];
}

bool TkLlmModel::init(const TkLlmHParams& h, int dev) {
    hp = h;
    device = dev;
    const int grp = h.n_kv_head > 0 ? h.n_head / h.n_kv_head : 0;
    const int64_t qd = (int64_t)h.n_head * h.head_dim, kvd = (int64_t)h.n_kv_head * h.head_dim;
    auto bad = [&](const char* why) { error = std::string("unsupported model geometry: ") + why; return false; };
    if (h.n_layer <= 0 || h.d_model % 256 || h.d_ff % 256 || qd % 256) return bad("d_model, d_ff and n_head*head_dim must be multiples of 256");
    if (h.n_kv_head <= 0 || h.n_head % h.n_kv_head || (grp != 1 && grp != 2 && grp != 4)) return bad("n_head / n_kv_head must be 1, 2 or 4");
    if ((grp * h.head_dim) % 256 || h.head_dim % 64) return bad("head_dim must be a multiple of 64 and (n_head/n_kv_head)*head_dim a multiple of 256");
    if (qd % 64 || kvd % 64 || h.d_ff % 64 || h.vocab % 64 || h.d_model % 64) return bad("row counts must be multiples of 64");
    if (h.ks_out != 1) return bad("ks_out must be 1");
    const int ksv[4] = {h.ks_qkv, h.ks_o, h.ks_gateup, h.ks_down};
    const int64_t kk[4] = {h.d_model, qd, h.d_model, h.d_ff};
    for (int i = 0; i < 4; ++i)
        if (ksv[i] <= 0 || ksv[i] > 8 || (kk[i] / 256) % ksv[i]) return bad("K-split must be in [1, 8] and divide K/256");
    HIPQ(hipSetDevice(device));
    layers.assign(h.n_layer, TkLlmLayer());
    return true;
}

TkLlmModel::~TkLlmModel() {
    (void)hipSetDevice(device);
    auto fr = [](TkDevTensor& t) { if (t.data) (void)hipFree(t.data); t.data = nullptr; };
    fr(token_embd); fr(out_norm); fr(output);
    for (auto& L : layers) { fr(L.attn_norm); fr(L.ffn_norm); fr(L.q); fr(L.k); fr(L.v); fr(L.o); fr(L.gate); fr(L.up); fr(L.down); }
}

/* dev_blocks: tensor in GGUF layout already in device memory */
bool TkLlmModel::install(TkDevTensor* t, int type, int64_t rows, int64_t cols, void* dev_blocks, hipStream_t s, int layer, int which) {
    if (t->data) { (void)hipFree(t->data); t->data = nullptr; }
    t->type = type; t->rows = rows; t->cols = cols;
    const bool is_matrix = rows > 1 && t != &token_embd;
    if (const TkLoraTensor* lt = (lora && is_matrix) ? lora->find(layer, which) : nullptr) {
        /* the reference's llama_model_apply_lora_from_file (tk_model_loader.c:259-270): W += (alpha / r) B A on the blocks as the file holds them */
        if (lt->n_out != rows || lt->k_in != cols) { error = "LoRA adapter does not fit this model (factor shapes against the base matrix)"; return false; }
        float *dA = nullptr, *dB = nullptr;
        HIPQ(hipMalloc((void**)&dA, lt->A.size() * 4));
        if (hipMalloc((void**)&dB, lt->B.size() * 4) != hipSuccess) { (void)hipFree(dA); error = "out of device memory (LoRA factors)"; return false; }
        bool ok = hipMemcpyAsync(dA, lt->A.data(), lt->A.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(dB, lt->B.data(), lt->B.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess;
        const bool taken = ok && tk_launch_lora_merge(type, dev_blocks, rows, cols, dA, dB, lt->r, lora->scale_of(*lt), s);
        const bool done = hipStreamSynchronize(s) == hipSuccess; /* the factors' host copies and dA / dB live until here */
        (void)hipFree(dA);
        (void)hipFree(dB);
        if (ok && !taken) { error = "LoRA merge needs a Q4_K, Q6_K or F16 matrix with columns % 256 == 0"; return false; }
        if (!ok || !done) { error = "LoRA merge failed on the device"; return false; }
        lora_merged++;
    }
    if (type == TK_TYPE_F32) {
        t->bytes = (size_t)rows * cols * 4;
        HIPQ(hipMalloc((void**)&t->data, t->bytes));
        HIPQ(hipMemcpyAsync(t->data, dev_blocks, t->bytes, hipMemcpyDeviceToDevice, s));
        return true;
    }
    if (type == TK_TYPE_F16) { /* fp16 checkpoints: matrices become the tiled GEMM's weight tiles (csrc/nn/tk_gemm_tiled.h), the embedding stays row-major */
        if (cols % 32 || (is_matrix && rows % 16)) { error = "f16 matrices need rows % 16 == 0 and columns % 32 == 0"; return false; }
        t->bytes = (size_t)rows * cols * 2;
        HIPQ(hipMalloc((void**)&t->data, t->bytes));
        if (is_matrix) {
            tk_launch_tile_weights(dev_blocks, 2, rows, cols, t->data, s);
            HIPQ(hipGetLastError());
            has_f16 = true;
        } else {
            HIPQ(hipMemcpyAsync(t->data, dev_blocks, t->bytes, hipMemcpyDeviceToDevice, s));
        }
        return true;
    }
    if (type != TK_TYPE_Q4_K && type != TK_TYPE_Q6_K) { error = "unsupported tensor type (want F32, F16, Q4_K or Q6_K)"; return false; }
    t->bytes = (size_t)rows * cols / 256 * tk_type_block_bytes(type);
    HIPQ(hipMalloc((void**)&t->data, t->bytes));
    if (!is_matrix) { /* token_embd stays in GGUF layout: one row is gathered per token */
        if (type != TK_TYPE_Q4_K) { error = "token_embd must be Q4_K or F16"; return false; }
        HIPQ(hipMemcpyAsync(t->data, dev_blocks, t->bytes, hipMemcpyDeviceToDevice, s));
        return true;
    }
    tk_launch_repack(type, dev_blocks, rows, cols, t->data, s);
    HIPQ(hipGetLastError());
    return true;
}

bool TkLlmModel::set_tensor(int layer, int which, int type, const void* host_blocks, size_t nbytes) {
    HIPQ(hipSetDevice(device));
    TkDevTensor* t = slot(layer, which);
    if (!t) { error = "no such tensor"; return false; }
    int64_t rows, cols;
    shape(layer, which, &rows, &cols);
    size_t expect = (size_t)rows * cols / tk_type_block_elems(type) * tk_type_block_bytes(type);
    if (expect != nbytes) { error = "tensor byte size does not match the model geometry"; return false; }
    void* tmp = nullptr;
    HIPQ(hipMalloc(&tmp, nbytes));
    HIPQ(hipMemcpy(tmp, host_blocks, nbytes, hipMemcpyHostToDevice));
    bool ok = install(t, type, rows, cols, tmp, nullptr, layer, which);
    (void)hipDeviceSynchronize();
    (void)hipFree(tmp);
    return ok;
}

bool TkLlmModel::fill_synthetic(uint64_t seed, bool f16) {
    HIPQ(hipSetDevice(device));
    auto type_of = [&](int l, int w) {
        const int t = recipe_type(hp, l, w);
        return (f16 && t != TK_TYPE_F32) ? (int)TK_TYPE_F16 : t; /* fp16 checkpoint: every matrix and the embedding f16, norms f32 */
    };
    /* scratch big enough for the largest tensor in GGUF layout */
    size_t maxb = 0;
    for (int l = -1; l < hp.n_layer; ++l)
        for (int w = 0; w < (l < 0 ? 3 : (int)TK_L_COUNT); ++w) {
            int64_t r, c;
            shape(l, w, &r, &c);
            int type = type_of(l, w);
            size_t b = (size_t)r * c / tk_type_block_elems(type) * tk_type_block_bytes(type);
            maxb = b > maxb ? b : maxb;
        }
    void* tmp = nullptr;
    HIPQ(hipMalloc(&tmp, maxb));
    bool ok = true;
    for (int l = -1; l < hp.n_layer && ok; ++l)
        for (int w = 0; w < (l < 0 ? 3 : (int)TK_L_COUNT) && ok; ++w) {
            int64_t r, c;
            shape(l, w, &r, &c);
            int type = type_of(l, w);
            uint64_t tid = l < 0 ? (uint64_t)w : (uint64_t)(16 + l * 16 + w);
            if (type == TK_TYPE_F32) tk_launch_synth_f32(seed, tid, r * c, (float*)tmp, nullptr);
            else if (type == TK_TYPE_F16) tk_launch_synth_f16(seed, tid, r * c, 0.02f, (uint16_t*)tmp, nullptr);
            else tk_launch_synth_blocks(type, seed, tid, r * c / 256, 0.02f, tmp, nullptr);
            ok = install(slot(l, w), type, r, c, tmp, nullptr, l, w);
            if (ok && hipStreamSynchronize(nullptr) != hipSuccess) { error = "synthetic weight generation failed"; ok = false; }
        }
    (void)hipFree(tmp);
    return ok;
}

bool TkLlmModel::ready() const {
    if (!token_embd.data || !out_norm.data || !output.data) return false;
    for (const auto& L : layers) {
        if (!L.attn_norm.data || !L.ffn_norm.data || !L.q.data || !L.k.data || !L.v.data || !L.o.data || !L.gate.data || !L.up.data || !L.down.data)
            return false;
        /* the tensors of one fused launch (q|k|v, gate|up) are all k-quants or all f16 */
        const bool f_qkv = L.q.type == TK_TYPE_F16, f_gu = L.gate.type == TK_TYPE_F16;
        if ((L.k.type == TK_TYPE_F16) != f_qkv || (L.v.type == TK_TYPE_F16) != f_qkv || (L.up.type == TK_TYPE_F16) != f_gu) return false;
    }
    return true;
}

/* ------------------------------------------------------------------ session ---------------- */

static bool alloc_act(TkActQ8* a, int K, bool want_f16, std::string& error) {
    a->aq_ts = TK_AQ_BYTES(K); a->ad_ts = TK_AD_FLOATS(K); a->abs_ts = TK_ABS_BYTES(K);
    a->af = nullptr; a->af_ts = (size_t)K * TK_ROW_SLOTS;
    if (want_f16 && hipMalloc((void**)&a->af, (size_t)TK_MAX_ROWS * K * 4) != hipSuccess) { error = "out of device memory (f16 activation buffers)"; return false; }
    if (a->af) (void)hipMemset(a->af, 0, (size_t)TK_MAX_ROWS * K * 4);
    if (hipMalloc((void**)&a->aq, TK_MAX_TILES * a->aq_ts) != hipSuccess || hipMalloc((void**)&a->ad, TK_MAX_TILES * a->ad_ts * 4) != hipSuccess ||
        hipMalloc((void**)&a->abs, TK_MAX_TILES * a->abs_ts) != hipSuccess || hipMalloc((void**)&a->abs16, TK_MAX_TILES * a->abs_ts * 2) != hipSuccess) {
        error = "out of device memory (activation buffers)";
        return false;
    }
    (void)hipMemset(a->aq, 0, TK_MAX_TILES * a->aq_ts);
    (void)hipMemset(a->ad, 0, TK_MAX_TILES * a->ad_ts * 4);
    (void)hipMemset(a->abs, 0, TK_MAX_TILES * a->abs_ts);
    (void)hipMemset(a->abs16, 0, TK_MAX_TILES * a->abs_ts * 2);
    return true;
}
static void free_act(TkActQ8* a) {
    if (a->aq) (void)hipFree(a->aq);
    if (a->ad) (void)hipFree(a->ad);
    if (a->abs) (void)hipFree(a->abs);
    if (a->abs16) (void)hipFree(a->abs16);
    if (a->af) (void)hipFree(a->af);
    *a = TkActQ8{};
}

bool TkLlmSession::init(TkLlmModel* m, int mseq, int mctx) {
    model = m;
    max_seq = mseq;
    max_ctx = mctx;
    if (!m || !m->ready()) { error = "model has missing tensors"; return false; }
    if (mseq <= 0 || mctx <= 0) { error = "max_seq and max_ctx must be positive"; return false; }
    const TkLlmHParams& h = m->hp;
    if (tk_attention_lds_bytes(h.n_head / h.n_kv_head, h.head_dim, mctx, 64) > 160 * 1024) {
        error = "max_ctx too large: the attention kernel keeps one score row per query head of a KV group in LDS (about 9000 positions for 4 x 128)";
        return false;
    }
    HIPQ(hipSetDevice(m->device));
    HIPQ(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    tk_attention_note_session(m->device, +1);
    counted_ = true;
    const int QD = h.n_head * h.head_dim, KVD = h.n_kv_head * h.head_dim, half = h.head_dim / 2;
    size_t kv = (size_t)h.n_layer * mseq * mctx * KVD;
    HIPQ(hipMalloc((void**)&kcache, kv * 2));
    HIPQ(hipMalloc((void**)&vcache, kv * 2));
    HIPQ(hipMemset(kcache, 0, kv * 2));
    HIPQ(hipMemset(vcache, 0, kv * 2));
    HIPQ(hipMalloc((void**)&x, (size_t)TK_MAX_ROWS * h.d_model * 4));
    HIPQ(hipMalloc((void**)&qbuf, (size_t)TK_MAX_ROWS * QD * 4));
    size_t pmax = (size_t)h.ks_qkv * (QD + 2 * KVD);
    pmax = std::max(pmax, (size_t)h.ks_o * h.d_model);
    pmax = std::max(pmax, (size_t)h.ks_gateup * 2 * h.d_ff);
    pmax = std::max(pmax, (size_t)h.ks_down * h.d_model);
    HIPQ(hipMalloc((void**)&partial, pmax * TK_MAX_ROWS * 4));
    /* twins for passes whose mat-vec launches run their producer themselves (enqueue_range): such a launch reads one (x, slab) pair while
     * its workgroups write the other */
    HIPQ(hipMalloc((void**)&x2, (size_t)TK_MAX_ROWS * h.d_model * 4));
    HIPQ(hipMalloc((void**)&partial2, pmax * TK_MAX_ROWS * 4));
    HIPQ(hipMalloc((void**)&logits, (size_t)TK_MAX_ROWS * h.vocab * 4));
    if (!alloc_act(&act_d, h.d_model, m->has_f16, error) || !alloc_act(&act_qd, QD, m->has_f16, error) || !alloc_act(&act_ff, h.d_ff, m->has_f16, error)) return false;
    HIPQ(hipMalloc((void**)&d_seq, TK_MAX_ROWS * 4));
    HIPQ(hipMalloc((void**)&d_pos, TK_MAX_ROWS * 4));
    HIPQ(hipMalloc((void**)&d_tok, TK_MAX_ROWS * 4));
    HIPQ(hipMalloc((void**)&d_nsteps, TK_MAX_ROWS * 4));
    HIPQ(hipMemset(d_nsteps, 0, TK_MAX_ROWS * 4));
    hist_cap = mctx;
    HIPQ(hipMalloc((void**)&d_hist, (size_t)hist_cap * TK_MAX_ROWS * 4));
    HIPQ(hipMalloc((void**)&d_tiles, (size_t)(1 + TK_MAX_ROWS) * 4));
    HIPQ(hipMemset(d_tiles, 0, (size_t)(1 + TK_MAX_ROWS) * 4));
    if (mctx > TK_LONG_ATT_MIN_POS && tk_attention_long_applies(1, h.n_head, h.n_kv_head, h.head_dim))
        HIPQ(hipMalloc((void**)&d_scores, tk_attention_long_scratch_floats(h.n_head, h.head_dim, mctx) * sizeof(float)));
    HIPQ(hipMalloc((void**)&d_mask, (size_t)TK_MAX_ROWS * (((size_t)h.vocab + 31) / 32) * 4));
    HIPQ(hipMalloc((void**)&d_mask_row, TK_MAX_ROWS * 4));
    HIPQ(hipMemset(d_mask_row, 0xFF, TK_MAX_ROWS * 4));
    mask_rows_dirty = false;
    HIPQ(hipMalloc((void**)&d_samp, TK_MAX_ROWS * sizeof(TkSampleRow)));
    HIPQ(hipMemset(d_samp, 0, TK_MAX_ROWS * sizeof(TkSampleRow))); /* temp 0: every row samples greedily */
    samp_dirty = false;
    /* RoPE table, double precision on the host (same formula as the oracle) */
    std::vector<float> cs((size_t)mctx * half), sn((size_t)mctx * half);
    for (int p = 0; p < mctx; ++p)
        for (int i = 0; i < half; ++i) {
            double theta = pow((double)h.rope_theta, -2.0 * i / (double)h.head_dim);
            double a = (double)p * theta;
            cs[(size_t)p * half + i] = (float)cos(a);
            sn[(size_t)p * half + i] = (float)sin(a);
        }
    HIPQ(hipMalloc((void**)&rope_cos, cs.size() * 4));
    HIPQ(hipMalloc((void**)&rope_sin, sn.size() * 4));
    HIPQ(hipMemcpy(rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    /* kernels that want more than the default 64 KiB of dynamic LDS: a per-device opt-in */
    if (const char* e = tk_llm_prepare_device(m->device)) { error = std::string("LDS opt-in failed: ") + e; return false; }
    if (m->has_f16 && !tk_gemm_tiled_prepare_device()) { error = "LDS opt-in of the tiled GEMM failed"; return false; }
    HIPQ(hipDeviceSynchronize());
    return true;
}

TkLlmSession::~TkLlmSession() {
    if (model && counted_) tk_attention_note_session(model->device, -1);
    if (model) (void)hipSetDevice(model->device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& v : graph_exec) for (auto& g : v) if (g) (void)hipGraphExecDestroy(g);
    for (auto& v : graph_prefill) for (auto& g : v) if (g) (void)hipGraphExecDestroy(g);
    for (auto& v : graph_head_nf) for (auto& g : v) if (g) (void)hipGraphExecDestroy(g);
    void* ptrs[] = {kcache, vcache, x, x2, qbuf, partial, partial2, logits, rope_cos, rope_sin, d_seq, d_pos, d_tok, d_nsteps, d_hist, d_mask, d_mask_row, d_samp, d_tab, d_tiles, d_scores};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    free_act(&act_d); free_act(&act_qd); free_act(&act_ff);
    if (stream) (void)hipStreamDestroy(stream);
}

bool TkLlmSession::reset() {
    HIPQ(hipSetDevice(model->device));
    HIPQ(hipMemsetAsync(d_nsteps, 0, TK_MAX_ROWS * 4, stream));
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

bool TkLlmSession::kv_write(int layer, int seq, int pos0, int n_pos, const uint16_t* k, const uint16_t* v) {
    const TkLlmHParams& h = model->hp;
    if (layer < 0 || layer >= h.n_layer || seq < 0 || seq >= max_seq || pos0 < 0 || n_pos <= 0 || pos0 + n_pos > max_ctx || !k || !v) { error = "kv_write: (layer, sequence, positions) outside the cache"; return false; }
    HIPQ(hipSetDevice(model->device));
    const size_t hd = (size_t)h.head_dim * 2; /* bytes of one cache row */
    for (int kvh = 0; kvh < h.n_kv_head; ++kvh) { /* host pitch = one position of all heads; device run of one head = consecutive positions */
        const size_t dst = ((((size_t)layer * max_seq + seq) * h.n_kv_head + kvh) * max_ctx + pos0) * h.head_dim;
        HIPQ(hipMemcpy2DAsync(kcache + dst, hd, k + (size_t)kvh * h.head_dim, hd * h.n_kv_head, hd, (size_t)n_pos, hipMemcpyHostToDevice, stream));
        HIPQ(hipMemcpy2DAsync(vcache + dst, hd, v + (size_t)kvh * h.head_dim, hd * h.n_kv_head, hd, (size_t)n_pos, hipMemcpyHostToDevice, stream));
    }
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

bool TkLlmSession::kv_read(int layer, int seq, int pos0, int n_pos, uint16_t* k, uint16_t* v) {
    const TkLlmHParams& h = model->hp;
    if (layer < 0 || layer >= h.n_layer || seq < 0 || seq >= max_seq || pos0 < 0 || n_pos <= 0 || pos0 + n_pos > max_ctx || !k || !v) { error = "kv_read: (layer, sequence, positions) outside the cache"; return false; }
    HIPQ(hipSetDevice(model->device));
    const size_t hd = (size_t)h.head_dim * 2;
    for (int kvh = 0; kvh < h.n_kv_head; ++kvh) {
        const size_t src = ((((size_t)layer * max_seq + seq) * h.n_kv_head + kvh) * max_ctx + pos0) * h.head_dim;
        HIPQ(hipMemcpy2DAsync(k + (size_t)kvh * h.head_dim, hd * h.n_kv_head, kcache + src, hd, hd, (size_t)n_pos, hipMemcpyDeviceToHost, stream));
        HIPQ(hipMemcpy2DAsync(v + (size_t)kvh * h.head_dim, hd * h.n_kv_head, vcache + src, hd, hd, (size_t)n_pos, hipMemcpyDeviceToHost, stream));
    }
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

static void set_act(TkGemvArgs& a, const TkActQ8& q) {
    a.aq = q.aq; a.ad = q.ad; a.abs = q.abs; a.abs16 = q.abs16; a.aq_ts = q.aq_ts; a.ad_ts = q.ad_ts; a.abs_ts = q.abs_ts;
}

static TkGemvSeg seg_of(const TkDevTensor& t) { return TkGemvSeg{t.data, t.type, (int)(t.rows / TK_TILE_ROWS)}; }

void TkLlmSession::enqueue_pass(int nrows, bool lm_head, bool fused_attn) { enqueue_range(nrows, 0, model->hp.n_layer, true, false, lm_head, fused_attn); }

/* layers [l0, l1) of one pass.  embed: the residual stream starts from the token embeddings (first pipeline stage), otherwise x
 * already holds it.  fold_out: finish the last layer's residual update so x is the complete stream (it leaves this GPU).
 * head: final norm + lm_head + arg max (last stage). */
/* one matmul group of a pass: the K-quant tensors of a launch go to the W4A8 kernels (K-split partial slabs, canonical plan `ks`); an f16
 * group goes, tensor by tensor, through the exact fp32 GEMM on the f16-rounded activations (one chain over K: a single slab).  Returns
 * the number of partial slabs the consumers must add. */
int TkLlmSession::enqueue_matmul(const TkDevTensor* const* t, int nseg, int K, int ks, int n_total, const TkActQ8& act, float* out, int nrows) {
    hipStream_t s = stream;
    if (t[0]->type == TK_TYPE_F16) {
        TkTiledGemm f{};
        for (int i = 0; i < nseg; ++i) { f.tiles[i] = t[i]->data; f.row_tiles[i] = (int)(t[i]->rows / TK_TILE_ROWS); }
        f.nseg = nseg; f.wbytes = 2; f.K = K; f.ks = ks; f.ldc = n_total; f.n_valid = n_total; f.nrows = nrows; f.slab_rows = TK_MAX_ROWS;
        f.a_img = act.af; f.a_ts = act.af_ts; f.out = out;
        /* shapes were validated when the model was installed (K a multiple of 256 ks); a refusal here would leave the slab unwritten, so it is
         * recorded and fails the pass (forward / decode / capture_pass check launch_error) instead of producing garbage silently */
        if (!tk_launch_gemm_tiled(f, s) && launch_error.empty()) launch_error = "tiled GEMM launch refused (shape outside what the f16 path supports)";
        return ks;
    }
    TkGemvArgs a{};
    for (int i = 0; i < nseg; ++i) a.seg[i] = seg_of(*t[i]);
    a.nseg = nseg; a.K = K; a.ks = ks; a.n_total = n_total; a.nrows = nrows;
    set_act(a, act); a.out = out;
    tk_launch_gemv(a, s);
    return ks;
}

static bool producer_fusion_enabled() {
    /* TK_MI355X_NO_FUSE=1: the norm / SwiGLU kernels stay launches of their own at every row count (A/B timing, bisecting) */
    const char* nf = getenv("TK_MI355X_NO_FUSE");
    return !(nf && nf[0] == '1');
}

void TkLlmSession::enqueue_range(int nrows, int l0, int l1, bool embed, bool fold_out, bool lm_head, bool fused_attn) {
    const TkLlmHParams& h = model->hp;
    const int D = h.d_model, QD = h.n_head * h.head_dim, KVD = h.n_kv_head * h.head_dim, FF = h.d_ff;
    hipStream_t s = stream;
    if (embed) tk_launch_embed(model->token_embd.data, model->token_embd.type, D, d_tok, nrows, x, s);
    /* One or two rows (the reference's own use: one runner, one token per step): a launch boundary (~1.5 us) plus a norm or SwiGLU kernel
     * that is one load latency long (5 - 6 us) costs more than that kernel's work repeated by every workgroup of the mat-vec launch that
     * consumes it — so q|k|v, gate|up and the logits matrix form their normalised int8 input themselves (TkGemvArgs::fuse = 1), the down
     * projection its SwiGLU input (fuse = 2): 8 launches per layer become 5.  Same arithmetic, value for value.  Such a launch reads
     * (x, slabs) of one buffer pair while workgroup 0 writes the updated stream into the other and all write their slabs there:
     *   q|k|v: x, partial -> x2, partial2;  attention reads partial2;  o -> partial;  gate|up: x2, partial -> x, partial2;
     *   down: partial2 -> partial — at every layer boundary x and `partial` hold what the unfused path leaves there. */
    bool kq = model->output.type != TK_TYPE_F16;
    for (int l = l0; l < l1 && kq; ++l) {
        const TkLlmLayer& L = model->layers[l];
        kq = L.q.type != TK_TYPE_F16 && L.k.type != TK_TYPE_F16 && L.v.type != TK_TYPE_F16 && L.o.type != TK_TYPE_F16 && L.gate.type != TK_TYPE_F16 &&
             L.up.type != TK_TYPE_F16 && L.down.type != TK_TYPE_F16;
    }
    const bool fuse = kq && producer_fusion_enabled() && tk_gemv_fuses_producer(nrows, D, h.ks_qkv, h.ks_down) &&
                      tk_gemv_fuses_producer(nrows, D, h.ks_gateup, h.ks_o) && tk_gemv_fuses_producer(nrows, FF, h.ks_down, h.ks_gateup) &&
                      tk_gemv_fuses_producer(nrows, D, 1, h.ks_down);
    auto fused_gemv = [&](const TkDevTensor* const* t, int nseg, int K, int ks, int n_total, float* out, int mode, const float* xin, float* xout,
                          const float* slab, int slab_ks, int slab_pitch, const float* w) {
        TkGemvArgs a{};
        for (int i = 0; i < nseg; ++i) a.seg[i] = seg_of(*t[i]);
        a.nseg = nseg; a.K = K; a.ks = ks; a.n_total = n_total; a.nrows = nrows; a.out = out;
        a.fuse = mode; a.fx_in = xin; a.fx_out = xout; a.fslab = slab; a.fks = slab_ks; a.fn_total = slab_pitch; a.fw = w; a.feps = h.rms_eps;
        tk_launch_gemv(a, s);
        return ks;
    };
    /* passes that hold several positions of a sequence (prompt chunks): 16 rows of a sequence per attention workgroup on the fp32 matrix pipe
     * (k_attention_prefill, bit-identical to k_attention); the row -> tile table is built once per pass from the sequence ids on the device */
    const bool tiled_attn = !fused_attn && l1 > l0 && tiled_pass && tk_attention_prefill_applies(h.n_head, h.n_kv_head, h.head_dim);
    /* few rows over a long context: the fused kernels would walk it as one latency chain per pair of heads */
    const bool long_attn = fused_attn && l1 > l0 && long_pass && d_scores && tk_attention_long_applies(nrows, h.n_head, h.n_kv_head, h.head_dim);
    if (tiled_attn) tk_launch_att_tiles(d_seq, nrows, d_tiles, s);
    int ks_res = 1; /* slabs of the pending residual update (the previous layer's down projection) */
    for (int l = l0; l < l1; ++l) {
        const TkLlmLayer& L = model->layers[l];
        const TkDevTensor* qkv[3] = {&L.q, &L.k, &L.v};
        const TkDevTensor* ot[1] = {&L.o};
        const TkDevTensor* gu[2] = {&L.gate, &L.up};
        const TkDevTensor* dn[1] = {&L.down};
        if (fuse) {
            const int ks_qkv = fused_gemv(qkv, 3, D, h.ks_qkv, QD + 2 * KVD, partial2, 1, x, x2, l == l0 ? nullptr : partial, ks_res, D, (const float*)L.attn_norm.data);
            if (!fused_attn)
                tk_launch_qkv_rope_append(partial2, ks_qkv, QD + 2 * KVD, h.n_head, h.n_kv_head, h.head_dim, rope_cos, rope_sin, d_seq, d_pos, nrows,
                                          qbuf, kcache, vcache, l, max_seq, max_ctx, s);
            if (long_attn) tk_launch_attention_long(partial2, ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head, h.head_dim, l, max_seq, max_ctx, d_scores, act_qd, s);
            else if (tiled_attn) tk_launch_attention_prefill(qbuf, kcache, vcache, d_seq, d_pos, d_tiles, nrows, h.n_head, h.n_kv_head, h.head_dim, l, max_seq, max_ctx, act_qd, s);
            else tk_launch_attention(qbuf, partial2, ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head,
                                     h.head_dim, l, max_seq, max_ctx, act_qd, fused_attn, s);
            const int ks_o = enqueue_matmul(ot, 1, QD, h.ks_o, D, act_qd, partial, nrows);
            const int ks_gu = fused_gemv(gu, 2, D, h.ks_gateup, 2 * FF, partial2, 1, x2, x, partial, ks_o, D, (const float*)L.ffn_norm.data);
            ks_res = fused_gemv(dn, 1, FF, h.ks_down, D, partial, 2, nullptr, nullptr, partial2, ks_gu, 2 * FF, nullptr);
            continue;
        }
        tk_launch_rmsnorm_q8(x, l == l0 ? nullptr : partial, ks_res, D, (const float*)L.attn_norm.data, h.rms_eps, D, nrows, act_d, s);
        const int ks_qkv = enqueue_matmul(qkv, 3, D, h.ks_qkv, QD + 2 * KVD, act_d, partial, nrows);
        if (!fused_attn)
            tk_launch_qkv_rope_append(partial, ks_qkv, QD + 2 * KVD, h.n_head, h.n_kv_head, h.head_dim, rope_cos, rope_sin, d_seq, d_pos, nrows,
                                      qbuf, kcache, vcache, l, max_seq, max_ctx, s);
        if (long_attn) tk_launch_attention_long(partial, ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head, h.head_dim, l, max_seq, max_ctx, d_scores, act_qd, s);
        else if (tiled_attn) tk_launch_attention_prefill(qbuf, kcache, vcache, d_seq, d_pos, d_tiles, nrows, h.n_head, h.n_kv_head, h.head_dim, l, max_seq, max_ctx, act_qd, s);
        else tk_launch_attention(qbuf, partial, ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head,
                                 h.head_dim, l, max_seq, max_ctx, act_qd, fused_attn, s);
        const int ks_o = enqueue_matmul(ot, 1, QD, h.ks_o, D, act_qd, partial, nrows);
        tk_launch_rmsnorm_q8(x, partial, ks_o, D, (const float*)L.ffn_norm.data, h.rms_eps, D, nrows, act_d, s);
        if (tk_gemv_fuses_swiglu(nrows, h.ks_gateup, L.gate.type, L.up.type)) { /* wide pass: SwiGLU in the launch's epilogue, `partial` holds h [rows][FF] */
            TkGemvArgs a{};
            a.seg[0] = seg_of(L.gate); a.seg[1] = seg_of(L.up);
            a.nseg = 2; a.K = D; a.ks = 1; a.n_total = FF; a.nrows = nrows; a.swiglu = 1;
            set_act(a, act_d); a.out = partial;
            tk_launch_gemv(a, s);
            tk_launch_quant_q8(partial, FF, nrows, act_ff, s);
        } else {
            const int ks_gu = enqueue_matmul(gu, 2, D, h.ks_gateup, 2 * FF, act_d, partial, nrows);
            tk_launch_swiglu_q8(partial, ks_gu, FF, nrows, act_ff, s);
        }
        ks_res = enqueue_matmul(dn, 1, FF, h.ks_down, D, act_ff, partial, nrows);
    }
    last_ks_res = ks_res;
    if (!lm_head) { /* prompt rows whose logits nobody reads (K/V are already appended), or a pipeline stage that hands x on */
        if (fold_out && l1 > l0) tk_launch_residual_fold(x, partial, ks_res, D, D, nrows, s);
        return;
    }
    const TkDevTensor* lm[1] = {&model->output};
    if (fuse) (void)fused_gemv(lm, 1, D, 1, h.vocab, logits, 1, x, x2, l1 > l0 ? partial : nullptr, ks_res, D, (const float*)model->out_norm.data);
    else {
        tk_launch_rmsnorm_q8(x, l1 > l0 ? partial : nullptr, ks_res, D, (const float*)model->out_norm.data, h.rms_eps, D, nrows, act_d, s);
        (void)enqueue_matmul(lm, 1, D, 1, h.vocab, act_d, logits, nrows);
    }
    tk_launch_argmax(logits, h.vocab, nrows, d_mask, d_mask_row, d_samp, d_tok, d_pos, d_nsteps, d_hist, TK_MAX_ROWS, s);
}

static bool graphs_enabled() {
    /* TK_MI355X_NO_GRAPH=1: eager launches, for profilers that cannot follow hipGraphLaunch (see DESIGN.md "Profiling") */
    const char* ng = getenv("TK_MI355X_NO_GRAPH");
    return !(ng && ng[0] == '1');
}

/* one capture at a time per process; relaxed mode: other host threads (detector / ASR / VAD streams) keep calling allocation and copy
 * APIs while this stream records */
static std::mutex g_capture_mu;

#ifndef TK_TILED_ATT_MIN_POS
#define TK_TILED_ATT_MIN_POS 128
#endif
void TkLlmSession::choose_attention(const int32_t* pos, int nrows) {
    int top = 0;
    for (int r = 0; r < nrows; ++r) top = pos[r] > top ? pos[r] : top;
    choose_attention_top(top, nrows);
}
void TkLlmSession::choose_attention_top(int top, int nrows) {
    tiled_pass = top >= TK_TILED_ATT_MIN_POS;
    long_pass = d_scores != nullptr && nrows <= TK_LONG_ATT_MAX_ROWS && top >= tk_long_att_min_pos(nrows);
}

bool TkLlmSession::capture_pass(hipGraphExec_t* slot, int nrows, bool lm_head, bool fused_attn) {
    if (*slot) return true;
    const auto t_cap = std::chrono::steady_clock::now();
    struct Tally { TkLlmSession* s; std::chrono::steady_clock::time_point t0; ~Tally() { s->n_captures++; s->capture_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } tally{this, t_cap};
    std::lock_guard<std::mutex> lk(g_capture_mu);
    hipGraph_t g = nullptr;
    HIPQ(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
    launch_error.clear();
    enqueue_pass(nrows, lm_head, fused_attn);
    /* from here on the stream must leave capture mode and the graph must be freed whatever fails */
    const hipError_t e_end = hipStreamEndCapture(stream, &g);
    hipError_t e_inst = hipSuccess;
    if (e_end == hipSuccess && launch_error.empty()) e_inst = hipGraphInstantiate(slot, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (e_end != hipSuccess || e_inst != hipSuccess || !launch_error.empty()) {
        if (*slot && e_inst != hipSuccess) { (void)hipGraphExecDestroy(*slot); }
        *slot = nullptr;
        (void)hipGetLastError();
        error = !launch_error.empty() ? launch_error
                : std::string("graph capture of a pass failed: ") + hipGetErrorString(e_end != hipSuccess ? e_end : e_inst);
        return false;
    }
    return true;
}

bool TkLlmSession::forward(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, float* logits_host, int32_t* argmax_host,
                           bool lm_head, const uint32_t* const* row_masks, const TkSampleRow* row_samp) {
    if (nrows <= 0 || nrows > TK_MAX_ROWS) { error = "nrows must be in [1,256]"; return false; }
    for (int r = 0; r < nrows; ++r) {
        if (seq[r] < 0 || seq[r] >= max_seq || pos[r] < 0 || pos[r] >= max_ctx || tok[r] < 0 || tok[r] >= model->hp.vocab) {
            error = "row out of range (sequence id, position or token id)";
            return false;
        }
    }
    HIPQ(hipSetDevice(model->device));
    HIPQ(hipMemcpyAsync(d_seq, seq, nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemcpyAsync(d_pos, pos, nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemcpyAsync(d_tok, tok, nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemsetAsync(d_nsteps, 0, TK_MAX_ROWS * 4, stream));
    choose_attention(pos, nrows);
    bool distinct = true; /* every sequence at most once in the pass -> rope/append can be fused into attention */
    for (int a = 0; a < nrows && distinct; ++a)
        for (int b = a + 1; b < nrows; ++b)
            if (seq[a] == seq[b]) { distinct = false; break; }
    if (lm_head) { /* per-row sampling masks: the masks of the constrained rows, compacted, + the row -> mask table the arg max kernel reads */
        int32_t rowtab[TK_MAX_ROWS];
        int nmask = 0;
        const size_t words = ((size_t)model->hp.vocab + 31) / 32;
        for (int r = 0; r < TK_MAX_ROWS; ++r) rowtab[r] = -1;
        for (int r = 0; r < nrows; ++r) {
            if (row_masks && row_masks[r]) {
                HIPQ(hipMemcpyAsync(d_mask + (size_t)nmask * words, row_masks[r], words * 4, hipMemcpyHostToDevice, stream));
                rowtab[r] = nmask++;
            }
        }
        if (nmask > 0 || mask_rows_dirty) {
            HIPQ(hipMemcpyAsync(d_mask_row, rowtab, TK_MAX_ROWS * 4, hipMemcpyHostToDevice, stream)); /* whole table (a wider pass may have left entries); pageable source: staged before the call returns */
            mask_rows_dirty = nmask > 0;
        }
        /* per-row sampling state (temp <= 0: greedy): data of the pass like the masks, so sampled and greedy rows share passes and graphs */
        bool any_samp = false;
        for (int r = 0; r < nrows && row_samp; ++r) any_samp = any_samp || row_samp[r].temp > 0.0f;
        if (any_samp) {
            if (model->hp.vocab > 65536) { error = "stochastic sampling supports vocabularies of at most 65536 tokens"; return false; }
            TkSampleRow tab[TK_MAX_ROWS] = {};
            for (int r = 0; r < nrows; ++r) tab[r] = row_samp[r];
            HIPQ(hipMemcpyAsync(d_samp, tab, sizeof tab, hipMemcpyHostToDevice, stream));
            samp_dirty = true;
        } else if (samp_dirty) {
            HIPQ(hipMemsetAsync(d_samp, 0, TK_MAX_ROWS * sizeof(TkSampleRow), stream));
            samp_dirty = false;
        }
    }
    /* a pass replays a captured graph (one per (row count, form)): a host that asks for one token at a time — the reference's runner
     * API — pays one graph launch, not ~260 kernel launches, per token; masked rows ride the same graphs (the table above is data) */
    hipGraphExec_t* slot = !graphs_enabled() ? nullptr
                           : !lm_head ? &graph_prefill[tiled_pass][nrows] : distinct ? &graph_exec[long_pass][nrows] : &graph_head_nf[tiled_pass][nrows];
    if (slot) {
        if (!capture_pass(slot, nrows, lm_head, lm_head && distinct)) return false;
        HIPQ(hipGraphLaunch(*slot, stream));
    } else {
        launch_error.clear();
        enqueue_pass(nrows, lm_head, distinct);
    }
    if (!launch_error.empty()) { error = launch_error; (void)hipStreamSynchronize(stream); return false; }
    HIPQ(hipGetLastError());
    if (!lm_head) { HIPQ(hipStreamSynchronize(stream)); return true; }
    if (logits_host) HIPQ(hipMemcpyAsync(logits_host, logits, (size_t)nrows * model->hp.vocab * 4, hipMemcpyDeviceToHost, stream));
    if (argmax_host) HIPQ(hipMemcpyAsync(argmax_host, d_tok, nrows * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

bool TkLlmSession::forward_stage(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, const float* x_in, float* x_out, bool x_on_host,
                                 int l0, int l1, bool head, int32_t* argmax_host) {
    const TkLlmHParams& h = model->hp;
    if (nrows <= 0 || nrows > TK_MAX_ROWS) { error = "nrows must be in [1,256]"; return false; }
    if (l0 < 0 || l1 < l0 || l1 > h.n_layer) { error = "bad layer range"; return false; }
    if ((tok == nullptr) == (x_in == nullptr)) { error = "a stage starts from tokens (first stage) or from a residual stream, not both"; return false; }
    if (head && l1 != h.n_layer) { error = "the head belongs to the stage that ends at the last layer"; return false; }
    if (!head && !x_out) { error = "a stage without the head must hand its residual stream on"; return false; }
    for (int r = 0; r < nrows; ++r)
        if (seq[r] < 0 || seq[r] >= max_seq || pos[r] < 0 || pos[r] >= max_ctx || (tok && (tok[r] < 0 || tok[r] >= h.vocab))) {
            error = "row out of range (sequence id, position or token id)";
            return false;
        }
    HIPQ(hipSetDevice(model->device));
    HIPQ(hipMemcpyAsync(d_seq, seq, nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemcpyAsync(d_pos, pos, nrows * 4, hipMemcpyHostToDevice, stream));
    if (tok) HIPQ(hipMemcpyAsync(d_tok, tok, nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemsetAsync(d_nsteps, 0, TK_MAX_ROWS * 4, stream));
    if (mask_rows_dirty) { HIPQ(hipMemsetAsync(d_mask_row, 0xFF, TK_MAX_ROWS * 4, stream)); mask_rows_dirty = false; }
    const size_t xb = (size_t)nrows * h.d_model * 4;
    if (x_in) HIPQ(hipMemcpyAsync(x, x_in, xb, x_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream));
    bool distinct = true;
    for (int a = 0; a < nrows && distinct; ++a)
        for (int b = a + 1; b < nrows; ++b)
            if (seq[a] == seq[b]) { distinct = false; break; }
    launch_error.clear();
    choose_attention(pos, nrows);
    enqueue_range(nrows, l0, l1, tok != nullptr, !head, head, distinct);
    if (!launch_error.empty()) { error = launch_error; (void)hipStreamSynchronize(stream); return false; }
    HIPQ(hipGetLastError());
    if (!head) HIPQ(hipMemcpyAsync(x_out, x, xb, x_on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, stream));
    else if (argmax_host) HIPQ(hipMemcpyAsync(argmax_host, d_tok, nrows * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

__global__ void k_stage_rows(const int32_t* __restrict__ tab, int n, int stride, int32_t* seq, int32_t* pos, int32_t* tok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { seq[i] = tab[i]; pos[i] = tab[stride + i]; tok[i] = tab[2 * stride + i]; }
}

/* Prompt rows that nobody samples from (all but the last token of every prompt): the whole schedule — (sequence, position, token) of
 * every row of every pass — is uploaded once, each pass is one tiny staging launch + one replay of a captured pass (no lm_head,
 * rope/append as its own kernel so a pass may hold several positions of one sequence), and the host synchronises once, in the final
 * sampling pass.  Same kernels, same order, same arithmetic as forward(): bit-identical results. */
bool TkLlmSession::prefill(int nseq, int n_prompt, const int32_t* tokens, int32_t* first_tokens_host) {
    if (nseq <= 0 || nseq > TK_MAX_ROWS || nseq > max_seq) { error = "nseq must be in [1, min(256, max_seq)]"; return false; }
    if (n_prompt <= 0 || n_prompt >= max_ctx) { error = "prompt does not fit the context"; return false; }
    for (int64_t i = 0; i < (int64_t)nseq * n_prompt; ++i)
        if (tokens[i] < 0 || tokens[i] >= model->hp.vocab) { error = "row out of range (sequence id, position or token id)"; return false; }
    HIPQ(hipSetDevice(model->device));
    /* all but the last prompt token: up to 256 rows per pass, positions ascending inside a sequence so causality holds inside a pass */
    const int64_t total = (int64_t)nseq * (n_prompt - 1);
    if (total > 0) {
        std::vector<int32_t> tab((size_t)3 * total);
        int64_t k = 0;
        for (int s = 0; s < nseq; ++s)
            for (int p = 0; p + 1 < n_prompt; ++p, ++k) {
                tab[k] = s; tab[total + k] = p; tab[2 * total + k] = tokens[(size_t)s * n_prompt + p];
            }
        if ((int64_t)tab_cap < 3 * total) {
            if (d_tab) (void)hipFree(d_tab);
            d_tab = nullptr;
            HIPQ(hipMalloc((void**)&d_tab, (size_t)3 * total * 4));
            tab_cap = (size_t)3 * total;
        }
        HIPQ(hipMemcpyAsync(d_tab, tab.data(), (size_t)3 * total * 4, hipMemcpyHostToDevice, stream));
        HIPQ(hipStreamSynchronize(stream)); /* `tab` is pageable host memory that dies with this scope */
        const bool use_graph = graphs_enabled();
        for (int64_t off = 0; off < total; off += TK_MAX_ROWS) {
            const int n = (int)std::min<int64_t>(TK_MAX_ROWS, total - off);
            choose_attention(tab.data() + total + off, n);
            if (use_graph && !capture_pass(&graph_prefill[tiled_pass][n], n, false, false)) return false;
            hipLaunchKernelGGL(k_stage_rows, dim3((n + 255) / 256), dim3(256), 0, stream, d_tab + off, n, (int)total, d_seq, d_pos, d_tok);
            if (use_graph) HIPQ(hipGraphLaunch(graph_prefill[tiled_pass][n], stream));
            else { launch_error.clear(); enqueue_pass(n, false, false); if (!launch_error.empty()) { error = launch_error; return false; } }
        }
        HIPQ(hipGetLastError());
    }
    /* last prompt token of every sequence: row r == sequence r, sampled -> decode() continues from here */
    std::vector<int32_t> sq(nseq), ps(nseq), tk(nseq);
    for (int s = 0; s < nseq; ++s) { sq[s] = s; ps[s] = n_prompt - 1; tk[s] = tokens[(size_t)s * n_prompt + n_prompt - 1]; }
    return forward(nseq, sq.data(), ps.data(), tk.data(), nullptr, first_tokens_host, true);
}

bool TkLlmSession::decode(int nrows, int n_steps, int32_t* out_tokens_host) {
    if (nrows <= 0 || nrows > TK_MAX_ROWS) { error = "nrows must be in [1,256]"; return false; }
    if (n_steps <= 0 || n_steps > hist_cap) { error = "n_steps exceeds the session context"; return false; }
    HIPQ(hipSetDevice(model->device));
    /* positions must stay inside the cache for the whole loop */
    int32_t hpos[TK_MAX_ROWS];
    HIPQ(hipMemcpyAsync(hpos, d_pos, nrows * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    for (int r = 0; r < nrows; ++r)
        if (hpos[r] + n_steps > max_ctx) { error = "decode would run past max_ctx"; return false; }
    const bool use_graph = graphs_enabled();
    int top0 = 0;
    for (int r = 0; r < nrows; ++r) top0 = hpos[r] > top0 ? hpos[r] : top0;
    HIPQ(hipMemsetAsync(d_nsteps, 0, TK_MAX_ROWS * 4, stream));
    if (mask_rows_dirty) { HIPQ(hipMemsetAsync(d_mask_row, 0xFF, TK_MAX_ROWS * 4, stream)); mask_rows_dirty = false; } /* the loop samples unconstrained */
    hipEvent_t e0, e1;
    HIPQ(hipEventCreate(&e0));
    HIPQ(hipEventCreate(&e1));
    HIPQ(hipEventRecord(e0, stream));
    for (int i = 0; i < n_steps; ++i) {
        choose_attention_top(top0 + i, nrows); /* positions advance on the device; the host knows where the loop stands */
        if (use_graph) {
            if (!capture_pass(&graph_exec[long_pass][nrows], nrows, true, true)) { (void)hipStreamSynchronize(stream); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return false; }
            HIPQ(hipGraphLaunch(graph_exec[long_pass][nrows], stream));
        } else { launch_error.clear(); enqueue_pass(nrows, true, true); if (!launch_error.empty()) { error = launch_error; break; } }
    }
    if (!launch_error.empty()) { (void)hipStreamSynchronize(stream); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); launch_error.clear(); return false; }
    HIPQ(hipGetLastError());
    HIPQ(hipEventRecord(e1, stream));
    if (out_tokens_host) HIPQ(hipMemcpyAsync(out_tokens_host, d_hist, (size_t)n_steps * TK_MAX_ROWS * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    float ms = 0.0f;
    HIPQ(hipEventElapsedTime(&ms, e0, e1));
    last_step_ms = ms / n_steps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return true;
}

bool TkLlmSession::time_gemv(int layer, int which, int nrows, int iters, float* avg_ms, double* algo_bytes) {
    const TkLlmHParams& h = model->hp;
    if (layer < 0 || layer >= h.n_layer || nrows < 1 || nrows > TK_MAX_ROWS) { error = "bad layer / nrows"; return false; }
    HIPQ(hipSetDevice(model->device));
    auto args_for = [&](int l, double* bytes) {
        const TkLlmLayer& L = model->layers[l];
        TkGemvArgs a{};
        a.nrows = nrows;
        if (which == 0) { /* gate+up */
            a.seg[0] = seg_of(L.gate); a.seg[1] = seg_of(L.up); a.nseg = 2; a.K = h.d_model; a.ks = h.ks_gateup; a.n_total = 2 * h.d_ff;
            if (tk_gemv_fuses_swiglu(nrows, h.ks_gateup, L.gate.type, L.up.type)) { a.swiglu = 1; a.n_total = h.d_ff; } /* the launch a wide pass makes */
            set_act(a, act_d); a.out = partial;
            *bytes = (double)L.gate.bytes + (double)L.up.bytes;
        } else if (which == 1) { /* down */
            a.seg[0] = seg_of(L.down); a.nseg = 1; a.K = h.d_ff; a.ks = h.ks_down; a.n_total = h.d_model;
            set_act(a, act_ff); a.out = partial;
            *bytes = (double)L.down.bytes;
        } else if (which == 4) { /* o */
            a.seg[0] = seg_of(L.o); a.nseg = 1; a.K = h.n_head * h.head_dim; a.ks = h.ks_o; a.n_total = h.d_model;
            set_act(a, act_qd); a.out = partial;
            *bytes = (double)L.o.bytes;
        } else if (which == 2) { /* qkv */
            a.seg[0] = seg_of(L.q); a.seg[1] = seg_of(L.k); a.seg[2] = seg_of(L.v); a.nseg = 3; a.K = h.d_model; a.ks = h.ks_qkv;
            a.n_total = (h.n_head + 2 * h.n_kv_head) * h.head_dim;
            set_act(a, act_d); a.out = partial;
            *bytes = (double)L.q.bytes + (double)L.k.bytes + (double)L.v.bytes;
        } else { /* lm head */
            a.seg[0] = seg_of(model->output); a.nseg = 1; a.K = h.d_model; a.ks = 1; a.n_total = h.vocab;
            set_act(a, act_d); a.out = logits;
            *bytes = (double)model->output.bytes;
        }
        /* algorithmic bytes of one launch: the weight tiles once, the int8 activation images (+ block scales and sub-block sums) once,
         * the fp32 K-split partial outputs once */
        *bytes += (double)nrows * ((double)a.K + (double)a.K / 256.0 * (4.0 + 16.0)) + (double)a.ks * nrows * (double)a.n_total * 4.0;
        return a;
    };
    /* every layer whose tensors have the same types as `layer`: cycling through them keeps each launch on
     * cold HBM lines (32 layers x 66 MB >> the 256 MiB Infinity Cache), as in a real decode step */
    std::vector<TkGemvArgs> set;
    double bytes = 0.0;
    const char* hot = getenv("TK_MI355X_TIME_HOT"); /* =1: the SAME layer's launch back to back (weights resident in L2 / Infinity Cache): the bound a prefetch could reach */
    for (int l = 0; l < h.n_layer; ++l) {
        if ((which == 3 || (hot && hot[0] == '1')) && l != layer) continue;
        if (model->layers[l].v.type != model->layers[layer].v.type || model->layers[l].down.type != model->layers[layer].down.type) continue;
        double b;
        set.push_back(args_for(l, &b));
        bytes = b;
    }
    *algo_bytes = bytes;
    hipEvent_t e0, e1;
    HIPQ(hipEventCreate(&e0));
    HIPQ(hipEventCreate(&e1));
    for (size_t i = 0; i < set.size(); ++i) tk_launch_gemv(set[i], stream);
    /* launched the way decode() launches them: as nodes of a hipGraph (unless TK_MI355X_NO_GRAPH=1, see decode()) */
    const bool use_graph = graphs_enabled();
    hipGraphExec_t ge = nullptr;
    if (use_graph) {
        std::lock_guard<std::mutex> lk(g_capture_mu);
        hipGraph_t g = nullptr;
        HIPQ(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < iters; ++i) tk_launch_gemv(set[(size_t)i % set.size()], stream);
        HIPQ(hipStreamEndCapture(stream, &g));
        HIPQ(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPQ(hipGraphDestroy(g));
        HIPQ(hipGraphLaunch(ge, stream)); /* warm */
    }
    HIPQ(hipEventRecord(e0, stream));
    if (use_graph) HIPQ(hipGraphLaunch(ge, stream));
    else for (int i = 0; i < iters; ++i) tk_launch_gemv(set[(size_t)i % set.size()], stream);
    HIPQ(hipEventRecord(e1, stream));
    HIPQ(hipStreamSynchronize(stream));
    if (ge) (void)hipGraphExecDestroy(ge);
    float ms = 0.0f;
    HIPQ(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return true;
}

/* stand-alone timing of the decode attention launch (k_attention, fused rope/append form) at `nrows` rows whose sequences all sit at
 * position ctx - 1: `iters` launches cycling through the layers (each layer's cache region is its own HBM range), launched as hipGraph
 * nodes like decode().  kv_bytes = the K and V rows one launch must read once: nrows * ctx * n_kv_head * head_dim * 2 B * 2. */
bool TkLlmSession::time_attention(int nrows, int ctx, int iters, float* avg_ms, double* kv_bytes) {
    const TkLlmHParams& h = model->hp;
    if (nrows < 1 || nrows > TK_MAX_ROWS || nrows > max_seq || ctx < 1 || ctx > max_ctx || iters < 1) { error = "bad nrows / ctx / iters"; return false; }
    HIPQ(hipSetDevice(model->device));
    std::vector<int32_t> sq(nrows), ps(nrows, ctx - 1);
    for (int r = 0; r < nrows; ++r) sq[r] = r;
    HIPQ(hipMemcpyAsync(d_seq, sq.data(), nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipMemcpyAsync(d_pos, ps.data(), nrows * 4, hipMemcpyHostToDevice, stream));
    HIPQ(hipStreamSynchronize(stream));
    const int QD = h.n_head * h.head_dim, KVD = h.n_kv_head * h.head_dim;
    /* TK_MI355X_TIME_UNFUSED=1: time the two-launch form instead (k_qkv_rope_append + k_attention<.., not fused>): what a decode pass would
     * cost without the fused prologue */
    const char* uf = getenv("TK_MI355X_TIME_UNFUSED");
    const char* lf = getenv("TK_MI355X_TIME_LONG"); /* =1: the three-launch long-context form (tk_launch_attention_long) where it applies */
    const bool longf = lf && lf[0] == '1' && d_scores && tk_attention_long_applies(nrows, h.n_head, h.n_kv_head, h.head_dim);
    const bool unfused = (uf && uf[0] == '1') || longf;
    auto launch = [&](int l) {
        if (longf) {
            tk_launch_attention_long(partial, h.ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head, h.head_dim, l % h.n_layer,
                                     max_seq, max_ctx, d_scores, act_qd, stream);
            return;
        }
        if (unfused)
            tk_launch_qkv_rope_append(partial, h.ks_qkv, QD + 2 * KVD, h.n_head, h.n_kv_head, h.head_dim, rope_cos, rope_sin, d_seq, d_pos, nrows, qbuf, kcache,
                                      vcache, l % h.n_layer, max_seq, max_ctx, stream);
        tk_launch_attention(qbuf, partial, h.ks_qkv, QD + 2 * KVD, rope_cos, rope_sin, kcache, vcache, d_seq, d_pos, nrows, h.n_head, h.n_kv_head,
                            h.head_dim, l % h.n_layer, max_seq, max_ctx, act_qd, !unfused, stream);
    };
    *kv_bytes = (double)nrows * ctx * KVD * 2.0 * 2.0;
    const bool use_graph = graphs_enabled();
    hipGraphExec_t ge = nullptr;
    launch(0);
    if (use_graph) {
        std::lock_guard<std::mutex> lk(g_capture_mu);
        hipGraph_t g = nullptr;
        HIPQ(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < iters; ++i) launch(i);
        HIPQ(hipStreamEndCapture(stream, &g));
        HIPQ(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPQ(hipGraphDestroy(g));
        HIPQ(hipGraphLaunch(ge, stream)); /* warm */
    }
    hipEvent_t e0, e1;
    HIPQ(hipEventCreate(&e0));
    HIPQ(hipEventCreate(&e1));
    HIPQ(hipEventRecord(e0, stream));
    if (use_graph) HIPQ(hipGraphLaunch(ge, stream));
    else for (int i = 0; i < iters; ++i) launch(i);
    HIPQ(hipEventRecord(e1, stream));
    HIPQ(hipStreamSynchronize(stream));
    if (ge) (void)hipGraphExecDestroy(ge);
    float ms = 0.0f;
    HIPQ(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return true;
}
