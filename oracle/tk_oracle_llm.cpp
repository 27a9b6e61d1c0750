/*
 * tk_oracle_llm.cpp — TEST INFRASTRUCTURE ONLY (oracle).  Never linked into, loaded by
 * or called from the product library; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED against the reference: the LLM arithmetic the reference runs lives in
 * ggml-org/llama.cpp, an un-fetched, un-pinned submodule (/root/reference/.gitmodules:1-3,
 * src/llama.cpp/ is empty) and no reference test pins a token or logit.
 * What this file restates instead:
 *   - the call sequence of the reference runner: prefill of the whole prompt then one
 *     llama_decode per token (src/ai_models/tk_runner_streaming.c:20-34, :57-85), KV cache
 *     cleared per prompt (:31), argmax sampling (the build's definition — SURVEY §0 F8);
 *   - the published Mistral-7B-v0.1 decoder (RMSNorm eps 1e-5, GQA 32q/8kv x 128, RoPE
 *     theta 1e4 on adjacent pairs as the GGUF llama-arch convention stores q/k, SwiGLU);
 *   - llama.cpp's CPU numerics structure for k-quants: activations are quantised per
 *     256-block to int8 (Q8_K, bsums) and the dot product is integer inside
 *     a block, float across blocks (ggml_vec_dot_q4_K_q8_K / q6_K_q8_K as published).
 *     The Q8_K quantiser is the published quantize_row_q8_K_ref (iscale = -127 / max of the signed extreme, d = 1 / iscale; round 6 — earlier
 *     rounds used d = amax / 127).  What still differs from the engine the reference calls is the K-split summation order of a matmul (part of
 *     this build's canonical order, exported with the model) — and that llama.cpp itself is absent.
 * It is pinned by (i) an independent fp32 torch implementation on de-quantised weights
 * (tests/golden/make_llm_golden.py -> tests/golden/llm_tiny.npz) and (ii) the codec
 * round-trip fixtures.
 *
 * Every float reduction below has ONE canonical order, chosen so a wave-parallel kernel can
 * reproduce it exactly; the HIP path must match these results bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../trackiellm_amd/csrc/common/tk_exact_math.h"
#include "../trackiellm_amd/csrc/common/tk_ggml_blocks.h"

extern "C" {

typedef struct {
    int32_t n_layer, d_model, n_head, n_kv_head, head_dim, d_ff, vocab, max_ctx, max_seq;
    float rms_eps, rope_theta;
    /* canonical K-split counts of the five matmul groups (part of the summation order) */
    int32_t ks_qkv, ks_o, ks_gateup, ks_down, ks_out;
} orc_llm_config_t;

enum { ORC_T_TOKEN_EMBD = 0, ORC_T_OUT_NORM = 1, ORC_T_OUTPUT = 2, ORC_T_LAYER0 = 16 };
enum { ORC_L_ATTN_NORM = 0, ORC_L_Q, ORC_L_K, ORC_L_V, ORC_L_O, ORC_L_FFN_NORM, ORC_L_GATE, ORC_L_UP, ORC_L_DOWN, ORC_L_COUNT };

struct orc_tensor {
    int type = TK_TYPE_F32;
    int64_t rows = 0, cols = 0;
    std::vector<uint8_t> data;
};

struct orc_llm {
    orc_llm_config_t cfg;
    orc_tensor token_embd, out_norm, output;
    std::vector<orc_tensor> layers; /* n_layer * ORC_L_COUNT */
    std::vector<uint16_t> kcache, vcache; /* [layer][seq][ctx][kv_head][head_dim] f16 */
    std::vector<float> rope_cos, rope_sin; /* [ctx][head_dim/2] */
};

/* ---------------- canonical primitives (also exported for kernel-level tests) ---------------- */

/* sum of 256 strided partials, then xor-butterfly tree per 64-lane wave, then waves in order */
static float orc_sum256(const float* partial) {
    float w[4];
    for (int wv = 0; wv < 4; ++wv) {
        float p[64];
        for (int i = 0; i < 64; ++i) p[i] = partial[wv * 64 + i];
        for (int s = 32; s >= 1; s >>= 1)
            for (int i = 0; i < s; ++i) p[i] = p[i] + p[i + s];
        w[wv] = p[0];
    }
    return ((w[0] + w[1]) + w[2]) + w[3];
}

void orc_rmsnorm(const float* x, const float* w, int n, float eps, float* out) {
    float partial[256];
    /* thread t of 256 owns the float4 groups t, t+256, ... and squares them element by element (the kernel's 16-byte loads) */
    for (int t = 0; t < 256; ++t) {
        float a = 0.0f;
        for (int g = t; g < n / 4; g += 256)
            for (int e = 0; e < 4; ++e) a = tk_fmaf(x[4 * g + e], x[4 * g + e], a);
        partial[t] = a;
    }
    float ss = orc_sum256(partial);
    float mean = tk_divf(ss, (float)n);
    float scale = tk_divf(1.0f, tk_sqrtf(mean + eps));
    for (int i = 0; i < n; ++i) out[i] = (x[i] * scale) * w[i];
}

/* Q8_K activation quantisation of one row, as ggml publishes it (quantize_row_q8_K_ref): per 256-block the signed value `max` of the FIRST element
 * of largest magnitude, iscale = -127 / max, q = min(127, nearest_int(iscale * x)), d = 1 / iscale (so d carries the sign of -max and the extreme
 * element becomes -127); an all-zero block has d = 0, q = 0.  q[n] int8, d[n/256], bsum32[n/32] (ggml keeps sums of 16: a Q4_K sub-block of 32 adds
 * two of them) */
void orc_q8k_quantize(const float* x, int n, int8_t* q, float* d, int32_t* bsum32) {
    for (int b = 0; b < n / 256; ++b) {
        const float* xb = x + 256 * b;
        float amax = 0.0f, mx = 0.0f;
        for (int i = 0; i < 256; ++i) {
            const float ax = tk_fabsf(xb[i]);
            if (ax > amax) { amax = ax; mx = xb[i]; }
        }
        const float iscale = amax > 0.0f ? tk_divf(-127.0f, mx) : 0.0f;
        d[b] = amax > 0.0f ? tk_divf(1.0f, iscale) : 0.0f;
        for (int j = 0; j < 8; ++j) {
            int32_t s = 0;
            for (int i = 0; i < 32; ++i) {
                int v = (int)tk_rintf(iscale * xb[32 * j + i]); /* nearest_int: round to nearest, ties to even */
                v = v > 127 ? 127 : v;
                q[256 * b + 32 * j + i] = (int8_t)v;
                s += v;
            }
            bsum32[8 * b + j] = s;
        }
    }
}

/* one (row, K-range) partial of a Q4_K x Q8_K dot: blocks ascending, two fmas per block.
 * The nibbles are unpacked to a byte array first so the integer dot auto-vectorises (AVX2). */
static float dot_q4k_range(const tk_block_q4_K* w, const int8_t* q, const float* d, const int32_t* bsum, int b0, int b1) {
    float acc = 0.0f;
    for (int b = b0; b < b1; ++b) {
        const tk_block_q4_K* blk = w + b;
        const int8_t* qb = q + 256 * b;
        int8_t u[256];
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 32; ++l) {
                uint8_t byte = blk->qs[32 * c + l];
                u[64 * c + l] = (int8_t)(byte & 0x0F);
                u[64 * c + 32 + l] = (int8_t)(byte >> 4);
            }
        int32_t P = 0, M = 0;
        for (int j = 0; j < 8; ++j) {
            uint8_t sc, mn;
            tk_q4k_get_scale_min(j, blk->scales, &sc, &mn);
            int32_t s = 0;
            for (int l = 0; l < 32; ++l) s += (int32_t)u[32 * j + l] * (int32_t)qb[32 * j + l];
            P += (int32_t)sc * s;
            M += (int32_t)mn * bsum[8 * b + j];
        }
        float dw = tk_f16_to_f32(blk->d), dmin = tk_f16_to_f32(blk->dmin);
        float s1f = dw * d[b];
        float s2f = dmin * d[b];
        acc = tk_fmaf(s1f, (float)P, acc);
        acc = tk_fmaf(-s2f, (float)M, acc);
    }
    return acc;
}

static float dot_q6k_range(const tk_block_q6_K* w, const int8_t* q, const float* d, int b0, int b1) {
    float acc = 0.0f;
    for (int b = b0; b < b1; ++b) {
        const tk_block_q6_K* blk = w + b;
        const int8_t* qb = q + 256 * b;
        int8_t u[256];
        for (int n = 0; n < 2; ++n)
            for (int l = 0; l < 32; ++l) {
                const uint8_t la = blk->ql[64 * n + l], lb = blk->ql[64 * n + 32 + l], h = blk->qh[32 * n + l];
                u[128 * n + l] = (int8_t)((la & 0x0F) | ((h & 3) << 4)) - 32;
                u[128 * n + 32 + l] = (int8_t)((lb & 0x0F) | (((h >> 2) & 3) << 4)) - 32;
                u[128 * n + 64 + l] = (int8_t)((la >> 4) | (((h >> 4) & 3) << 4)) - 32;
                u[128 * n + 96 + l] = (int8_t)((lb >> 4) | (((h >> 6) & 3) << 4)) - 32;
            }
        int32_t P = 0;
        for (int g = 0; g < 16; ++g) {
            int32_t s = 0;
            for (int i = 0; i < 16; ++i) s += (int32_t)u[16 * g + i] * (int32_t)qb[16 * g + i];
            P += (int32_t)blk->scales[g] * s;
        }
        float dw = tk_f16_to_f32(blk->d);
        acc = tk_fmaf(dw * d[b], (float)P, acc);
    }
    return acc;
}

/* y[rows] = W[rows x K] . x, K split into `ks` ranges whose partials are added in order */
void orc_gemv_q8(int type, const void* w, int64_t rows, int64_t K, int ks, const int8_t* q, const float* d,
                 const int32_t* bsum, float* y) {
    int nb = (int)(K / 256);
    int per = nb / ks;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float acc = 0.0f;
        for (int s = 0; s < ks; ++s) {
            float p;
            if (type == TK_TYPE_Q4_K) p = dot_q4k_range((const tk_block_q4_K*)w + r * nb, q, d, bsum, s * per, (s + 1) * per);
            else p = dot_q6k_range((const tk_block_q6_K*)w + r * nb, q, d, s * per, (s + 1) * per);
            acc = (s == 0) ? p : acc + p;
        }
        y[r] = acc;
    }
}

void orc_dequant_row(int type, const void* w, int64_t K, int64_t row, float* out) {
    int nb = (int)(K / 256);
    for (int64_t i = 0; i < K; ++i) {
        if (type == TK_TYPE_Q4_K) out[i] = tk_q4k_dequant((const tk_block_q4_K*)w + row * nb + i / 256, (int)(i % 256));
        else if (type == TK_TYPE_Q6_K) out[i] = tk_q6k_dequant((const tk_block_q6_K*)w + row * nb + i / 256, (int)(i % 256));
        else if (type == TK_TYPE_F16) out[i] = tk_f16_to_f32(((const uint16_t*)w)[row * K + i]);
        else out[i] = ((const float*)w)[row * K + i];
    }
}

void orc_quantize_rows(int type, const float* x, int64_t n, void* out) {
    int64_t nb = n / 256;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b) {
        if (type == TK_TYPE_Q4_K) tk_quantize_q4_K(x + 256 * b, (tk_block_q4_K*)out + b);
        else tk_quantize_q6_K(x + 256 * b, (tk_block_q6_K*)out + b);
    }
}

/* exposed exact-math probes (tests pin them against libm) */
float orc_expf(float x) { return tk_expf(x); }
float orc_logf(float x) { return tk_logf(x); }
float orc_tanhf(float x) { return tk_tanhf(x); }
float orc_geluf(float x) { return tk_geluf(x); }
float orc_siluf(float x) { return tk_siluf(x); }
uint16_t orc_f32_to_f16(float x) { return tk_f32_to_f16(x); }
float orc_f16_to_f32(uint16_t h) { return tk_f16_to_f32(h); }

/* ---------------- model ---------------- */

static int use_more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; }

/* tensor type recipe "Q4_K_M" (SURVEY §8d): attn_v / ffn_down Q6_K where use_more_bits, output Q6_K */
int orc_llm_tensor_type(const orc_llm_config_t* cfg, int layer, int which) {
    if (layer < 0) return which == ORC_T_OUTPUT ? TK_TYPE_Q6_K : (which == ORC_T_TOKEN_EMBD ? TK_TYPE_Q4_K : TK_TYPE_F32);
    if (which == ORC_L_ATTN_NORM || which == ORC_L_FFN_NORM) return TK_TYPE_F32;
    if ((which == ORC_L_V || which == ORC_L_DOWN) && use_more_bits(layer, cfg->n_layer)) return TK_TYPE_Q6_K;
    return TK_TYPE_Q4_K;
}

static void tensor_shape(const orc_llm_config_t* c, int layer, int which, int64_t* rows, int64_t* cols) {
    int64_t d = c->d_model, kv = (int64_t)c->n_kv_head * c->head_dim, qd = (int64_t)c->n_head * c->head_dim;
    if (layer < 0) {
        if (which == ORC_T_OUT_NORM) { *rows = 1; *cols = d; }
        else { *rows = c->vocab; *cols = d; }
        return;
    }
    switch (which) {
        case ORC_L_ATTN_NORM: case ORC_L_FFN_NORM: *rows = 1; *cols = d; break;
        case ORC_L_Q: *rows = qd; *cols = d; break;
        case ORC_L_K: case ORC_L_V: *rows = kv; *cols = d; break;
        case ORC_L_O: *rows = d; *cols = qd; break;
        case ORC_L_GATE: case ORC_L_UP: *rows = c->d_ff; *cols = d; break;
        default: *rows = d; *cols = c->d_ff; break;
    }
}

static uint64_t tensor_id(int layer, int which) { return layer < 0 ? (uint64_t)which : (uint64_t)(ORC_T_LAYER0 + layer * 16 + which); }

static void synth_tensor(orc_tensor* t, uint64_t seed, uint64_t tid, int type, int64_t rows, int64_t cols) {
    t->type = type; t->rows = rows; t->cols = cols;
    int64_t n = rows * cols;
    if (type == TK_TYPE_F32) {
        t->data.resize(n * 4);
        float* f = (float*)t->data.data();
        for (int64_t i = 0; i < n; ++i) f[i] = 1.0f + 0.1f * tk_synth_normal(seed, tid, (uint64_t)i);
        return;
    }
    if (type == TK_TYPE_F16) {
        t->data.resize(n * 2);
        uint16_t* hp = (uint16_t*)t->data.data();
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) hp[i] = tk_f32_to_f16(0.02f * tk_synth_normal(seed, tid, (uint64_t)i));
        return;
    }
    int64_t nb = n / 256;
    t->data.resize(nb * tk_type_block_bytes(type));
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b) {
        float x[256];
        for (int i = 0; i < 256; ++i) x[i] = 0.02f * tk_synth_normal(seed, tid, (uint64_t)(256 * b + i));
        if (type == TK_TYPE_Q4_K) tk_quantize_q4_K(x, (tk_block_q4_K*)t->data.data() + b);
        else tk_quantize_q6_K(x, (tk_block_q6_K*)t->data.data() + b);
    }
}

orc_llm* orc_llm_create(const orc_llm_config_t* cfg) {
    orc_llm* m = new orc_llm();
    m->cfg = *cfg;
    m->layers.resize((size_t)cfg->n_layer * ORC_L_COUNT);
    size_t kv = (size_t)cfg->n_layer * cfg->max_seq * cfg->max_ctx * cfg->n_kv_head * cfg->head_dim;
    m->kcache.assign(kv, 0);
    m->vcache.assign(kv, 0);
    int half = cfg->head_dim / 2;
    m->rope_cos.resize((size_t)cfg->max_ctx * half);
    m->rope_sin.resize((size_t)cfg->max_ctx * half);
    for (int p = 0; p < cfg->max_ctx; ++p)
        for (int i = 0; i < half; ++i) {
            double theta = pow((double)cfg->rope_theta, -2.0 * i / (double)cfg->head_dim);
            double a = (double)p * theta;
            m->rope_cos[(size_t)p * half + i] = (float)cos(a);
            m->rope_sin[(size_t)p * half + i] = (float)sin(a);
        }
    return m;
}

void orc_llm_destroy(orc_llm* m) { delete m; }

static int g_synth_f16 = 0; /* orc_llm_synth_f16: every matrix and the embedding as IEEE f16 (the fp16 checkpoint recipe), norms f32 */
static int synth_type(const orc_llm_config_t* c, int layer, int which) {
    const int t = orc_llm_tensor_type(c, layer, which);
    return (g_synth_f16 && t != TK_TYPE_F32) ? (int)TK_TYPE_F16 : t;
}

void orc_llm_synth(orc_llm* m, uint64_t seed) {
    const orc_llm_config_t* c = &m->cfg;
    int64_t r, k;
    tensor_shape(c, -1, ORC_T_TOKEN_EMBD, &r, &k);
    synth_tensor(&m->token_embd, seed, tensor_id(-1, ORC_T_TOKEN_EMBD), synth_type(c, -1, ORC_T_TOKEN_EMBD), r, k);
    tensor_shape(c, -1, ORC_T_OUT_NORM, &r, &k);
    synth_tensor(&m->out_norm, seed, tensor_id(-1, ORC_T_OUT_NORM), TK_TYPE_F32, r, k);
    tensor_shape(c, -1, ORC_T_OUTPUT, &r, &k);
    synth_tensor(&m->output, seed, tensor_id(-1, ORC_T_OUTPUT), synth_type(c, -1, ORC_T_OUTPUT), r, k);
    for (int l = 0; l < c->n_layer; ++l)
        for (int w = 0; w < ORC_L_COUNT; ++w) {
            tensor_shape(c, l, w, &r, &k);
            synth_tensor(&m->layers[(size_t)l * ORC_L_COUNT + w], seed, tensor_id(l, w), synth_type(c, l, w), r, k);
        }
}

void orc_llm_synth_f16(orc_llm* m, uint64_t seed) {
    g_synth_f16 = 1;
    orc_llm_synth(m, seed);
    g_synth_f16 = 0;
}

/* load one tensor in GGUF block layout (layer = -1 for the three global tensors) */
int orc_llm_set_tensor(orc_llm* m, int layer, int which, int type, const void* data, int64_t nbytes) {
    orc_tensor* t = layer < 0 ? (which == ORC_T_TOKEN_EMBD ? &m->token_embd : which == ORC_T_OUT_NORM ? &m->out_norm : &m->output)
                              : &m->layers[(size_t)layer * ORC_L_COUNT + which];
    tensor_shape(&m->cfg, layer, which, &t->rows, &t->cols);
    t->type = type;
    int64_t expect = t->rows * t->cols / (int64_t)tk_type_block_elems(type) * (int64_t)tk_type_block_bytes(type);
    if (expect != nbytes) return -1;
    t->data.assign((const uint8_t*)data, (const uint8_t*)data + nbytes);
    return 0;
}

/* copy a tensor's GGUF-layout bytes out (lets tests feed identical blocks to the product) */
int64_t orc_llm_get_tensor(orc_llm* m, int layer, int which, int* type, void* out, int64_t cap) {
    orc_tensor* t = layer < 0 ? (which == ORC_T_TOKEN_EMBD ? &m->token_embd : which == ORC_T_OUT_NORM ? &m->out_norm : &m->output)
                              : &m->layers[(size_t)layer * ORC_L_COUNT + which];
    if (type) *type = t->type;
    if (out && cap >= (int64_t)t->data.size()) memcpy(out, t->data.data(), t->data.size());
    return (int64_t)t->data.size();
}

/* LoRA merge, the reference's llama_model_apply_lora_from_file (src/ai_models/tk_model_loader.c:259-270; llama.cpp un-vendored, ggml's add on a
 * quantised destination restated): W' = W + scale (B A), A [r][k_in], B [n_out][r]; per weight delta = fma chain over j = 0 .. r-1 from 0.0f of
 * B[n][j] A[j][k], w' = w + scale * delta (multiply, then add), every 256-weight block quantised back to the tensor's own type.  0 on success. */
int orc_llm_apply_lora(orc_llm* m, int layer, int which, const float* A, const float* B, int r, float scale) {
    orc_tensor* t = layer < 0 ? (which == ORC_T_TOKEN_EMBD ? &m->token_embd : which == ORC_T_OUT_NORM ? &m->out_norm : &m->output)
                              : &m->layers[(size_t)layer * ORC_L_COUNT + which];
    if (t->type != TK_TYPE_Q4_K && t->type != TK_TYPE_Q6_K && t->type != TK_TYPE_F16) return -1;
    if (t->cols % 256 || r < 1) return -1;
    const int64_t K = t->cols, nblk = K / 256;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < t->rows; ++n)
        for (int64_t bk = 0; bk < nblk; ++bk) {
            float x[256];
            const int64_t b = n * nblk + bk;
            for (int e = 0; e < 256; ++e) {
                float w;
                if (t->type == TK_TYPE_Q4_K) w = tk_q4k_dequant((const tk_block_q4_K*)t->data.data() + b, e);
                else if (t->type == TK_TYPE_Q6_K) w = tk_q6k_dequant((const tk_block_q6_K*)t->data.data() + b, e);
                else w = tk_f16_to_f32(((const uint16_t*)t->data.data())[b * 256 + e]);
                float delta = 0.0f;
                for (int j = 0; j < r; ++j) delta = tk_fmaf(B[n * r + j], A[(int64_t)j * K + bk * 256 + e], delta);
                x[e] = w + scale * delta;
            }
            if (t->type == TK_TYPE_Q4_K) tk_quantize_q4_K(x, (tk_block_q4_K*)t->data.data() + b);
            else if (t->type == TK_TYPE_Q6_K) tk_quantize_q6_K(x, (tk_block_q6_K*)t->data.data() + b);
            else for (int e = 0; e < 256; ++e) ((uint16_t*)t->data.data())[b * 256 + e] = tk_f32_to_f16(x[e]);
        }
    return 0;
}

/* structure-check mode: skip activation quantisation, plain fp32 dot on de-quantised weights
 * (used ONLY by tests/golden/make_llm_golden.py to compare with HF at ~1e-5) */
static int g_fp32_activations = 0;
void orc_set_fp32_activations(int on) { g_fp32_activations = on; }

static void gemv_f32(const orc_tensor& t, const float* h, float* y) {
    int K = (int)t.cols;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < t.rows; ++r) {
        std::vector<float> w(K);
        orc_dequant_row(t.type, t.data.data(), K, r, w.data());
        double a = 0.0;
        for (int i = 0; i < K; ++i) a += (double)w[i] * h[i];
        y[r] = (float)a;
    }
}

/* f16 weight matrix (fp16 checkpoints): the activation row is rounded through f16 — what a CPU engine's f16 matmul does to its f32 input
 * (ggml converts src1 to f16, then dots in f32).  K-split like the quantised path: per slab ONE fp32 fma chain over k ascending from zero,
 * the slabs added in ascending order */
static void gemv_f16(const orc_tensor& t, int ks, const float* h, float* y) {
    const int K = (int)t.cols, Kr = K / ks;
    std::vector<float> a(K);
    for (int i = 0; i < K; ++i) a[i] = tk_f16_to_f32(tk_f32_to_f16(h[i]));
    const uint16_t* w = (const uint16_t*)t.data.data();
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < t.rows; ++r) {
        float total = 0.0f;
        for (int s = 0; s < ks; ++s) {
            float acc = 0.0f;
            for (int i = s * Kr; i < (s + 1) * Kr; ++i) acc = tk_fmaf(a[i], tk_f16_to_f32(w[r * K + i]), acc);
            total = s == 0 ? acc : total + acc;
        }
        y[r] = total;
    }
}

/* a matmul on an already quantised row (q / d / bs) or, for f16 tensors, on the f32 row itself */
static void mat(const orc_tensor& t, int ks, const float* h, const int8_t* q, const float* d, const int32_t* bs, float* y) {
    if (g_fp32_activations) { gemv_f32(t, h, y); return; }
    if (t.type == TK_TYPE_F16) { gemv_f16(t, ks, h, y); return; }
    orc_gemv_q8(t.type, t.data.data(), t.rows, (int)t.cols, ks, q, d, bs, y);
}

static void matvec(const orc_tensor& t, int ks, const float* h, float* y, std::vector<int8_t>& q, std::vector<float>& d,
                   std::vector<int32_t>& bs) {
    if (g_fp32_activations) { gemv_f32(t, h, y); return; }
    if (t.type == TK_TYPE_F16) { gemv_f16(t, ks, h, y); return; }
    int K = (int)t.cols;
    q.resize(K); d.resize(K / 256); bs.resize(K / 32);
    orc_q8k_quantize(h, K, q.data(), d.data(), bs.data());
    orc_gemv_q8(t.type, t.data.data(), t.rows, K, ks, q.data(), d.data(), bs.data(), y);
}

/*
 * One pass over n_rows (seq, pos, token) rows.  All rows' K/V are appended before any row
 * attends, so several rows of one sequence (prefill) see each other causally.
 * logits: [n_rows][vocab] (may be NULL), argmax: [n_rows] first index of the maximum.
 */
void orc_llm_forward(orc_llm* m, int n_rows, const int32_t* seq, const int32_t* pos, const int32_t* tok, float* logits,
                     int32_t* argmax) {
    const orc_llm_config_t& c = m->cfg;
    const int D = c.d_model, HD = c.head_dim, NH = c.n_head, NKV = c.n_kv_head, FF = c.d_ff, half = HD / 2;
    const int QD = NH * HD, KVD = NKV * HD, grp = NH / NKV;
    std::vector<float> x((size_t)n_rows * D), h(D), qv((size_t)n_rows * QD), kv(KVD), vv(KVD), att((size_t)n_rows * QD), o(D);
    std::vector<float> gate(FF), up(FF), act(FF), sc(c.max_ctx), lg(c.vocab);
    std::vector<int8_t> q8; std::vector<float> qd; std::vector<int32_t> qb;
    const float att_scale = tk_divf(1.0f, tk_sqrtf((float)HD));

    for (int r = 0; r < n_rows; ++r) orc_dequant_row(m->token_embd.type, m->token_embd.data.data(), D, tok[r], &x[(size_t)r * D]);

    for (int l = 0; l < c.n_layer; ++l) {
        const orc_tensor* L = &m->layers[(size_t)l * ORC_L_COUNT];
        /* phase 1: norm, q/k/v, rope, kv append for every row */
        for (int r = 0; r < n_rows; ++r) {
            float* xr = &x[(size_t)r * D];
            orc_rmsnorm(xr, (const float*)L[ORC_L_ATTN_NORM].data.data(), D, c.rms_eps, h.data());
            q8.resize(D); qd.resize(D / 256); qb.resize(D / 32);
            orc_q8k_quantize(h.data(), D, q8.data(), qd.data(), qb.data());
            float* qr = &qv[(size_t)r * QD];
            mat(L[ORC_L_Q], c.ks_qkv, h.data(), q8.data(), qd.data(), qb.data(), qr);
            mat(L[ORC_L_K], c.ks_qkv, h.data(), q8.data(), qd.data(), qb.data(), kv.data());
            mat(L[ORC_L_V], c.ks_qkv, h.data(), q8.data(), qd.data(), qb.data(), vv.data());
            const float* cs = &m->rope_cos[(size_t)pos[r] * half];
            const float* sn = &m->rope_sin[(size_t)pos[r] * half];
            for (int hh = 0; hh < NH; ++hh)
                for (int i = 0; i < half; ++i) {
                    float a = qr[hh * HD + 2 * i], b = qr[hh * HD + 2 * i + 1];
                    qr[hh * HD + 2 * i] = tk_fmaf(-b, sn[i], a * cs[i]);
                    qr[hh * HD + 2 * i + 1] = tk_fmaf(a, sn[i], b * cs[i]);
                }
            size_t base = ((((size_t)l * c.max_seq + seq[r]) * c.max_ctx + pos[r]) * NKV) * HD;
            for (int hh = 0; hh < NKV; ++hh)
                for (int i = 0; i < half; ++i) {
                    float a = kv[hh * HD + 2 * i], b = kv[hh * HD + 2 * i + 1];
                    m->kcache[base + hh * HD + 2 * i] = tk_f32_to_f16(tk_fmaf(-b, sn[i], a * cs[i]));
                    m->kcache[base + hh * HD + 2 * i + 1] = tk_f32_to_f16(tk_fmaf(a, sn[i], b * cs[i]));
                }
            for (int i = 0; i < KVD; ++i) m->vcache[base + i] = tk_f32_to_f16(vv[i]);
        }
        /* phase 2: attention, o-proj, ffn per row */
        for (int r = 0; r < n_rows; ++r) {
            float* xr = &x[(size_t)r * D];
            const float* qr = &qv[(size_t)r * QD];
            float* ar = &att[(size_t)r * QD];
            int T = pos[r] + 1;
            size_t sbase = (((size_t)l * c.max_seq + seq[r]) * c.max_ctx) * NKV * HD;
            for (int hh = 0; hh < NH; ++hh) {
                int kvh = hh / grp;
                float mx = -INFINITY;
                for (int t = 0; t < T; ++t) {
                    const uint16_t* kr = &m->kcache[sbase + ((size_t)t * NKV + kvh) * HD];
                    float a = 0.0f;
                    for (int i = 0; i < HD; ++i) a = tk_fmaf(qr[hh * HD + i], tk_f16_to_f32(kr[i]), a);
                    sc[t] = a * att_scale;
                    mx = tk_fmaxf(mx, sc[t]);
                }
                /* canonical order: 4 interleaved partial sums over positions (t mod 4), combined in order */
                float lp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int t = 0; t < T; ++t) { sc[t] = tk_expf(sc[t] - mx); lp[t & 3] = lp[t & 3] + sc[t]; }
                const float lsum = ((lp[0] + lp[1]) + lp[2]) + lp[3];
                for (int i = 0; i < HD; ++i) {
                    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    for (int t = 0; t < T; ++t)
                        a[t & 3] = tk_fmaf(sc[t], tk_f16_to_f32(m->vcache[sbase + ((size_t)t * NKV + kvh) * HD + i]), a[t & 3]);
                    ar[hh * HD + i] = tk_divf(((a[0] + a[1]) + a[2]) + a[3], lsum);
                }
            }
            matvec(L[ORC_L_O], c.ks_o, ar, o.data(), q8, qd, qb);
            for (int i = 0; i < D; ++i) xr[i] = xr[i] + o[i];
            orc_rmsnorm(xr, (const float*)L[ORC_L_FFN_NORM].data.data(), D, c.rms_eps, h.data());
            q8.resize(D); qd.resize(D / 256); qb.resize(D / 32);
            orc_q8k_quantize(h.data(), D, q8.data(), qd.data(), qb.data());
            mat(L[ORC_L_GATE], c.ks_gateup, h.data(), q8.data(), qd.data(), qb.data(), gate.data());
            mat(L[ORC_L_UP], c.ks_gateup, h.data(), q8.data(), qd.data(), qb.data(), up.data());
            for (int i = 0; i < FF; ++i) act[i] = tk_siluf(gate[i]) * up[i];
            matvec(L[ORC_L_DOWN], c.ks_down, act.data(), o.data(), q8, qd, qb);
            for (int i = 0; i < D; ++i) xr[i] = xr[i] + o[i];
        }
    }
    for (int r = 0; r < n_rows; ++r) {
        orc_rmsnorm(&x[(size_t)r * D], (const float*)m->out_norm.data.data(), D, c.rms_eps, h.data());
        matvec(m->output, c.ks_out, h.data(), lg.data(), q8, qd, qb);
        if (logits) memcpy(logits + (size_t)r * c.vocab, lg.data(), sizeof(float) * c.vocab);
        if (argmax) {
            int best = 0;
            for (int i = 1; i < c.vocab; ++i) if (lg[i] > lg[best]) best = i;
            argmax[r] = best;
        }
    }
}

/* test hooks: write / read cache rows [pos0, pos0 + n_pos) of one (layer, sequence) as f16 bits in [position][kv head][dim] order, so a
 * long-context attention case needs one decode step instead of a whole prefill (the product's twin: tk_mi355x_llm_session_kv_write) */
void orc_llm_kv_write(orc_llm* m, int layer, int seq, int pos0, int n_pos, const uint16_t* k, const uint16_t* v) {
    const orc_llm_config_t& c = m->cfg;
    const size_t row = (size_t)c.n_kv_head * c.head_dim;
    const size_t base = (((size_t)layer * c.max_seq + seq) * c.max_ctx + pos0) * row;
    memcpy(&m->kcache[base], k, (size_t)n_pos * row * 2);
    memcpy(&m->vcache[base], v, (size_t)n_pos * row * 2);
}

void orc_llm_kv_read(orc_llm* m, int layer, int seq, int pos0, int n_pos, uint16_t* k, uint16_t* v) {
    const orc_llm_config_t& c = m->cfg;
    const size_t row = (size_t)c.n_kv_head * c.head_dim;
    const size_t base = (((size_t)layer * c.max_seq + seq) * c.max_ctx + pos0) * row;
    memcpy(k, &m->kcache[base], (size_t)n_pos * row * 2);
    memcpy(v, &m->vcache[base], (size_t)n_pos * row * 2);
}

/* The stochastic sampler (test infrastructure, like the rest of this file).  What it restates: the chain the reference installs with
 * llama_sampling_default_params() and runs in llama_sampling_sample (/root/reference/src/ai_models/tk_runner_lifecycle.c:76-77,
 * tk_runner_streaming.c:60-61, seed tk_runner_lifecycle.c:49): top-k, top-p, min-p, temperature, one draw.  llama.cpp is an empty,
 * un-pinned submodule of the reference: its generator (std::mt19937 + std::discrete_distribution) cannot be reproduced bit for bit —
 * parity unpinned vs llama.cpp.  The arithmetic below is the canonical order the HIP kernel (k_argmax / sample_row) follows:
 *   candidates: allowed tokens ordered by (logit descending under the order-preserving integer key, id ascending), the first
 *               K = min(top_k or 64, 64, allowed);
 *   p_i = exp(l_i - l_0) / S, S summed in candidate order;  top-p: shortest prefix whose running sum of p reaches top_p;
 *   min-p: drop the tail with p_i < min_p * p_0;  w_i = exp((l_i - l_0) / temp);  u = (splitmix64(seed, counter) >> 40) * 2^-24;
 *   the first i whose running sum of w exceeds u * W (W summed in candidate order), else the last candidate.
 * allow: (vocab + 31) / 32 words or NULL.  temp <= 0: the arg max over the allowed tokens (first index). */
static uint32_t orc_sample_key(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

int32_t orc_sample_row(const float* logits, int vocab, const uint32_t* allow, float temp, int32_t top_k, float top_p, float min_p, uint64_t seed,
                       uint32_t counter) {
    std::vector<int32_t> ids;
    for (int i = 0; i < vocab; ++i)
        if (!allow || ((allow[i >> 5] >> (i & 31)) & 1u)) ids.push_back(i);
    if (ids.empty()) return 0;
    if (!(temp > 0.0f)) {
        int32_t best = ids[0];
        for (int32_t i : ids) if (logits[i] > logits[best]) best = i;
        return best;
    }
    int K = top_k > 0 ? top_k : 64;
    if (K > 64) K = 64;
    if (K > (int)ids.size()) K = (int)ids.size();
    std::partial_sort(ids.begin(), ids.begin() + K, ids.end(), [&](int32_t a, int32_t b) {
        const uint32_t ka = orc_sample_key(logits[a]), kb = orc_sample_key(logits[b]);
        return ka != kb ? ka > kb : a < b;
    });
    const float l0 = logits[ids[0]];
    float S = 0.0f;
    for (int i = 0; i < K; ++i) S = S + tk_expf(logits[ids[i]] - l0);
    int n = K;
    if (top_p < 1.0f) {
        float c = 0.0f;
        for (int i = 0; i < K; ++i) {
            c = c + tk_divf(tk_expf(logits[ids[i]] - l0), S);
            if (c >= top_p) { n = i + 1; break; }
        }
    }
    if (min_p > 0.0f) {
        const float thr = min_p * tk_divf(1.0f, S);
        while (n > 1 && tk_divf(tk_expf(logits[ids[n - 1]] - l0), S) < thr) --n;
    }
    float W = 0.0f;
    for (int i = 0; i < n; ++i) W = W + tk_expf(tk_divf(logits[ids[i]] - l0, temp));
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)counter + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
    const float target = u * W;
    float c = 0.0f;
    for (int i = 0; i < n; ++i) {
        c = c + tk_expf(tk_divf(logits[ids[i]] - l0, temp));
        if (c > target) return ids[i];
    }
    return ids[n - 1];
}

void orc_llm_reset(orc_llm* m) {
    memset(m->kcache.data(), 0, m->kcache.size() * 2);
    memset(m->vcache.data(), 0, m->vcache.size() * 2);
}

} /* extern "C" */

extern "C" float orc_sqrtf(float x) { return tk_sqrtf(x); }
