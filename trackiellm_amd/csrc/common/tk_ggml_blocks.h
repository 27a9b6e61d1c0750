/*
 * tk_ggml_blocks.h — on-disk (GGUF) block formats of the ggml k-quants used by
 * Mistral-7B Q4_K_M, plus the deterministic quantisers this build uses to make
 * synthetic checkpoints and test GGUFs.
 *
 * The formats are the public ggml layouts (third-party: ggml-org/llama.cpp,
 * un-pinned and absent from /root/reference — SURVEY.md §0 F1, §8c).  The
 * reference only ever passes the file path to llama.cpp
 * (src/ai_models/tk_model_loader.c:245-251), so these structs are restated from
 * the published format description, not from reference code:
 *   Q4_K: 256 weights / 144 B: f16 d, f16 dmin, 12 B of packed 6-bit
 *         (scale, min) pairs for 8 sub-blocks of 32, 128 B of 4-bit quants.
 *         w = d*sc[j]*q - dmin*m[j]
 *   Q6_K: 256 weights / 210 B: 128 B low nibbles, 64 B high 2-bit pairs,
 *         16 int8 scales (groups of 16), f16 d.   w = d*sc[g]*(q-32)
 * Host + device code (the quantisers run inside the synthetic-weight kernel
 * and inside the oracle; they are bit-identical by construction).
 */
#ifndef TK_GGML_BLOCKS_H
#define TK_GGML_BLOCKS_H

#include "tk_exact_math.h"

#define TK_QK_K 256

enum tk_ggml_type {
    TK_TYPE_F32 = 0,
    TK_TYPE_F16 = 1,
    TK_TYPE_Q4_K = 12,
    TK_TYPE_Q6_K = 14,
};

typedef struct {
    uint16_t d;
    uint16_t dmin;
    uint8_t scales[12];
    uint8_t qs[128];
} tk_block_q4_K; /* 144 B */

typedef struct {
    uint8_t ql[128];
    uint8_t qh[64];
    int8_t scales[16];
    uint16_t d;
} tk_block_q6_K; /* 210 B */

TK_HD size_t tk_type_block_bytes(int type) {
    return type == TK_TYPE_Q4_K ? 144 : type == TK_TYPE_Q6_K ? 210 : type == TK_TYPE_F16 ? 2 : 4;
}
TK_HD size_t tk_type_block_elems(int type) {
    return (type == TK_TYPE_Q4_K || type == TK_TYPE_Q6_K) ? 256 : 1;
}

/* 6-bit (scale, min) pair j of a Q4_K block */
TK_HD void tk_q4k_get_scale_min(int j, const uint8_t* q, uint8_t* sc, uint8_t* m) {
    if (j < 4) {
        *sc = q[j] & 63;
        *m = q[j + 4] & 63;
    } else {
        *sc = (uint8_t)((q[j + 4] & 0x0F) | ((q[j - 4] >> 6) << 4));
        *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4));
    }
}

TK_HD void tk_q4k_set_scale_min(int j, uint8_t* q, uint8_t sc, uint8_t m) {
    if (j < 4) {
        q[j] = (uint8_t)((q[j] & 0xC0) | sc);
        q[j + 4] = (uint8_t)((q[j + 4] & 0xC0) | m);
    } else {
        q[j + 4] = (uint8_t)((sc & 0x0F) | ((m & 0x0F) << 4));
        q[j - 4] = (uint8_t)((q[j - 4] & 0x3F) | ((sc >> 4) << 6));
        q[j] = (uint8_t)((q[j] & 0x3F) | ((m >> 4) << 6));
    }
}

/* weight i (0..255) of a Q4_K block as the integer triple the dot product uses */
TK_HD int tk_q4k_quant(const tk_block_q4_K* b, int i) {
    int c = i >> 6;          /* 64-weight chunk */
    int r = i & 63;
    uint8_t byte = b->qs[c * 32 + (r & 31)];
    return (r < 32) ? (byte & 0x0F) : (byte >> 4);
}

TK_HD float tk_q4k_dequant(const tk_block_q4_K* b, int i) {
    uint8_t sc, m;
    tk_q4k_get_scale_min(i >> 5, b->scales, &sc, &m);
    float d = tk_f16_to_f32(b->d), dmin = tk_f16_to_f32(b->dmin);
    return (d * (float)sc) * (float)tk_q4k_quant(b, i) - dmin * (float)m;
}

/* weight i (0..255) of a Q6_K block, q in [0,63] (the stored value, before -32) */
TK_HD int tk_q6k_quant(const tk_block_q6_K* b, int i) {
    int n = i >> 7;          /* 128-weight half */
    int r = i & 127;
    int l = r & 31;
    int quarter = r >> 5;    /* 0..3 */
    uint8_t qlb = b->ql[n * 64 + (quarter & 1) * 32 + l];
    int lo = (quarter < 2) ? (qlb & 0x0F) : (qlb >> 4);
    int hi = (b->qh[n * 32 + l] >> (2 * quarter)) & 3;
    return lo | (hi << 4);
}

TK_HD float tk_q6k_dequant(const tk_block_q6_K* b, int i) {
    float d = tk_f16_to_f32(b->d);
    return (d * (float)b->scales[i >> 4]) * (float)(tk_q6k_quant(b, i) - 32);
}

TK_HD void tk_q6k_set_quant(tk_block_q6_K* b, int i, int q) {
    int n = i >> 7, r = i & 127, l = r & 31, quarter = r >> 5;
    uint8_t* qlb = &b->ql[n * 64 + (quarter & 1) * 32 + l];
    if (quarter < 2) *qlb = (uint8_t)((*qlb & 0xF0) | (q & 0x0F));
    else *qlb = (uint8_t)((*qlb & 0x0F) | ((q & 0x0F) << 4));
    uint8_t* qhb = &b->qh[n * 32 + l];
    *qhb = (uint8_t)((*qhb & ~(3 << (2 * quarter))) | ((q >> 4) << (2 * quarter)));
}

/*
 * Deterministic min/max quantisers ("the build's own Q4_K_M recipe").
 * Not llama.cpp's iterative search: one pass, IEEE ops only, so host and device
 * produce identical blocks.
 */
TK_HD void tk_quantize_q4_K(const float* x, tk_block_q4_K* out) {
    float scales[8], mins[8];
    float max_scale = 0.0f, max_min = 0.0f;
    for (int j = 0; j < 8; ++j) {
        float mn = x[32 * j], mx = x[32 * j];
        for (int i = 1; i < 32; ++i) {
            float v = x[32 * j + i];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        if (mn > 0.0f) mn = 0.0f;
        scales[j] = tk_divf(mx - mn, 15.0f);
        if (scales[j] < 0.0f) scales[j] = 0.0f;
        mins[j] = -mn;
        max_scale = scales[j] > max_scale ? scales[j] : max_scale;
        max_min = mins[j] > max_min ? mins[j] : max_min;
    }
    float d = tk_divf(max_scale, 63.0f), dmin = tk_divf(max_min, 63.0f);
    out->d = tk_f32_to_f16(d);
    out->dmin = tk_f32_to_f16(dmin);
    float dq = tk_f16_to_f32(out->d), dminq = tk_f16_to_f32(out->dmin);
    for (int k = 0; k < 12; ++k) out->scales[k] = 0;
    for (int k = 0; k < 128; ++k) out->qs[k] = 0;
    for (int j = 0; j < 8; ++j) {
        int sc = dq > 0.0f ? (int)tk_rintf(tk_divf(scales[j], dq)) : 0;
        int m = dminq > 0.0f ? (int)tk_rintf(tk_divf(mins[j], dminq)) : 0;
        sc = sc > 63 ? 63 : sc;
        m = m > 63 ? 63 : m;
        tk_q4k_set_scale_min(j, out->scales, (uint8_t)sc, (uint8_t)m);
        float dl = dq * (float)sc, ml = dminq * (float)m;
        for (int i = 0; i < 32; ++i) {
            int q = 0;
            if (dl > 0.0f) {
                q = (int)tk_rintf(tk_divf(x[32 * j + i] + ml, dl));
                q = q < 0 ? 0 : (q > 15 ? 15 : q);
            }
            int idx = 32 * j + i;
            int c = idx >> 6, r = idx & 63;
            uint8_t* byte = &out->qs[c * 32 + (r & 31)];
            if (r < 32) *byte = (uint8_t)((*byte & 0xF0) | q);
            else *byte = (uint8_t)((*byte & 0x0F) | (q << 4));
        }
    }
}

TK_HD void tk_quantize_q6_K(const float* x, tk_block_q6_K* out) {
    float gscale[16];
    float max_abs_scale = 0.0f;
    for (int g = 0; g < 16; ++g) {
        float amax = 0.0f;
        for (int i = 0; i < 16; ++i) {
            float a = tk_fabsf(x[16 * g + i]);
            amax = a > amax ? a : amax;
        }
        gscale[g] = tk_divf(amax, 31.0f);
        max_abs_scale = gscale[g] > max_abs_scale ? gscale[g] : max_abs_scale;
    }
    float d = tk_divf(max_abs_scale, 127.0f);
    out->d = tk_f32_to_f16(d);
    float dq = tk_f16_to_f32(out->d);
    for (int k = 0; k < 128; ++k) out->ql[k] = 0;
    for (int k = 0; k < 64; ++k) out->qh[k] = 0;
    for (int g = 0; g < 16; ++g) {
        int sc = dq > 0.0f ? (int)tk_rintf(tk_divf(gscale[g], dq)) : 0;
        sc = sc > 127 ? 127 : sc;
        out->scales[g] = (int8_t)sc;
        float dl = dq * (float)sc;
        for (int i = 0; i < 16; ++i) {
            int q = 32;
            if (dl > 0.0f) {
                q = (int)tk_rintf(tk_divf(x[16 * g + i], dl)) + 32;
                q = q < 0 ? 0 : (q > 63 ? 63 : q);
            }
            tk_q6k_set_quant(out, 16 * g + i, q);
        }
    }
}

/* ---- seeded synthetic tensors (splitmix64, seed stated per item) ---- */

TK_HD uint64_t tk_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* counter-based: element `index` of stream (seed, tensor_id). Irwin-Hall(4) ~ N(0,1) */
TK_HD float tk_synth_normal(uint64_t seed, uint64_t tensor_id, uint64_t index) {
    uint64_t h = tk_splitmix64(tk_splitmix64(seed ^ (tensor_id * 0xD6E8FEB86659FD93ull)) + index);
    int32_t s = (int32_t)(h & 0xffff) + (int32_t)((h >> 16) & 0xffff) +
                (int32_t)((h >> 32) & 0xffff) + (int32_t)((h >> 48) & 0xffff);
    /* sum of 4 U[0,65535]: mean 131070, std 65536/sqrt(3) */
    return (float)(s - 131070) * 2.64289216e-5f; /* sqrt(3)/65536 */
}

TK_HD uint32_t tk_synth_u32(uint64_t seed, uint64_t tensor_id, uint64_t index) {
    uint64_t h = tk_splitmix64(tk_splitmix64(seed ^ (tensor_id * 0xD6E8FEB86659FD93ull)) + index);
    return (uint32_t)(h >> 32);
}

#endif /* TK_GGML_BLOCKS_H */
