/*
 * tk_sample_device.h — the canonical stochastic token draw as device code (HIP sources only): top-k by radix select over
 * (order-preserving logit key, ~id), top-p, min-p, temperature, one draw from splitmix64(seed, counter).  One workgroup per row.
 * Restated on the CPU in oracle/tk_oracle_llm.cpp (orc_sample_row); the chain is the one the reference installs with
 * llama_sampling_default_params() (src/ai_models/tk_runner_lifecycle.c:76-77) and whisper.cpp's temperature fallback draws from.
 */
#ifndef TK_SAMPLE_DEVICE_H
#define TK_SAMPLE_DEVICE_H

#include <hip/hip_runtime.h>

#include "tk_exact_math.h"
#include "tk_sample.h"

#ifndef TK_WAVE
#define TK_WAVE 64
#endif

__device__ __forceinline__ uint32_t sample_key(float f) { /* unsigned order == float order (-0 < +0) */
    const uint32_t u = tk_f32_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ void sample_row(const float* lg, int vocab, const uint32_t* allow, const TkSampleRow& sp, int32_t* out_id, uint32_t* lds_u32) {
    /* lds_u32: 256 histogram bins + 8 control words + 64 x (key, id) + 64 x (logit, id) sorted */
    uint32_t* hist = lds_u32;
    uint32_t* ctl = lds_u32 + 256; /* [0] digit found, [1] k left, [2] list count, [3] allowed count */
    uint32_t* lkey = ctl + 8;      /* [64] keys as collected */
    uint32_t* lid = lkey + 64;     /* [64] ids as collected */
    float* slog = (float*)(lid + 64); /* [64] logits, sorted */
    uint32_t* sid = (uint32_t*)(slog + 64); /* [64] ids, sorted */
    const int t = threadIdx.x, nthr = blockDim.x;
    auto allowed = [&](int i) { return !allow || ((allow[i >> 5] >> (i & 31)) & 1u); };
    if (t < 8) ctl[t] = 0;
    __syncthreads();
    { /* how many tokens may be drawn at all */
        uint32_t n = 0;
        for (int i = t; i < vocab; i += nthr) n += allowed(i) ? 1u : 0u;
        for (int s = 32; s >= 1; s >>= 1) n += __shfl_xor(n, s, TK_WAVE);
        if ((t & 63) == 0) atomicAdd(&ctl[3], n);
    }
    __syncthreads();
    int K = sp.top_k > 0 ? sp.top_k : TK_SAMPLE_MAX_K;
    K = K > TK_SAMPLE_MAX_K ? TK_SAMPLE_MAX_K : K;
    K = K > (int)ctl[3] ? (int)ctl[3] : K;
    if (K <= 0) { if (t == 0) *out_id = 0; return; } /* nothing allowed: cannot happen behind a grammar mask, which always allows something */
    /* composite key = (key << 32) | ~id: the K-th largest composite is unique.  Rounds: key bits 31..0, then id bits 15..0 (vocab <= 65536) */
    uint32_t pre_key = 0, pre_id = 0; /* decided prefixes */
    int kleft = K;
    for (int round = 0; round < 6; ++round) {
        for (int i = t; i < 256; i += nthr) hist[i] = 0;
        __syncthreads();
        const int shift = round < 4 ? 24 - 8 * round : 8 - 8 * (round - 4);
        for (int i = t; i < vocab; i += nthr) {
            if (!allowed(i)) continue;
            const uint32_t k = sample_key(lg[i]), ni = (~(uint32_t)i) & 0xFFFFu;
            bool in;
            uint32_t digit;
            if (round < 4) { in = round == 0 || (k >> (shift + 8)) == (pre_key >> (shift + 8)); digit = (k >> shift) & 255u; }
            else { in = k == pre_key && (round == 4 || (ni >> 8) == (pre_id >> 8)); digit = (ni >> shift) & 255u; }
            if (in) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        if (t == 0) { /* the digit whose bucket holds the kleft-th largest of what is still in */
            uint32_t above = 0;
            int d = 255;
            for (; d > 0; --d) {
                if (above + hist[d] >= (uint32_t)kleft) break;
                above += hist[d];
            }
            ctl[0] = (uint32_t)d;
            ctl[1] = (uint32_t)kleft - above;
        }
        __syncthreads();
        if (round < 4) pre_key |= ctl[0] << shift; else pre_id |= ctl[0] << shift;
        kleft = (int)ctl[1];
        __syncthreads();
    }
    /* everything at or above the threshold composite: exactly K entries */
    for (int i = t; i < vocab; i += nthr) {
        if (!allowed(i)) continue;
        const uint32_t k = sample_key(lg[i]), ni = (~(uint32_t)i) & 0xFFFFu;
        if (k > pre_key || (k == pre_key && ni >= pre_id)) {
            const uint32_t slot = atomicAdd(&ctl[2], 1u);
            if (slot < TK_SAMPLE_MAX_K) { lkey[slot] = k; lid[slot] = (uint32_t)i; }
        }
    }
    __syncthreads();
    if (t < K) { /* rank by comparison: descending key, ascending id */
        const uint32_t k = lkey[t], id = lid[t];
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += (lkey[j] > k || (lkey[j] == k && lid[j] < id)) ? 1 : 0;
        slog[rank] = lg[id];
        sid[rank] = id;
    }
    __syncthreads();
    if (t == 0) {
        const float l0 = slog[0];
        float sum = 0.0f;
        for (int i = 0; i < K; ++i) sum = sum + tk_expf(slog[i] - l0);
        int n = K;
        if (sp.top_p < 1.0f) {
            float c = 0.0f;
            for (int i = 0; i < K; ++i) {
                c = c + tk_divf(tk_expf(slog[i] - l0), sum);
                if (c >= sp.top_p) { n = i + 1; break; }
            }
        }
        if (sp.min_p > 0.0f) {
            const float thr = sp.min_p * tk_divf(1.0f, sum); /* p_0 = exp(0) / sum */
            while (n > 1 && tk_divf(tk_expf(slog[n - 1] - l0), sum) < thr) --n;
        }
        float W = 0.0f;
        for (int i = 0; i < n; ++i) W = W + tk_expf(tk_divf(slog[i] - l0, sp.temp));
        uint64_t z = sp.seed + 0x9E3779B97F4A7C15ull * ((uint64_t)sp.counter + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f; /* 24 bits * 2^-24: exact */
        const float target = u * W;
        int pick = n - 1;
        float c = 0.0f;
        for (int i = 0; i < n; ++i) {
            c = c + tk_expf(tk_divf(slog[i] - l0, sp.temp));
            if (c > target) { pick = i; break; }
        }
        *out_id = (int32_t)sid[pick];
    }
}


#endif
