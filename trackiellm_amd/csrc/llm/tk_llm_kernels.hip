/*
 * tk_llm_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Mistral-7B GGUF decode path
 * that the reference hands to llama.cpp (src/ai_models/tk_runner_streaming.c:34,77 `llama_decode`).
 *
 * Numerics contract = oracle/tk_oracle_llm.cpp: int8 (Q8_K-style) activations, integer dot
 * inside a 256-block, float accumulation across blocks in ascending order, canonical
 * reduction trees elsewhere.  Results are bit-identical to the oracle.
 *
 * Roofline: the dominant kernels are k_gemv_w4a8 (passes of <= 32 rows), k_gemm_w4a8 (33..192 rows) and k_gemm32_w4a8 (193..256 rows):
 * weights streamed once per pass, 0.5625 / 0.8203 B per weight; MFMA carries the int8 contraction
 * [16 row slots x 64 k] x [64 k x 16 weight rows] (32 x 32 x 32 in the widest kernel), so one weight pass serves up to 256
 * (sequence, position) rows.  Measured fractions and where the time goes: DESIGN.md §7.
 */
#include "tk_llm_kernels.h"

#include "../common/tk_exact_math.h"
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../common/tk_ggml_blocks.h"
#include "../common/tk_sample_device.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define TK_WAVE 64

/* ------------------------------------------------------------------------------------------
 * synthetic checkpoints (no real weights offline) — generated directly in HBM
 * ------------------------------------------------------------------------------------------ */
__global__ void k_synth_blocks(int type, uint64_t seed, uint64_t tid, int64_t nblocks, float scale, uint8_t* out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    float x[256];
    for (int i = 0; i < 256; ++i) x[i] = scale * tk_synth_normal(seed, tid, (uint64_t)(256 * b + i));
    if (type == TK_TYPE_Q4_K) {
        tk_block_q4_K blk;
        tk_quantize_q4_K(x, &blk);
        ((tk_block_q4_K*)out)[b] = blk;
    } else {
        tk_block_q6_K blk;
        tk_quantize_q6_K(x, &blk);
        ((tk_block_q6_K*)out)[b] = blk;
    }
}

__global__ void k_synth_f32(uint64_t seed, uint64_t tid, int64_t n, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0f + 0.1f * tk_synth_normal(seed, tid, (uint64_t)i);
}

__global__ void k_synth_f16(uint64_t seed, uint64_t tid, int64_t n, float scale, uint16_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tk_f32_to_f16(scale * tk_synth_normal(seed, tid, (uint64_t)i));
}
void tk_launch_synth_f16(uint64_t seed, uint64_t tensor_id, int64_t n, float scale, uint16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_synth_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, tensor_id, n, scale, out);
}

void tk_launch_synth_blocks(int type, uint64_t seed, uint64_t tensor_id, int64_t nblocks, float scale, void* out, hipStream_t s) {
    int64_t grid = (nblocks + 63) / 64;
    hipLaunchKernelGGL(k_synth_blocks, dim3((unsigned)grid), dim3(64), 0, s, type, seed, tensor_id, nblocks, scale, (uint8_t*)out);
}
void tk_launch_synth_f32(uint64_t seed, uint64_t tensor_id, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_synth_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, tensor_id, n, out);
}

/* ------------------------------------------------------------------------------------------
 * LoRA merge at load (tk_lora.h; the reference applies an adapter once, in place, right after the model is loaded:
 * src/ai_models/tk_model_loader.c:259-270).  W' = W + scale (B A), per 256-weight block of a row:
 *     delta[k] = fma chain over j = 0 .. r-1 of B[n][j] * A[j][k] (from 0.0f),   w'[k] = w[k] + scale * delta[k]   (multiply, then add)
 * and the block is quantised back to its own type by the build's block quantiser (a quantised destination without a base model: what
 * ggml's add does on one).  One wave per block: every lane finishes four weights, lane 0 quantises.  Load-time only.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(64) void k_lora_merge(int type, void* blocks, int64_t K, const float* A, const float* B, int r, float scale) {
    __shared__ float x[256];
    const int64_t nblk = K / 256, b = blockIdx.x, n = b / nblk, k0 = (b % nblk) * 256;
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) {
        float w;
        if (type == TK_TYPE_Q4_K) w = tk_q4k_dequant((const tk_block_q4_K*)blocks + b, e);
        else if (type == TK_TYPE_Q6_K) w = tk_q6k_dequant((const tk_block_q6_K*)blocks + b, e);
        else w = tk_f16_to_f32(((const uint16_t*)blocks)[b * 256 + e]);
        float delta = 0.0f;
        for (int j = 0; j < r; ++j) delta = tk_fmaf(B[n * r + j], A[(int64_t)j * K + k0 + e], delta);
        x[e] = w + scale * delta;
    }
    __syncthreads();
    if (type == TK_TYPE_F16) {
        for (int e = lane; e < 256; e += 64) ((uint16_t*)blocks)[b * 256 + e] = tk_f32_to_f16(x[e]);
    } else if (lane == 0) {
        if (type == TK_TYPE_Q4_K) { tk_block_q4_K blk; tk_quantize_q4_K(x, &blk); ((tk_block_q4_K*)blocks)[b] = blk; }
        else { tk_block_q6_K blk; tk_quantize_q6_K(x, &blk); ((tk_block_q6_K*)blocks)[b] = blk; }
    }
}
bool tk_launch_lora_merge(int type, void* blocks, int64_t rows, int64_t K, const float* A, const float* B, int r, float scale, hipStream_t s) {
    if ((type != TK_TYPE_Q4_K && type != TK_TYPE_Q6_K && type != TK_TYPE_F16) || K % 256 || rows < 1 || r < 1 || rows * (K / 256) > 0x7fffffffLL) return false;
    hipLaunchKernelGGL(k_lora_merge, dim3((unsigned)(rows * (K / 256))), dim3(64), 0, s, type, blocks, K, A, B, r, scale);
    return true;
}

/* ------------------------------------------------------------------------------------------
 * GGUF blocks -> MFMA-fragment tiles (tk_llm_layout.h).  One wave per tile.
 * ------------------------------------------------------------------------------------------ */
__global__ void k_repack_q4k(const tk_block_q4_K* src, int64_t nblk, uint8_t* tiles) {
    const int lane = threadIdx.x;
    const int n = lane & 15, g = lane >> 4;
    const int64_t rt = blockIdx.y, blk = blockIdx.x;
    const tk_block_q4_K* b = src + (rt * 16 + n) * nblk + blk;
    uint8_t* tile = tiles + (rt * nblk + blk) * TK_Q4K_TILE_BYTES;
    for (int i = 0; i < 2; ++i) {
        uint32_t dw[4];
        for (int s = 0; s < 4; ++s) {
            int k0 = 32 * (4 * i + s) + 8 * g;
            uint32_t v = 0;
            for (int t = 0; t < 4; ++t) v |= (uint32_t)(tk_q4k_quant(b, k0 + t) | (tk_q4k_quant(b, k0 + 4 + t) << 4)) << (8 * t);
            dw[s] = v;
        }
        *(uint4*)(tile + 1024 * i + lane * 16) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
    }
    if (g == 0) {
        uint32_t h[4];
        h[0] = (uint32_t)b->d | ((uint32_t)b->dmin << 16);
        for (int k = 0; k < 3; ++k)
            h[1 + k] = (uint32_t)b->scales[4 * k] | ((uint32_t)b->scales[4 * k + 1] << 8) | ((uint32_t)b->scales[4 * k + 2] << 16) |
                       ((uint32_t)b->scales[4 * k + 3] << 24);
        *(uint4*)(tile + 2048 + n * 16) = make_uint4(h[0], h[1], h[2], h[3]);
    }
}

__global__ void k_repack_q6k(const tk_block_q6_K* src, int64_t nblk, uint8_t* tiles) {
    const int lane = threadIdx.x;
    const int n = lane & 15, g = lane >> 4;
    const int64_t rt = blockIdx.y, blk = blockIdx.x;
    const tk_block_q6_K* b = src + (rt * 16 + n) * nblk + blk;
    uint8_t* tile = tiles + (rt * nblk + blk) * TK_Q6K_TILE_BYTES;
    uint32_t hi[4] = {0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) {
        uint32_t dw[4];
        for (int s = 0; s < 4; ++s) {
            int j = 4 * i + s;
            int k0 = 32 * j + 8 * g;
            uint32_t v = 0;
            for (int t = 0; t < 4; ++t) {
                int qa = (tk_q6k_quant(b, k0 + t) ^ 32);     /* 6-bit two's complement of q-32 */
                int qb = (tk_q6k_quant(b, k0 + 4 + t) ^ 32);
                v |= (uint32_t)((qa & 15) | ((qb & 15) << 4)) << (8 * t);
                int u = j >> 1, e = j & 1;
                hi[u] |= (uint32_t)(qa >> 4) << (8 * t + 2 * (2 * e));
                hi[u] |= (uint32_t)(qb >> 4) << (8 * t + 2 * (2 * e + 1));
            }
            dw[s] = v;
        }
        *(uint4*)(tile + 1024 * i + lane * 16) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
    }
    *(uint4*)(tile + 2048 + lane * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if (g == 0) {
        uint32_t sc[4];
        for (int k = 0; k < 4; ++k)
            sc[k] = (uint32_t)(uint8_t)b->scales[4 * k] | ((uint32_t)(uint8_t)b->scales[4 * k + 1] << 8) |
                    ((uint32_t)(uint8_t)b->scales[4 * k + 2] << 16) | ((uint32_t)(uint8_t)b->scales[4 * k + 3] << 24);
        *(uint4*)(tile + 3072 + n * 16) = make_uint4(sc[0], sc[1], sc[2], sc[3]);
        *(uint16_t*)(tile + 3328 + n * 2) = b->d;
    }
}

void tk_launch_repack(int type, const void* blocks, int64_t rows, int64_t K, uint8_t* tiles, hipStream_t s) {
    dim3 grid((unsigned)(K / 256), (unsigned)(rows / 16));
    if (type == TK_TYPE_Q4_K) hipLaunchKernelGGL(k_repack_q4k, grid, dim3(64), 0, s, (const tk_block_q4_K*)blocks, K / 256, tiles);
    else hipLaunchKernelGGL(k_repack_q6k, grid, dim3(64), 0, s, (const tk_block_q6_K*)blocks, K / 256, tiles);
}

/* ------------------------------------------------------------------------------------------
 * token embedding: one Q4_K row (GGUF layout) de-quantised per row slot
 * ------------------------------------------------------------------------------------------ */
__global__ void k_embed(const void* embd, int type, int D, const int32_t* tok, float* x) {
    const int r = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    if (type == TK_TYPE_F16) {
        x[(int64_t)r * D + i] = tk_f16_to_f32(((const uint16_t*)embd)[(int64_t)tok[r] * D + i]);
    } else {
        const tk_block_q4_K* row = (const tk_block_q4_K*)embd + (int64_t)tok[r] * (D / 256);
        x[(int64_t)r * D + i] = tk_q4k_dequant(row + i / 256, i % 256);
    }
}

void tk_launch_embed(const void* embd, int type, int D, const int32_t* tok, int nrows, float* x, hipStream_t s) {
    hipLaunchKernelGGL(k_embed, dim3((D + 255) / 256, nrows), dim3(256), 0, s, embd, type, D, tok, x);
}

/* ------------------------------------------------------------------------------------------
 * shared device helpers
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ float wave_xor_f(float v, int mask) { return __shfl_xor(v, mask, TK_WAVE); }

/* canonical sum of 256 per-thread partials (oracle: orc_sum256) */
__device__ __forceinline__ float block_sum256(float v, float* red /* >= 4 floats LDS */) {
    for (int s = 32; s >= 1; s >>= 1) v = v + wave_xor_f(v, s);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return r;
}

/*
 * Quantise one 8-element chunk (chunk index c of the row) to the aq/ad/abs images.
 * Must be called by all 32 lanes of a half-wave that together hold one 256-block
 * (lane order == chunk order), so the block amax / sub-block sums come from shuffles.
 */
__device__ __forceinline__ void quantize_chunk8(const float* v, int c, int slot, TkActQ8 out) {
    if (out.af) { /* the f16-weight matmuls' input: the same values rounded through f16, in the tiled GEMM's operand-image order (csrc/nn/tk_gemm_tiled.h) */
        float* dst = out.af + (size_t)(slot >> 4) * out.af_ts + (size_t)(c >> 1) * 256 + (size_t)(slot & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k16 = 8 * (c & 1) + i; /* position inside the group of 16: g = k16 % 4, t = k16 / 4 */
            dst[(k16 & 3) * 64 + (k16 >> 2)] = tk_f16_to_f32(tk_f32_to_f16(v[i]));
        }
    }
    /* ggml's quantize_row_q8_K_ref: the signed value of the block's FIRST element of largest magnitude (element index = 8 x chunk-in-block + i),
     * iscale = -127 / max, q = min(127, nearest_int(iscale * x)), d = 1 / iscale */
    float amax = 0.0f, mval = 0.0f;
    int midx = 8 * (c & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float ax = tk_fabsf(v[i]);
        if (ax > amax) { amax = ax; mval = v[i]; midx = 8 * (c & 31) + i; }
    }
#pragma unroll
    for (int s = 1; s <= 16; s <<= 1) {
        const float oa = wave_xor_f(amax, s), ov = wave_xor_f(mval, s);
        const int oi = __shfl_xor(midx, s, TK_WAVE);
        if (oa > amax || (oa == amax && oi < midx)) { amax = oa; mval = ov; midx = oi; }
    }
    const float iscale = amax > 0.0f ? tk_divf(-127.0f, mval) : 0.0f;
    int q[8];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int t = (int)tk_rintf(iscale * v[i]);
        t = t > 127 ? 127 : t;
        q[i] = t;
        sum += t;
    }
    uint32_t lo = (uint32_t)(q[0] & 255) | ((uint32_t)(q[1] & 255) << 8) | ((uint32_t)(q[2] & 255) << 16) | ((uint32_t)(q[3] & 255) << 24);
    uint32_t hi = (uint32_t)(q[4] & 255) | ((uint32_t)(q[5] & 255) << 8) | ((uint32_t)(q[6] & 255) << 16) | ((uint32_t)(q[7] & 255) << 24);
    const int mt = slot >> 4, sl = slot & 15; /* M-tile and slot inside it */
    /* image order [K/64][4 g][16 slot][2 sub-blocks][8]: c = 4 * sub-block + g; a lane's operand of one x64 MFMA is one 16-byte read */
    *(uint2*)(out.aq + mt * out.aq_ts + (size_t)(c >> 3) * 1024 + ((size_t)(c & 3) * TK_ROW_SLOTS + sl) * 16 + ((c >> 2) & 1) * 8) = make_uint2(lo, hi);
    sum += __shfl_xor(sum, 1, TK_WAVE);
    sum += __shfl_xor(sum, 2, TK_WAVE);
    if ((c & 3) == 0) { /* sub-block sum as two int8: sum = 64 * h + l, l in [0, 63], h in [-64, 63]  (|sum| <= 32 * 127) */
        int8_t* mb = out.abs + mt * out.abs_ts + (size_t)(c >> 5) * 256 + (size_t)sl * 8 + ((c >> 2) & 7);
        mb[0] = (int8_t)(sum & 63);
        mb[128] = (int8_t)(sum >> 6);
        /* and as two f16 numbers, sum = 2 hh + ll (|hh| <= 2032 and ll in {0, 1} are exact in f16): the batched kernel contracts them with
         * (2 m, m) in ONE f16 MFMA */
        if (out.abs16) { /* null: an image built in LDS by a mat-vec launch for itself (TkGemvArgs::fuse), which reads only (l, h) */
            uint16_t* hb = out.abs16 + mt * out.abs_ts + (size_t)(c >> 5) * 256 + (size_t)sl * 8 + ((c >> 2) & 7);
            hb[0] = tk_f32_to_f16((float)(sum >> 1));
            hb[128] = tk_f32_to_f16((float)(sum & 1));
        }
    }
    if ((c & 31) == 0) out.ad[mt * out.ad_ts + (size_t)(c >> 5) * TK_ROW_SLOTS + sl] = amax > 0.0f ? tk_divf(1.0f, iscale) : 0.0f;
}

/* ------------------------------------------------------------------------------------------
 * residual add (sum of K-split partials, ascending) + RMSNorm + Q8 quantise.  One WG per row.
 *
 * A row is 16 KB of residual stream plus up to 8 K-split slabs of 16 KB pulled through one CU, and at 16 rows per pass only 16 CUs
 * take part: the kernel is pure memory latency.  So the workgroup is as wide as the row has 16-byte groups (1024 threads for
 * d_model 4096) and every thread requests its residual group and ALL its slab groups before it adds the first: one round trip
 * instead of one per slab.  The arithmetic keeps the oracle's order (orc_rmsnorm): slabs added in ascending order, then the sum of
 * squares as 256 chains — chain t runs over the groups t, t + 256, ... element by element — joined by the canonical 256-sum; the
 * chains read the finished row back from LDS, so how many threads loaded it does not enter the result.
 * ------------------------------------------------------------------------------------------ */
#define TK_RMS_MAX_KS 8
__global__ __launch_bounds__(1024) void k_rmsnorm_q8(float* __restrict__ x, const float* __restrict__ partial, int ks, int n_total,
                                                      const float* __restrict__ w, float eps, int D, TkActQ8 out) {
    extern __shared__ float sh[]; /* D floats + 4 */
    float* hbuf = sh;
    float* red = sh + D;
    const int r = blockIdx.x, t = threadIdx.x, nthr = blockDim.x;
    float* xr = x + (int64_t)r * D;
    const int ngrp = D / 4;
    for (int g = t; g < ngrp; g += nthr) {
        v4f v = *(const v4f*)(xr + 4 * g);
        if (partial) {
            v4f p[TK_RMS_MAX_KS];
#pragma unroll
            for (int s = 0; s < TK_RMS_MAX_KS; ++s)
                if (s < ks) p[s] = *(const v4f*)(partial + ((int64_t)s * TK_MAX_ROWS + r) * n_total + 4 * g);
            v4f o = p[0];
#pragma unroll
            for (int s = 1; s < TK_RMS_MAX_KS; ++s)
                if (s < ks) o = o + p[s];
            v = v + o;
            *(v4f*)(xr + 4 * g) = v;
        }
        *(v4f*)(hbuf + 4 * g) = v;
    }
    /* norm weights of this thread's first chunk are requested before the reductions so their latency hides under them */
    const int nchunk = D / 8;
    v4f wa[2];
    if (t < nchunk) { wa[0] = *(const v4f*)(w + 8 * t); wa[1] = *(const v4f*)(w + 8 * t + 4); }
    __syncthreads();
    float ss = 0.0f;
    if (t < 256)
        for (int g = t; g < ngrp; g += 256) {
            const v4f v = *(const v4f*)(hbuf + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) ss = tk_fmaf(v[i], v[i], ss);
        }
    /* canonical sum of the 256 chains (block_sum256's arithmetic, carried by the first four waves of a wider workgroup) */
    for (int s = 32; s >= 1; s >>= 1) ss = ss + wave_xor_f(ss, s);
    if (t < 256 && (t & 63) == 0) red[t >> 6] = ss;
    __syncthreads();
    const float tot = ((red[0] + red[1]) + red[2]) + red[3];
    const float mean = tk_divf(tot, (float)D);
    const float scale = tk_divf(1.0f, tk_sqrtf(mean + eps));
    for (int c = t; c < nchunk; c += nthr) { /* nchunk and nthr are multiples of 32: half-waves enter together, as quantize_chunk8 needs */
        v4f w0, w1;
        if (c == t) { w0 = wa[0]; w1 = wa[1]; }
        else { w0 = *(const v4f*)(w + 8 * c); w1 = *(const v4f*)(w + 8 * c + 4); }
        const v4f h0 = *(const v4f*)(hbuf + 8 * c), h1 = *(const v4f*)(hbuf + 8 * c + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (h0[i] * scale) * w0[i];
            v[4 + i] = (h1[i] * scale) * w1[i];
        }
        quantize_chunk8(v, c, r, out);
    }
}

/* x += (p_0 + p_1 + ... + p_{ks-1}), exactly the residual update k_rmsnorm_q8 performs before it normalises: used where the
 * residual stream leaves this GPU (pipeline-stage boundary) and the next layer's norm runs elsewhere */
__global__ __launch_bounds__(256) void k_residual_fold(float* __restrict__ x, const float* __restrict__ partial, int ks, int n_total, int D) {
    const int r = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (4 * g >= D) return;
    v4f o = *(const v4f*)(partial + (int64_t)r * n_total + 4 * g);
    for (int s = 1; s < ks; ++s) o = o + *(const v4f*)(partial + ((int64_t)s * TK_MAX_ROWS + r) * n_total + 4 * g);
    v4f* xp = (v4f*)(x + (int64_t)r * D + 4 * g);
    *xp = *xp + o;
}

void tk_launch_residual_fold(float* x, const float* partial, int ks, int n_total, int D, int nrows, hipStream_t s) {
    hipLaunchKernelGGL(k_residual_fold, dim3((D / 4 + 255) / 256, nrows), dim3(256), 0, s, x, partial, ks, n_total, D);
}

void tk_launch_rmsnorm_q8(float* x, const float* partial, int ks, int n_total_partial, const float* w, float eps, int D, int nrows,
                          TkActQ8 out, hipStream_t s) {
    int nthr = D / 4; /* one thread per 16-byte group of the row, 256 .. 1024 */
    nthr = nthr < 256 ? 256 : nthr > 1024 ? 1024 : (nthr + 255) / 256 * 256;
    /* (round 6: four-wave workgroups while several decode streams share the device — so that the norm fits beside other streams' resident mat-vec
     * workgroups instead of waiting for one to retire — measured SLOWER: 3.99 against 3.90 ms per step and group at 3 x 16 rows, profiles/r06_northstar_explore.txt) */
    hipLaunchKernelGGL(k_rmsnorm_q8, dim3(nrows), dim3(nthr), (D + 4) * sizeof(float), s, x, partial, ks, n_total_partial, w, eps, D, out);
}

/* ------------------------------------------------------------------------------------------
 * W4A8 / W6A8 GEMV on MFMA:   out[ks][slot][n] = sum_{k in K-range ks} W[n][k] * a[slot][k]
 *
 * One workgroup per CU (k_gemv_w4a8 below spells out the mapping): wave w owns one 16-row weight tile run and walks its K-range
 * block by block; its weight stream is one contiguous run of tiles.  The int8 activations of the K-range (shared by the
 * waves) are staged once in LDS in MFMA A-operand order.  Per 256-block and wave: 3 x 1 KiB coalesced non-temporal loads
 * (Q4_K), the sub-block scales folded into the int8 B operand so the eight sub-blocks accumulate inside chained
 * 16x16x64 i8 MFMAs (Q6_K: half-masked 16x16x32 MFMAs + 24-bit integer mads), two fp32 FMAs per (row, weight row).
 * Algorithmic bytes per launch = rows * K * (144 | 210) / 256.
 * ------------------------------------------------------------------------------------------ */
struct FragQ4 { uint4 q0, q1, h; };
struct FragQ6 { uint4 q0, q1, qh, sc; uint32_t d; };

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg_nt(const uint8_t* p) {
    const v4u v = __builtin_nontemporal_load((const v4u*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ FragQ4 load_q4(const uint8_t* tile, int lane) {
    FragQ4 f;
    f.q0 = ldg_nt(tile + lane * 16);
    f.q1 = ldg_nt(tile + 1024 + lane * 16);
    f.h = ldg_nt(tile + 2048 + (lane & 15) * 16);
    return f;
}

__device__ __forceinline__ FragQ6 load_q6(const uint8_t* tile, int lane) {
    FragQ6 f;
    f.q0 = ldg_nt(tile + lane * 16);
    f.q1 = ldg_nt(tile + 1024 + lane * 16);
    f.qh = ldg_nt(tile + 2048 + lane * 16);
    f.sc = ldg_nt(tile + 3072 + (lane & 15) * 16);
    f.d = *(const uint16_t*)(tile + 3328 + (lane & 15) * 2);
    return f;
}

__device__ __forceinline__ float f16bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }

/* one weight tile x one 16-row M-tile; a pass with two M-tiles calls it twice per tile (the nibble unpack is repeated,
 * ~15 % more VALU, but the live register set stays that of a single tile: no spills at 4 tiles in flight) */
/* a * b + c for |a|, |b| < 2^23 (v_mad_i32_i24, full rate; a 32-bit v_mul_lo is quarter rate).  Every use multiplies a
 * 6/8-bit scale with one MFMA partial (|.| <= 32 * 127 * 15 for Q4_K, 16 * 127 * 128 for Q6_K), so it is exact. */
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

/* 6-bit (scale, min) pairs of a Q4_K header as byte vectors: sc/mn 0..3 in *_lo, 4..7 in *_hi (one byte each) */
__device__ __forceinline__ void q4k_scales(const uint4& h, uint32_t* sc_lo, uint32_t* sc_hi, uint32_t* mn_lo, uint32_t* mn_hi) {
    const uint32_t s0 = h.y, s1 = h.z, s2 = h.w;
    *sc_lo = s0 & 0x3F3F3F3Fu;
    *mn_lo = s1 & 0x3F3F3F3Fu;
    *sc_hi = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
    *mn_hi = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
}

typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
/* both 16-bit halves of q (four nibble bytes, each <= 15) times s <= 7: no byte carries, one v_pk_mul_lo_u16 */
__device__ __forceinline__ uint32_t pk_scale(uint32_t q, unsigned short s) {
    const v2u16 r = __builtin_bit_cast(v2u16, q) * (v2u16){s, s};
    return __builtin_bit_cast(uint32_t, r);
}

/*
 * One Q4_K weight tile (16 weight rows x 256 k) against MT 16-row M-tiles of int8 activations.
 *
 *   sum_k a_k w_k  =  d * sum_j sc_j (a . q)_j  -  dmin * sum_j mn_j bsum_j          (per weight row, 8 sub-blocks j)
 *
 * Both integer sums run on the matrix cores with no per-sub-block VALU work on the results:
 *  - the 6-bit sub-block scale is split sc = 8 sh + sl (3 bits each) and folded into the B operand: q * sl and q * sh stay
 *    inside int8 (<= 105), so P = 8 * sum_j (a . q sh_j) + sum_j (a . q sl_j) accumulates over all eight sub-blocks INSIDE
 *    two MFMA accumulators; two sub-blocks share one v_mfma_i32_16x16x64_i8 (the k order inside an MFMA is free as long as
 *    A and B agree, and they do: both come from the same (sub-block, 8-wide k slice) pairs);
 *  - the "min" term is a tiny int8 contraction of its own: the 6-bit mins are the B operand (k-slots 0..7 of lane group 0,
 *    zero elsewhere) and the sub-block sums arrive as two int8 images l, h with bsum = 64 h + l, so M = 64 C_h + C_l.
 * Integer-exact: P and M equal the scalar sums of the oracle bit for bit; the two fp32 FMAs per block are the oracle's.
 */
/* B operands of one Q4_K tile, ready for the matrix cores; the packed fragment is dead once this exists, so the
 * registers of the fragment can take the next tile's load while the MFMAs of this one run */
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
struct OpsQ4 { v4i bl[4], bh[4]; long bm; v8h bm16; float dw, dmin; };

__device__ __forceinline__ void unpack_q4(const FragQ4& f, int lane, OpsQ4& o) {
    const uint32_t qs[8] = {f.q0.x, f.q0.y, f.q0.z, f.q0.w, f.q1.x, f.q1.y, f.q1.z, f.q1.w};
    uint32_t sc_lo, sc_hi, mn_lo, mn_hi;
    q4k_scales(f.h, &sc_lo, &sc_hi, &mn_lo, &mn_hi);
    /* 3-bit digits of the eight scales, one per 16-bit half: D[2 * (j >> 2) + (j & 1)] holds scale j in half (j >> 1) & 1 */
    const uint32_t dl_lo = sc_lo & 0x07070707u, dl_hi = sc_hi & 0x07070707u;
    const uint32_t dh_lo = (sc_lo >> 3) & 0x07070707u, dh_hi = (sc_hi >> 3) & 0x07070707u;
    const uint32_t DL[4] = {dl_lo & 0x00FF00FFu, (dl_lo >> 8) & 0x00FF00FFu, dl_hi & 0x00FF00FFu, (dl_hi >> 8) & 0x00FF00FFu};
    const uint32_t DH[4] = {dh_lo & 0x00FF00FFu, (dh_lo >> 8) & 0x00FF00FFu, dh_hi & 0x00FF00FFu, (dh_hi >> 8) & 0x00FF00FFu};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = qs[j];
        const uint32_t lo = x & 0x0F0F0F0Fu;
        const uint32_t hi = (x >> 4) & 0x0F0F0F0Fu;
        const int di = 2 * (j >> 2) + (j & 1), sh = 16 * ((j >> 1) & 1), e = j & 1;
        const unsigned short sl = (unsigned short)(DL[di] >> sh), shh = (unsigned short)(DH[di] >> sh);
        o.bl[j >> 1][2 * e] = (int)pk_scale(lo, sl);
        o.bl[j >> 1][2 * e + 1] = (int)pk_scale(hi, sl);
        o.bh[j >> 1][2 * e] = (int)pk_scale(lo, shh);
        o.bh[j >> 1][2 * e + 1] = (int)pk_scale(hi, shh);
    }
    o.bm = (lane >> 4) == 0 ? (long)(((unsigned long)mn_hi << 32) | mn_lo) : 0L;
    /* f16 twin for the one-MFMA min term: k-slots 0..7 (lane group 0) pair 2 m_j with hh_j, k-slots 8..15 (group 1) pair m_j with ll_j */
    const float wsc = (lane >> 4) == 0 ? 2.0f : ((lane >> 4) == 1 ? 1.0f : 0.0f);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.bm16[e] = (_Float16)((float)(((e < 4 ? mn_lo : mn_hi) >> (8 * (e & 3))) & 0xFFu) * wsc);
    o.dw = f16bits_to_f32(f.h.x & 0xffffu);
    o.dmin = f16bits_to_f32(f.h.x >> 16);
}

template <int MT>
__device__ __forceinline__ void mma_q4(const OpsQ4& o, const uint8_t* lds_act, const uint8_t* lds_amn, const float* lds_ad, size_t act_ts, int amn_ts,
                                       int ad_ts, int blk, int lane, float (*acc)[4]) {
    const int g = lane >> 4;
    const v4i zero = {0, 0, 0, 0};
    v4i Pl[MT], Ph[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) Pl[m] = Ph[m] = zero;
    const uint8_t* ap = lds_act + (size_t)blk * 4096 + lane * 16;
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const v4i a = *(const v4i*)(ap + m * act_ts + j2 * 1024);
            Pl[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, o.bl[j2], Pl[m], 0, 0, 0);
            Ph[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, o.bh[j2], Ph[m], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const uint8_t* mp = lds_amn + m * amn_ts + (size_t)blk * 256 + (lane & 15) * 8;
        const v4i cl = __builtin_amdgcn_mfma_i32_16x16x32_i8(*(const long*)mp, o.bm, zero, 0, 0, 0);
        const v4i ch = __builtin_amdgcn_mfma_i32_16x16x32_i8(*(const long*)(mp + 128), o.bm, zero, 0, 0, 0);
        const v4f da = *(const v4f*)(lds_ad + m * ad_ts + blk * TK_ROW_SLOTS + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int P = (Ph[m][r] << 3) + Pl[m][r];
            const int M = (ch[r] << 6) + cl[r];
            acc[m][r] = tk_fmaf(o.dw * da[r], (float)P, acc[m][r]);
            acc[m][r] = tk_fmaf(-(o.dmin * da[r]), (float)M, acc[m][r]);
        }
    }
}

/* Q6_K: int8 operands 4 (q - 32) and the sixteen int8 group scales (one per 16 k: two per sub-block, applied to the two
 * half-masked MFMA results of the sub-block) */
struct OpsQ6 { long b[8]; uint32_t sc[4]; float dw; };

__device__ __forceinline__ void unpack_q6(const FragQ6& f, OpsQ6& o) {
    const uint32_t qs[8] = {f.q0.x, f.q0.y, f.q0.z, f.q0.w, f.q1.x, f.q1.y, f.q1.z, f.q1.w};
    const uint32_t qh[4] = {f.qh.x, f.qh.y, f.qh.z, f.qh.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = qs[j];
        const uint32_t H = qh[j >> 1];
        const int e = j & 1;
        const uint32_t lo = ((x << 2) & 0x3C3C3C3Cu) | ((H << (6 - 4 * e)) & 0xC0C0C0C0u);
        const uint32_t hi = ((x >> 2) & 0x3C3C3C3Cu) | ((H << (4 - 4 * e)) & 0xC0C0C0C0u);
        o.b[j] = (long)(((unsigned long)hi << 32) | lo);
    }
    o.sc[0] = f.sc.x; o.sc[1] = f.sc.y; o.sc[2] = f.sc.z; o.sc[3] = f.sc.w;
    o.dw = f16bits_to_f32(f.d);
}

template <int MT>
__device__ __forceinline__ void mma_q6(const OpsQ6& o, const uint8_t* lds_act, const float* lds_ad, size_t act_ts, int ad_ts, int blk, int lane,
                                       float (*acc)[4]) {
    const int g = lane >> 4;
    const v4i zero = {0, 0, 0, 0};
    const long mask_a = (g < 2) ? -1L : 0L;
    int P[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[m][r] = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = j & 1;
        const uint32_t w = o.sc[j >> 1];
        const int sca = (int)(int8_t)(w >> (16 * e));
        const int scb = (int)(int8_t)(w >> (16 * e + 8));
        const long ba = o.b[j] & mask_a, bb = o.b[j] & ~mask_a;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const long a = *(const long*)(lds_act + m * act_ts + (size_t)blk * 4096 + (j >> 1) * 1024 + lane * 16 + (j & 1) * 8);
            const v4i ca = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, ba, zero, 0, 0, 0);
            const v4i cb = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, bb, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[m][r] = mad24(scb, cb[r], mad24(sca, ca[r], P[m][r]));
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const v4f da = *(const v4f*)(lds_ad + m * ad_ts + blk * TK_ROW_SLOTS + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1f = (o.dw * da[r]) * 0.25f; /* operand bytes are 4*(q-32): exact power-of-two fold */
            acc[m][r] = tk_fmaf(s1f, (float)P[m][r], acc[m][r]);
        }
    }
}

size_t tk_gemv_lds_bytes(int K, int ks, int mtiles) {
    size_t Kr = (size_t)K / ks;
    return (size_t)mtiles * (Kr * TK_ROW_SLOTS + (Kr / 256) * TK_ROW_SLOTS * 4 + (Kr / 256) * 256);
}

/*
 * Work mapping (XCD/CU-aware, persistent-style): the launch has one workgroup per CU (256, rounded to a
 * multiple of ks).  Workgroup b owns K-range b % ks and the row tiles {b / ks + w * groups}, one per wave,
 * so (a) every CU streams the same number of 16-row tiles (+-1), (b) the K-range's int8 activations are
 * staged once per CU, (c) all waves of a CU walk disjoint contiguous tile runs.
 */
/* TYPES: bit 0 = the launch contains Q4_K tiles, bit 1 = Q6_K tiles; single-type launches keep only one fragment ring in registers */
/* FUSE (TkGemvArgs::fuse, MT = 1 only): 0 = the activation image comes from global memory; 1 / 2 = every workgroup forms it itself —
 * the norm's or SwiGLU's arithmetic, value for value what k_rmsnorm_q8 / k_swiglu_q8 write — under the latency of its first weight tiles */
template <int PF, int MT, int TYPES, int FUSE>
__global__ __launch_bounds__(512) void k_gemv_w4a8(TkGemvArgs a, int groups, int total_row_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* scalar: tile/segment/type selection stays on the SALU */
    const int Kr = a.K / a.ks;
    const int nb = Kr / 256;           /* host guarantees nb % PF == 0 */
    const int ngrp = nb / PF;
    const int nblk_total = a.K / 256;
    const int ksi = blockIdx.x % a.ks;
    const int blk0 = ksi * nb;

    /* which segment / tile does this wave own */
    int rt = blockIdx.x / a.ks + wave * groups;
    const bool active = rt < total_row_tiles;
    if (!active) rt = 0;
    int seg = 0, row_base = 0;
    while (seg < a.nseg - 1 && rt >= a.seg[seg].row_tiles) {
        rt -= a.seg[seg].row_tiles;
        row_base += a.seg[seg].row_tiles * TK_TILE_ROWS;
        ++seg;
    }
    const int type = a.seg[seg].type;
    /* compile-time tile pitch in single-type launches: tile addresses become scalar base + immediate */
    const size_t tile_bytes = TYPES == 1 ? (size_t)TK_Q4K_TILE_BYTES : TYPES == 2 ? (size_t)TK_Q6K_TILE_BYTES
                                         : (type == TK_TYPE_Q4_K ? (size_t)TK_Q4K_TILE_BYTES : (size_t)TK_Q6K_TILE_BYTES);
    const uint8_t* tile = a.seg[seg].tiles + ((size_t)rt * nblk_total + blk0) * tile_bytes;

    /* LDS: [MT] activation images, then [MT] block scales, then [MT] sub-block sums */
    const size_t act_ts = (size_t)Kr * TK_ROW_SLOTS;
    const int ad_ts = nb * TK_ROW_SLOTS, abs_ts = nb * 256; /* per-tile strides: floats, bytes */
    uint8_t* lds_act = lds;
    float* lds_ad = (float*)(lds + MT * act_ts);
    uint8_t* lds_abs = (uint8_t*)(lds_ad + MT * ad_ts); /* sub-block sums as (l, h) int8 images, 256 B per block */

    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = 0.0f;

    /* The first PF weight tiles are requested before the activations are staged so HBM latency overlaps the LDS
     * fill.  Loop bodies below contain NO conditional loads: hipcc then keeps counted vmcnt waits and the next
     * group's tiles stay in flight under the current group's MFMAs (a branch around a load costs a vmcnt(0)). */
    constexpr bool HAS4 = (TYPES & 1) != 0, HAS6 = (TYPES & 2) != 0;
    const bool is4 = HAS4 && (!HAS6 || type == TK_TYPE_Q4_K);
    FragQ4 f4[HAS4 ? PF : 1];
    FragQ6 f6[HAS6 ? PF : 1];
    if (HAS4 && is4) {
#pragma unroll
        for (int u = 0; u < PF; ++u) f4[HAS4 ? u : 0] = load_q4(tile + (size_t)u * tile_bytes, lane);
    }
    if (HAS6 && !is4) {
#pragma unroll
        for (int u = 0; u < PF; ++u) f6[HAS6 ? u : 0] = load_q6(tile + (size_t)u * tile_bytes, lane);
    }
    if (FUSE != 0) {
        const int nw = nthr >> 6, hw = tid >> 5, nhw = nthr >> 5;
        TkActQ8 lo{}; /* the LDS image as quantize_chunk8's destination: same layout as the global one, block 0 = this K-range's first */
        lo.aq = (int8_t*)lds_act; lo.ad = lds_ad; lo.abs = (int8_t*)lds_abs; lo.aq_ts = act_ts; lo.ad_ts = (size_t)ad_ts; lo.abs_ts = (size_t)abs_ts;
        if (FUSE == 1) {
            const int D = a.K, ngrp4 = D / 4;
            float* hbuf = (float*)(lds_abs + MT * abs_ts); /* the finished row, then 4 floats of the canonical sum */
            float* red = hbuf + D;
            /* norm weights of this half-wave's first block: requested before the reductions so their latency hides under them */
            v4f wa0 = {0.0f, 0.0f, 0.0f, 0.0f}, wa1 = wa0;
            if (hw < nb) { const int c = 32 * (blk0 + hw) + (lane & 31); wa0 = *(const v4f*)(a.fw + 8 * c); wa1 = *(const v4f*)(a.fw + 8 * c + 4); }
            for (int r = 0; r < a.nrows; ++r) {
                if (r > 0) __syncthreads(); /* hbuf and red are reused */
                for (int gi = tid; gi < ngrp4; gi += nthr) { /* the row and ALL its slabs requested before the first add (k_rmsnorm_q8) */
                    v4f v = *(const v4f*)(a.fx_in + (int64_t)r * D + 4 * gi);
                    if (a.fslab) {
                        v4f p[TK_RMS_MAX_KS];
#pragma unroll
                        for (int sl = 0; sl < TK_RMS_MAX_KS; ++sl)
                            if (sl < a.fks) p[sl] = *(const v4f*)(a.fslab + ((int64_t)sl * TK_MAX_ROWS + r) * a.fn_total + 4 * gi);
                        v4f o = p[0];
#pragma unroll
                        for (int sl = 1; sl < TK_RMS_MAX_KS; ++sl)
                            if (sl < a.fks) o = o + p[sl];
                        v = v + o;
                    }
                    if (blockIdx.x == 0) *(v4f*)(a.fx_out + (int64_t)r * D + 4 * gi) = v; /* the residual stream moves on once */
                    *(v4f*)(hbuf + 4 * gi) = v;
                }
                __syncthreads();
                /* sum of squares as the canonical 256 chains; chain t = 64 vw + lane belongs to "wave" vw of k_rmsnorm_q8's first four */
                for (int vw = wave; vw < 4; vw += nw) {
                    float ss = 0.0f;
                    for (int gi = 64 * vw + lane; gi < ngrp4; gi += 256) {
                        const v4f v = *(const v4f*)(hbuf + 4 * gi);
#pragma unroll
                        for (int i = 0; i < 4; ++i) ss = tk_fmaf(v[i], v[i], ss);
                    }
                    for (int sh = 32; sh >= 1; sh >>= 1) ss = ss + wave_xor_f(ss, sh);
                    if (lane == 0) red[vw] = ss;
                }
                __syncthreads();
                const float tot = ((red[0] + red[1]) + red[2]) + red[3];
                const float scale = tk_divf(1.0f, tk_sqrtf(tk_divf(tot, (float)D) + a.feps));
                for (int b = hw; b < nb; b += nhw) { /* a half-wave quantises one 256-block of this K-range */
                    const int cl = 32 * b + (lane & 31), c = 32 * blk0 + cl;
                    v4f w0, w1;
                    if (b == hw) { w0 = wa0; w1 = wa1; }
                    else { w0 = *(const v4f*)(a.fw + 8 * c); w1 = *(const v4f*)(a.fw + 8 * c + 4); }
                    const v4f h0 = *(const v4f*)(hbuf + 8 * c), h1 = *(const v4f*)(hbuf + 8 * c + 4);
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = (h0[i] * scale) * w0[i];
                        v[4 + i] = (h1[i] * scale) * w1[i];
                    }
                    quantize_chunk8(v, cl, r, lo);
                }
            }
        } else {
            const int FF = a.K;
            for (int r = 0; r < a.nrows; ++r)
                for (int b = hw; b < nb; b += nhw) {
                    const int cl = 32 * b + (lane & 31), c = 32 * blk0 + cl;
                    v4f g[2], u[2];
                    {
                        v4f pg[TK_RMS_MAX_KS][2], pu[TK_RMS_MAX_KS][2];
#pragma unroll
                        for (int sl = 0; sl < TK_RMS_MAX_KS; ++sl)
                            if (sl < a.fks) {
                                const float* row = a.fslab + ((int64_t)sl * TK_MAX_ROWS + r) * (2 * (int64_t)FF);
                                pg[sl][0] = *(const v4f*)(row + 8 * c); pg[sl][1] = *(const v4f*)(row + 8 * c + 4);
                                pu[sl][0] = *(const v4f*)(row + FF + 8 * c); pu[sl][1] = *(const v4f*)(row + FF + 8 * c + 4);
                            }
                        g[0] = pg[0][0]; g[1] = pg[0][1]; u[0] = pu[0][0]; u[1] = pu[0][1];
#pragma unroll
                        for (int sl = 1; sl < TK_RMS_MAX_KS; ++sl)
                            if (sl < a.fks) { g[0] = g[0] + pg[sl][0]; g[1] = g[1] + pg[sl][1]; u[0] = u[0] + pu[sl][0]; u[1] = u[1] + pu[sl][1]; }
                    }
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = tk_siluf(g[i >> 2][i & 3]) * u[i >> 2][i & 3];
                    quantize_chunk8(v, cl, r, lo);
                }
        }
        __builtin_amdgcn_s_waitcnt(0); /* the image's stores and the first weight tiles */
    } else {
        /* activations: LDS-DMA (global_load_lds_dwordx4), one contiguous 1 KiB piece per wave-instruction, no VGPR
         * round trip and no per-piece wait: the whole K-range image is in flight at once. */
        const int nw = nthr >> 6, npiece = Kr * TK_ROW_SLOTS / 1024;
        const int nd = nb * 4, ns = nb * 16; /* uint4 counts of the scale and (l, h) images */
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint8_t* src = (const uint8_t*)a.aq + m * a.aq_ts + (size_t)blk0 * 256 * TK_ROW_SLOTS;
            /* the CUs of one XCD (blockIdx / 8 = index inside the XCD) all stage the same image from the same L2: each starts at
             * its own piece so they do not hammer one L2 channel in lock step */
            /* pieces go in groups of four (npiece = Kr / 64 is a multiple of 4): one address and one LDS base per 4 KiB, immediate offsets */
            const int ngroup = npiece >> 2;
            const int rot = (int)((blockIdx.x >> 3) & 31) * ngroup >> 5;
            for (int c0 = wave; c0 < ngroup; c0 += nw) {
                const int c = c0 + rot < ngroup ? c0 + rot : c0 + rot - ngroup;
                const auto gs = (const __attribute__((address_space(1))) void*)(src + (size_t)c * 4096 + lane * 16);
                const auto ls = (__attribute__((address_space(3))) void*)(lds_act + m * act_ts + c * 4096);
                __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gs, ls, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(gs, ls, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds(gs, ls, 16, 3072, 0);
            }
        }
        /* block scales and sub-block sums: <= 5 x 16 B per thread and tile, loaded together, stored together */
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint4* sd = (const uint4*)(a.ad + m * a.ad_ts + (size_t)blk0 * TK_ROW_SLOTS);
            const uint4* sb = (const uint4*)(a.abs + m * a.abs_ts + (size_t)blk0 * 256);
            uint4 t0 = make_uint4(0, 0, 0, 0), t1[4];
            if (tid < nd) t0 = sd[tid];
#pragma unroll
            for (int k = 0; k < 4; ++k) t1[k] = (tid + k * nthr < ns) ? sb[tid + k * nthr] : make_uint4(0, 0, 0, 0);
            if (tid < nd) ((uint4*)(lds_ad + m * ad_ts))[tid] = t0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid + k * nthr < ns) ((uint4*)(lds_abs + m * abs_ts))[tid + k * nthr] = t1[k];
        }
        __builtin_amdgcn_s_waitcnt(0); /* vmcnt(0): the LDS-DMA pieces have landed (the first weight tiles too) */
    }
    __syncthreads();
    if (!active) return;

    /* steady state per tile: unpack (frees the packed fragment) -> request the tile PF blocks ahead into the same registers ->
     * MFMAs; so a tile has PF - 1 blocks of MFMA time plus its own to arrive */
    if (HAS4 && is4) {
        const uint8_t* tp = tile + PF * tile_bytes; /* one moving wave-uniform pointer: no per-load 64-bit VGPR address chains */
#pragma unroll 1
        for (int g = 0; g < ngrp - 1; ++g, tp += PF * tile_bytes) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                OpsQ4 o;
                __builtin_amdgcn_sched_barrier(0);
                unpack_q4(f4[HAS4 ? u : 0], lane, o);
                __builtin_amdgcn_sched_barrier(0);
                f4[HAS4 ? u : 0] = load_q4(tp + u * tile_bytes, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_q4<MT>(o, lds_act, lds_abs, lds_ad, act_ts, abs_ts, ad_ts, g * PF + u, lane, acc);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            OpsQ4 o;
            __builtin_amdgcn_sched_barrier(0);
            unpack_q4(f4[HAS4 ? u : 0], lane, o);
            mma_q4<MT>(o, lds_act, lds_abs, lds_ad, act_ts, abs_ts, ad_ts, (ngrp - 1) * PF + u, lane, acc);
        }
    }
    if (HAS6 && !is4) {
        const uint8_t* tp = tile + PF * tile_bytes;
#pragma unroll 1
        for (int g = 0; g < ngrp - 1; ++g, tp += PF * tile_bytes) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                OpsQ6 o;
                __builtin_amdgcn_sched_barrier(0);
                unpack_q6(f6[HAS6 ? u : 0], o);
                __builtin_amdgcn_sched_barrier(0);
                f6[HAS6 ? u : 0] = load_q6(tp + u * tile_bytes, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_q6<MT>(o, lds_act, lds_ad, act_ts, ad_ts, g * PF + u, lane, acc);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            OpsQ6 o;
            __builtin_amdgcn_sched_barrier(0);
            unpack_q6(f6[HAS6 ? u : 0], o);
            mma_q6<MT>(o, lds_act, lds_ad, act_ts, ad_ts, (ngrp - 1) * PF + u, lane, acc);
        }
    }

    const int n = a.col0 + row_base + rt * TK_TILE_ROWS + (lane & 15);
    const int g = lane >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * TK_ROW_SLOTS + 4 * g + r;
            /* default-policy stores: the <= 1 MB of slabs of a narrow pass stay in L2 / the Infinity Cache for the consumer kernel, which is one
             * load latency long (decode step 2.277 -> 2.207 ms at 16 rows, 2.108 -> 2.038 ms at 1 row against non-temporal stores,
             * profiles/r03_smallbatch_variants.txt); the wide kernels keep non-temporal stores (29 MB per launch would only evict) */
            if (row < a.nrows) a.out[((size_t)ksi * TK_MAX_ROWS + row) * a.n_total + n] = acc[m][r];
        }
}

/* ------------------------------------------------------------------------------------------
 * Batched passes (33..256 rows: prefill chunks, wide decode batches): same work mapping and the same per-row arithmetic as
 * k_gemv_w4a8, but the weight tile is unpacked ONCE and multiplied against MT = 4, 6, 8, 10 or 12 M-tiles (the pass's rows / 16 rounded up to the next
 * instantiation; 14 and 16 exist for the TK_MI355X_G32_FROM=257 A/B against the 32x32x32 kernel), so the dequantisation VALU
 * work and the HBM bytes per row drop by MT.  The K-range's activations no longer fit in LDS, so they stream through a
 * two-slot ring of 256-k blocks (MT x {4 KiB int8 image, 512 B f16 sub-block sums, 64 B scales} each):
 *
 *   every TK_RING_BLOCKS blocks:  s_waitcnt vmcnt(0) -> my DMA pieces of this slot (and my weight tile) have landed
 *                                 barrier            -> everybody's pieces have; everybody is done with the other slot
 *                                 DMA the next TK_RING_BLOCKS blocks into the other slot
 *   every block:                  unpack tile b, request tile b+1 into the same registers, MT x MFMA chains with the LDS operand reads
 *                                 of M-tile m+1 issued before the MFMAs of M-tile m and the fp32 FMAs of m-1 filling the MFMA shadow
 *
 * so a slot's activations have a whole slot time (> 2 us) and the next weight tile a whole block time to arrive.
 * Q6_K tiles are folded like Q4_K ones here: v = scale * (q - 32) (14 bits signed) is split v = 64 vh + vl (vl 0..63,
 * vh -64..64, both int8) with packed 16-bit arithmetic, once per tile, so each M-tile costs 8 chained MFMAs and no VALU
 * scale work.  Integer-exact like everything else: P is the oracle's integer.
 * ------------------------------------------------------------------------------------------ */
typedef short v2s16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_ashr16(uint32_t x, int n) {
    const v2s16 r = __builtin_bit_cast(v2s16, x) >> (v2s16){(short)n, (short)n};
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_shl16(uint32_t x, int n) {
    const v2u16 r = __builtin_bit_cast(v2u16, x) << (v2u16){(unsigned short)n, (unsigned short)n};
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_mul16(uint32_t x, short s) {
    const v2s16 r = __builtin_bit_cast(v2s16, x) * (v2s16){s, s};
    return __builtin_bit_cast(uint32_t, r);
}

/* four int8 operand bytes b_t = 4 q_t of one dword -> the two digit dwords of v_t = s * q_t */
__device__ __forceinline__ void q6_digits(uint32_t b, short s, int* vl, int* vh) {
    const uint32_t qe = pk_ashr16(pk_shl16(b, 8), 10); /* (q_0, q_2) sign-extended to 16 bits */
    const uint32_t qo = pk_ashr16(b, 10);              /* (q_1, q_3) */
    const uint32_t we = pk_mul16(qe, s), wo = pk_mul16(qo, s);
    const uint32_t le = we & 0x003F003Fu, lo = wo & 0x003F003Fu;
    const uint32_t he = pk_ashr16(we, 6), ho = pk_ashr16(wo, 6);
    /* bytes (e.0, o.0, e.2, o.2): v_perm_b32 selector, source bytes 0..3 = second operand, 4..7 = first */
    *vl = (int)__builtin_amdgcn_perm(lo, le, 0x06020400u);
    *vh = (int)__builtin_amdgcn_perm(ho, he, 0x06020400u);
}

__device__ __forceinline__ void unpack_q6_fold(const FragQ6& f, int lane, OpsQ4& o) { /* bm / dmin stay unused: Q6_K has no min term */
    const uint32_t qs[8] = {f.q0.x, f.q0.y, f.q0.z, f.q0.w, f.q1.x, f.q1.y, f.q1.z, f.q1.w};
    const uint32_t qh[4] = {f.qh.x, f.qh.y, f.qh.z, f.qh.w};
    const uint32_t scw[4] = {f.sc.x, f.sc.y, f.sc.z, f.sc.w};
    const int upper = (lane >> 5) & 1; /* lane groups 2, 3 hold k 16..31 of every sub-block: the second group scale */
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = qs[j];
        const uint32_t H = qh[j >> 1];
        const int e = j & 1;
        const uint32_t lo = ((x << 2) & 0x3C3C3C3Cu) | ((H << (6 - 4 * e)) & 0xC0C0C0C0u);
        const uint32_t hi = ((x >> 2) & 0x3C3C3C3Cu) | ((H << (4 - 4 * e)) & 0xC0C0C0C0u);
        const short s = (short)(int8_t)(scw[j >> 1] >> (16 * e + 8 * upper));
        int l0, h0, l1, h1;
        q6_digits(lo, s, &l0, &h0);
        q6_digits(hi, s, &l1, &h1);
        o.bl[j >> 1][2 * e] = l0; o.bl[j >> 1][2 * e + 1] = l1;
        o.bh[j >> 1][2 * e] = h0; o.bh[j >> 1][2 * e + 1] = h1;
    }
    o.dw = f16bits_to_f32(f.d);
}

/* A-side operands of one M-tile of one ring block, read from LDS one tile AHEAD of the MFMAs that consume them */
struct ATile { v4i a[4]; v8h mn; v4f da; };
template <bool MINS>
__device__ __forceinline__ void lds_tile(ATile& t, const uint8_t* act, const uint8_t* amn, const uint8_t* ad, int lane) {
    const uint8_t* ap = act + lane * 16;
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        t.a[j2] = *(const v4i*)(ap + j2 * 1024);
    }
    if (MINS) t.mn = *(const v8h*)(amn + ((lane >> 4) & 1) * 256 + (lane & 15) * 16); /* groups 2, 3 re-read finite numbers: their B operand is 0 */
    t.da = *(const v4f*)(ad + 16 * (lane >> 4));
}
/* integer results of one M-tile, finished (two fp32 FMAs per row) while the next tile's MFMAs run */
struct PTile { v4i pl, ph; v4f cm, da; };

#define TK_RING_TILE_BYTES (256 * TK_ROW_SLOTS + 512 + TK_ROW_SLOTS * 4) /* one M-tile of one 256-k block: image + f16 sums + scales */
#ifndef TK_GEMM_LDS_DEPTH
#define TK_GEMM_LDS_DEPTH 2
#endif
#ifndef TK_RING_BLOCKS
#define TK_RING_BLOCKS 1 /* 256-k blocks per ring slot (per barrier): 2 slots x 1 block x 16 M-tiles = 138 KiB of the 160 KiB LDS */
#endif

#define TK_MFMA64 __builtin_amdgcn_mfma_i32_16x16x64_i8
template <bool Q4>
__device__ __forceinline__ void finish_tile(const PTile& R, const OpsQ4& o, float* acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] = tk_fmaf(o.dw * R.da[r], (float)((R.ph[r] << (Q4 ? 3 : 6)) + R.pl[r]), acc[r]);
        if (Q4) acc[r] = tk_fmaf(-(o.dmin * R.da[r]), R.cm[r], acc[r]); /* cm = sum_j m_j bsum_j, an exact integer below 2^24 */
    }
}

/* one 256-k block: NT weight tiles (operands in registers) x MT M-tiles (operands streamed from the LDS ring, each read once
 * for all NT weight tiles) */
/* rot (0 or MT / 2, wave-uniform): register slot m of this wave works on M-tile m ^ rot.  The second half of a workgroup's waves (the
 * SIMD partners of the first half) walk the M-tiles from the middle, so the two waves of a SIMD never want the same LDS rows and the
 * matrix pipe at the same moment (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  M-tiles are independent accumulators: the order
 * does not enter the result; the epilogue stores slot m to rows of tile m ^ rot. */
template <int MT, int NT, bool Q4>
__device__ __forceinline__ void gemm_block(const OpsQ4 (&o)[NT], const uint8_t* chunk, int rot, int lane, float (&acc)[NT][MT][4]) {
    constexpr int OFF_AMN = MT * 4096, OFF_AD = MT * 4096 + MT * 512;
    const v4i zero = {0, 0, 0, 0};
    constexpr int AD = TK_GEMM_LDS_DEPTH; /* M-tiles of LDS operand reads in flight ahead of the MFMAs */
    ATile T[AD + 1];
    PTile R[NT];
    /* (m ^ rot) * S = m * S + rot * S for the lower half of the slots, m * S - rot * S for the upper half: two bases, static offsets */
    const uint8_t* act[2] = {chunk + rot * 4096, chunk - rot * 4096};
    const uint8_t* amn[2] = {chunk + OFF_AMN + rot * 512, chunk + OFF_AMN - rot * 512};
    const uint8_t* adp[2] = {chunk + OFF_AD + rot * 64, chunk + OFF_AD - rot * 64};
#define TK_LDS_TILE(slot, m) lds_tile<Q4>(T[slot], act[(m) >= MT / 2] + (m) * 4096, amn[(m) >= MT / 2] + (m) * 512, adp[(m) >= MT / 2] + (m) * 64, lane)
#pragma unroll
    for (int m = 0; m < AD && m < MT; ++m) TK_LDS_TILE(m, m);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m + AD < MT) TK_LDS_TILE((m + AD) % (AD + 1), m + AD);
        __builtin_amdgcn_sched_barrier(0);
        const ATile& t = T[m % (AD + 1)];
        PTile c[NT];
#pragma unroll
        for (int w = 0; w < NT; ++w) {
            c[w].pl = TK_MFMA64(t.a[0], o[w].bl[0], zero, 0, 0, 0);
            c[w].ph = TK_MFMA64(t.a[0], o[w].bh[0], zero, 0, 0, 0);
        }
#pragma unroll
        for (int j2 = 1; j2 < 4; ++j2)
#pragma unroll
            for (int w = 0; w < NT; ++w) {
                c[w].pl = TK_MFMA64(t.a[j2], o[w].bl[j2], c[w].pl, 0, 0, 0);
                c[w].ph = TK_MFMA64(t.a[j2], o[w].bh[j2], c[w].ph, 0, 0, 0);
            }
#pragma unroll
        for (int w = 0; w < NT; ++w) {
            if (Q4) c[w].cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.mn, o[w].bm16, (v4f){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            c[w].da = t.da;
        }
        if (m > 0) {
#pragma unroll
            for (int w = 0; w < NT; ++w) finish_tile<Q4>(R[w], o[w], acc[w][m - 1]);
        }
#pragma unroll
        for (int w = 0; w < NT; ++w) R[w] = c[w];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int w = 0; w < NT; ++w) finish_tile<Q4>(R[w], o[w], acc[w][MT - 1]);
#undef TK_LDS_TILE
}

/* NT = weight tiles per wave: 2 (adjacent row tiles, same tensor) when a CU owns enough tiles to keep four such waves busy —
 * the LDS operand stream, the bound of this kernel, is then read once per TWO weight tiles */
template <int MT, int TYPES, int NT>
__global__ __launch_bounds__(NT == 2 ? 256 : 512) void k_gemm_w4a8(TkGemvArgs a, int groups, int total_row_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int CH = MT * TK_RING_TILE_BYTES;   /* one block of the ring */
    constexpr int CB = TK_RING_BLOCKS;            /* blocks per ring slot = per barrier */
    constexpr int OFF_AMN = MT * 4096, OFF_AD = MT * 4096 + MT * 512;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int nb = a.K / a.ks / 256;
    const int nblk_total = a.K / 256;
    const int ksi = blockIdx.x % a.ks;
    const int blk0 = ksi * nb;
    const int rot = wave >= 4 ? MT / 2 : 0; /* waves 4..7 share their SIMDs with waves 0..3: they start from the middle M-tile (gemm_block) */

    int rt = NT * (blockIdx.x / a.ks + wave * groups); /* first of this wave's NT adjacent row tiles */
    const bool active = rt < total_row_tiles;
    if (!active) rt = 0;
    int seg = 0, row_base = 0;
    while (seg < a.nseg - 1 && rt >= a.seg[seg].row_tiles) {
        rt -= a.seg[seg].row_tiles;
        row_base += a.seg[seg].row_tiles * TK_TILE_ROWS;
        ++seg;
    }
    const int type = a.seg[seg].type;
    constexpr bool HAS4 = (TYPES & 1) != 0, HAS6 = (TYPES & 2) != 0;
    const bool is4 = HAS4 && (!HAS6 || type == TK_TYPE_Q4_K);
    const size_t tile_bytes = is4 ? (size_t)TK_Q4K_TILE_BYTES : (size_t)TK_Q6K_TILE_BYTES;
    const size_t tile_pitch = (size_t)nblk_total * tile_bytes; /* to the same block of the next row tile */
    const uint8_t* tile = a.seg[seg].tiles + ((size_t)rt * nblk_total + blk0) * tile_bytes;

    float acc[NT][MT][4];
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[w][m][r] = 0.0f;

    /* ring staging: whole M-tiles dealt round-robin to the waves; per tile one address computation and one LDS base feed four 1 KiB
     * LDS-DMA instructions (immediate offsets) for the int8 image, then 512 B of f16 sub-block sums and 64 B of scales.  (The device-side
     * timestamp trace in profiles/r01_gemm_batched.txt had the issue of the staging at 16 % of a step when every 1 KiB piece carried its
     * own address arithmetic and M0 write; dealing half-tiles for a better balance over 7 waves measured slower than this.) */
    auto stage = [&](int c, int slot) {
        uint8_t* dst = lds + slot * CH;
        for (int m = wave; m < MT; m += nw) {
            const uint8_t* src = (const uint8_t*)a.aq + m * a.aq_ts + (size_t)(blk0 + c) * 4096 + lane * 16;
            const auto gs = (const __attribute__((address_space(1))) void*)src;
            const auto ls = (__attribute__((address_space(3))) void*)(dst + m * 4096);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 3072, 0);
            const uint8_t* sm = (const uint8_t*)(a.abs16 + m * a.abs_ts + (size_t)(blk0 + c) * 256) + lane * 16;
            if (lane < 32)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sm,
                                                 (__attribute__((address_space(3))) void*)(dst + OFF_AMN + m * 512), 16, 0, 0);
            const uint8_t* sd = (const uint8_t*)(a.ad + m * a.ad_ts + (size_t)(blk0 + c) * TK_ROW_SLOTS) + lane * 16;
            if (lane < 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sd,
                                                 (__attribute__((address_space(3))) void*)(dst + OFF_AD + m * 64), 16, 0, 0);
        }
    };

    FragQ4 f4[HAS4 ? NT : 1];
    FragQ6 f6[HAS6 ? NT : 1];
    if (active) {
#pragma unroll
        for (int w = 0; w < NT; ++w) {
            if (HAS4 && is4) f4[HAS4 ? w : 0] = load_q4(tile + w * tile_pitch, lane);
            if (HAS6 && !is4) f6[HAS6 ? w : 0] = load_q6(tile + w * tile_pitch, lane);
        }
    }
    for (int i = 0; i < CB && i < nb; ++i) stage(i, i);

#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        if (b % CB == 0) { /* chunk boundary: CB blocks per barrier */
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            for (int i = 0; i < CB && b + CB + i < nb; ++i) stage(b + CB + i, ((b / CB + 1) & 1) * CB + i);
        }
        if (!active) continue;
        const uint8_t* chunk = lds + (((b / CB) & 1) * CB + b % CB) * CH;
        const uint8_t* next = tile + (size_t)(b + 1 < nb ? b + 1 : b) * tile_bytes; /* the last step re-requests its own tile: no branch around a load */
        OpsQ4 o[NT];
        if (HAS4 && is4) {
#pragma unroll
            for (int w = 0; w < NT; ++w) unpack_q4(f4[HAS4 ? w : 0], lane, o[w]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NT; ++w) f4[HAS4 ? w : 0] = load_q4(next + w * tile_pitch, lane);
            __builtin_amdgcn_sched_barrier(0);
            gemm_block<MT, NT, true>(o, chunk, rot, lane, acc);
        }
        if (HAS6 && !is4) {
#pragma unroll
            for (int w = 0; w < NT; ++w) unpack_q6_fold(f6[HAS6 ? w : 0], lane, o[w]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NT; ++w) f6[HAS6 ? w : 0] = load_q6(next + w * tile_pitch, lane);
            __builtin_amdgcn_sched_barrier(0);
            gemm_block<MT, NT, false>(o, chunk, rot, lane, acc);
        }
    }
    if (!active) return;

    const int g = lane >> 4;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
        const int n = a.col0 + row_base + (rt + w) * TK_TILE_ROWS + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mt = rot == 0 ? m : (m < MT / 2 ? m + rot : m - rot); /* register slot m holds M-tile m ^ rot (the half-swap: MT = 6 is no power of two) */
                const int row = mt * TK_ROW_SLOTS + 4 * g + r;
                if (row < a.nrows) __builtin_nontemporal_store(acc[w][m][r], &a.out[((size_t)ksi * TK_MAX_ROWS + row) * a.n_total + n]);
            }
    }
}


/* ------------------------------------------------------------------------------------------
 * Wide passes (193..256 rows by default, from 129 with TK_MI355X_G32_FROM=129): the same arithmetic on v_mfma_i32_32x32x32_i8.
 *
 * Why: per (16 rows x 16 weight rows x 256 k) the 16x16x64 formulation above issues 8 int8 MFMAs + 1 f16 MFMA (9 x 8 issue cycles) and four
 * ds_read_b128; the 32x32x32 form does the same multiply-adds with HALF the MFMA issue slots and HALF the LDS operand bytes per MAC (one
 * 16-byte A read feeds 32 weight rows instead of 16), which is what the 16x16 kernel is bound by (profiles/r02_gemm_sq_counters.txt:
 * issue port 67 %, matrix pipe 43 %).  Matrix-pipe cycles per MAC are unchanged (the Q4_K scale digit split still doubles them).
 *
 * Mapping: a wave owns TWO adjacent 16-row weight tiles (32 weight rows = the N of the MFMA) and four 32-row M-tiles (128 rows); the
 * workgroup's waves w and w + 4 — SIMD partners — own the same weight rows and the two row halves, so while one finishes a tile on the
 * VALU (two fp32 FMAs per output and block, the oracle's) the other runs its MFMAs.  Both read the SAME HBM layouts as the kernels above:
 *  - weight tiles: lane (n, g) of a 16-row tile holds the k-slice g of every sub-block; the B operand of the 32x32x32 MFMA wants lane
 *    (n32, h) to hold 16 k of weight row n32.  One v_permlane16_swap_b32 per dword pair (tile 0's dword, tile 1's dword) moves the odd
 *    16-lane rows of tile 0 against the even rows of tile 1: afterwards register 0 holds k-slice 2h and register 1 k-slice 2h + 1 of
 *    weight row n32 in lane (n32, h) — for rows 0..15 from tile 0, rows 16..31 from tile 1.  No second copy of the weights in HBM.
 *  - activation image [K/64][4 g][16 slots][2 sub-blocks][8]: lane (m32, h) reads the 16 bytes of (g = 2h + gp, slot m32 % 16) of M-tile
 *    m32 / 16 — the same (sub-block pair, k-slice) its B operand register gp carries; the k order inside an MFMA is free as long as A
 *    and B agree.  One base VGPR, immediate offsets for all 64 reads of a block.
 *  - the Q4_K min term: one v_mfma_f32_32x32x16_f16 per M-tile on the (hh, ll) f16 sums, as above.
 * Integer-exact: P and M are the oracle's integers; the float epilogue per block is unchanged.
 * ------------------------------------------------------------------------------------------ */
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));

struct Ops32 { v4i bl[8], bh[8]; v8h bm16; float dw, dmin; };

/* (a: this lane's dword of tile 0, b: of tile 1) -> (k-slice 2h, k-slice 2h + 1) of weight row n32 = lane & 31, h = lane >> 5 */
__device__ __forceinline__ void pair_swap(uint32_t a, uint32_t b, uint32_t* s0, uint32_t* s1) {
    const v2u32 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    *s0 = r.x;
    *s1 = r.y;
}

__device__ __forceinline__ void unpack_q4_x32(const FragQ4& f0, const FragQ4& f1, int lane, Ops32& o) {
    const bool up = (lane & 16) != 0; /* weight rows 16..31 of the pair: tile 1's header */
    const uint4 hdr = up ? f1.h : f0.h;
    const uint32_t q0[8] = {f0.q0.x, f0.q0.y, f0.q0.z, f0.q0.w, f0.q1.x, f0.q1.y, f0.q1.z, f0.q1.w};
    const uint32_t q1[8] = {f1.q0.x, f1.q0.y, f1.q0.z, f1.q0.w, f1.q1.x, f1.q1.y, f1.q1.z, f1.q1.w};
    uint32_t sc_lo, sc_hi, mn_lo, mn_hi;
    q4k_scales(hdr, &sc_lo, &sc_hi, &mn_lo, &mn_hi);
    const uint32_t dl_lo = sc_lo & 0x07070707u, dl_hi = sc_hi & 0x07070707u;
    const uint32_t dh_lo = (sc_lo >> 3) & 0x07070707u, dh_hi = (sc_hi >> 3) & 0x07070707u;
    const uint32_t DL[4] = {dl_lo & 0x00FF00FFu, (dl_lo >> 8) & 0x00FF00FFu, dl_hi & 0x00FF00FFu, (dl_hi >> 8) & 0x00FF00FFu};
    const uint32_t DH[4] = {dh_lo & 0x00FF00FFu, (dh_lo >> 8) & 0x00FF00FFu, dh_hi & 0x00FF00FFu, (dh_hi >> 8) & 0x00FF00FFu};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t s[2];
        pair_swap(q0[j], q1[j], &s[0], &s[1]);
        const int di = 2 * (j >> 2) + (j & 1), sh = 16 * ((j >> 1) & 1), e = j & 1;
        const unsigned short sl = (unsigned short)(DL[di] >> sh), shh = (unsigned short)(DH[di] >> sh);
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) { /* MFMA u = 2 (j / 2) + gp contracts k-slices 2h + gp of sub-blocks 2 (j / 2) and 2 (j / 2) + 1 */
            const uint32_t lo = s[gp] & 0x0F0F0F0Fu, hi = (s[gp] >> 4) & 0x0F0F0F0Fu;
            const int u = 2 * (j >> 1) + gp;
            o.bl[u][2 * e] = (int)pk_scale(lo, sl);
            o.bl[u][2 * e + 1] = (int)pk_scale(hi, sl);
            o.bh[u][2 * e] = (int)pk_scale(lo, shh);
            o.bh[u][2 * e + 1] = (int)pk_scale(hi, shh);
        }
    }
    /* min term: k-slots 0..7 (lane half 0) pair 2 m_j with hh_j, k-slots 8..15 (half 1) pair m_j with ll_j */
    const float wsc = (lane >> 5) == 0 ? 2.0f : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.bm16[e] = (_Float16)((float)(((e < 4 ? mn_lo : mn_hi) >> (8 * (e & 3))) & 0xFFu) * wsc);
    o.dw = f16bits_to_f32(hdr.x & 0xffffu);
    o.dmin = f16bits_to_f32(hdr.x >> 16);
}

__device__ __forceinline__ void unpack_q6_x32(const FragQ6& f0, const FragQ6& f1, int lane, Ops32& o) {
    const bool up = (lane & 16) != 0;
    const uint32_t x0[8] = {f0.q0.x, f0.q0.y, f0.q0.z, f0.q0.w, f0.q1.x, f0.q1.y, f0.q1.z, f0.q1.w};
    const uint32_t x1[8] = {f1.q0.x, f1.q0.y, f1.q0.z, f1.q0.w, f1.q1.x, f1.q1.y, f1.q1.z, f1.q1.w};
    const uint32_t h0[4] = {f0.qh.x, f0.qh.y, f0.qh.z, f0.qh.w}, h1[4] = {f1.qh.x, f1.qh.y, f1.qh.z, f1.qh.w};
    const uint32_t scw[4] = {up ? f1.sc.x : f0.sc.x, up ? f1.sc.y : f0.sc.y, up ? f1.sc.z : f0.sc.z, up ? f1.sc.w : f0.sc.w};
    const int upper = lane >> 5; /* lane half h holds k 16 h .. 16 h + 15 of every sub-block: group scale 2 j + h */
    uint32_t H[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) pair_swap(h0[u], h1[u], &H[u][0], &H[u][1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t s[2];
        pair_swap(x0[j], x1[j], &s[0], &s[1]);
        const int e = j & 1;
        const short sc = (short)(int8_t)(scw[j >> 1] >> (16 * e + 8 * upper));
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const uint32_t x = s[gp], Hh = H[j >> 1][gp];
            const uint32_t lo = ((x << 2) & 0x3C3C3C3Cu) | ((Hh << (6 - 4 * e)) & 0xC0C0C0C0u);
            const uint32_t hi = ((x >> 2) & 0x3C3C3C3Cu) | ((Hh << (4 - 4 * e)) & 0xC0C0C0C0u);
            int l0, hh0, l1, hh1;
            q6_digits(lo, sc, &l0, &hh0);
            q6_digits(hi, sc, &l1, &hh1);
            const int u = 2 * (j >> 1) + gp;
            o.bl[u][2 * e] = l0; o.bl[u][2 * e + 1] = l1;
            o.bh[u][2 * e] = hh0; o.bh[u][2 * e + 1] = hh1;
        }
    }
    o.dw = f16bits_to_f32(up ? f1.d : f0.d);
    o.dmin = 0.0f;
}

/* s_waitcnt vmcnt(n) alone (expcnt / lgkmcnt untouched): until all but this wave's n youngest vector-memory operations are done.  The
 * LDS-DMA pieces of a chunk are invisible to the compiler's own wait insertion, so the ring is guarded by hand. */
__device__ __forceinline__ void wait_vmcnt(int n) {
#define TK_VMW(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
    switch (n) {
        TK_VMW(0) TK_VMW(1) TK_VMW(2) TK_VMW(3) TK_VMW(4) TK_VMW(5) TK_VMW(6) TK_VMW(7) TK_VMW(8) TK_VMW(9) TK_VMW(10) TK_VMW(11) TK_VMW(12)
        TK_VMW(13) TK_VMW(14) TK_VMW(15) TK_VMW(16) TK_VMW(17) TK_VMW(18) TK_VMW(19) TK_VMW(20) TK_VMW(21) TK_VMW(22) TK_VMW(23) TK_VMW(24)
        TK_VMW(25) TK_VMW(26) TK_VMW(27) TK_VMW(28)
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
#undef TK_VMW
}

#define TK_MFMA32 __builtin_amdgcn_mfma_i32_32x32x32_i8
#define TK_G32_MTW 4 /* 32-row M-tiles per wave */

/* A-side operands of one 32-row M-tile of one ring block that are requested a tile AHEAD: the first four of the eight 16-byte reads of
 * the int8 MFMAs and the min-term operand; the other four are requested when the tile starts and land under its first MFMAs */
struct ATile32 { v4i a[4]; v8h mn; };
struct Ptrs32 { const uint8_t *ap, *mp, *dp; };

#define TK_G32_MT 8 /* 16-row M-tiles of one ring block: a workgroup's 128 rows */
__device__ __forceinline__ Ptrs32 block_ptrs32(const uint8_t* blk, int lane) {
    constexpr int OFF_AMN = TK_G32_MT * 4096, OFF_AD = TK_G32_MT * 4096 + TK_G32_MT * 512;
    const int h = lane >> 5;
    Ptrs32 p;
    p.ap = blk + ((lane >> 4) & 1) * 4096 + (h * 32 + (lane & 15)) * 16;            /* + t * 8192 + (u >> 1) * 1024 + (u & 1) * 256 */
    p.mp = blk + OFF_AMN + ((lane >> 4) & 1) * 512 + h * 256 + (lane & 15) * 16;   /* + t * 1024 */
    p.dp = blk + OFF_AD + h * 16;                                                  /* + t * 128 + b * 32 */
    return p;
}

template <bool Q4>
__device__ __forceinline__ void load_atile32(ATile32& T, const Ptrs32& p, int t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) T.a[u] = *(const v4i*)(p.ap + t * 8192 + (u >> 1) * 1024 + (u & 1) * 256);
    if (Q4) T.mn = *(const v8h*)(p.mp + t * 1024);
}

/* one 256-k block: this wave's 32 weight rows x its four 32-row M-tiles.  T arrives holding tile 0's operands; the operands of tile
 * t + 1 are requested as soon as the MFMAs of tile t have issued — into the same registers — and land while tile t is finished on the VALU. */
/* MTW = 32-row M-tiles a workgroup's row range holds: 3 for passes of 129..192 rows (two halves of 96 rows), 4 for 193..256 — a pass of 160
 * rows walks 3 + 3 tiles, not 4 + 4 (round 5: the 129th row cost + 63 %, profiles/r05_width_curve.txt).  A compile-time bound: 48 or 64
 * accumulators and no branch in the loop.  (One tile fewer for the second half of 129..160- and 193..224-row passes as a scalar `break` in the
 * unrolled loop was built too: 108 - 392 bytes of spill per lane at the 256-register budget of two workgroups per CU; not kept.) */
template <bool Q4, int MTW, typename Hook>
__device__ __forceinline__ void gemm_block32(const Ops32& o, ATile32& T, const Ptrs32& p, float (&acc)[MTW][16], Hook&& after_mfmas) {
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        /* P = 8 Ph + Pl (64 Ph + Pl for Q6_K) inside ONE accumulator: the high-digit chain first, its result shifted on the VALU, then the
         * low-digit chain on top of it (sixteen live registers fewer than two accumulators, and the finishing below needs no shift-add);
         * the independent min-term MFMA sits where the shift waits for the last high-digit MFMA */
        v4i A[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) A[u] = T.a[u];
        v4f da[4]; /* the block scales of this tile's rows: requested here, used after the MFMAs (a read issued in the finishing loop is waited for at once) */
#pragma unroll
        for (int u = 4; u < 8; ++u) A[u] = *(const v4i*)(p.ap + t * 8192 + (u >> 1) * 1024 + (u & 1) * 256);
#pragma unroll
        for (int b = 0; b < 4; ++b) da[b] = *(const v4f*)(p.dp + t * 128 + b * 32);
        v16i ph = zero;
        v16f cm;
        v16i pl;
#pragma unroll
        for (int u = 0; u < 8; ++u) ph = TK_MFMA32(A[u], o.bh[u], ph, 0, 0, 0);
        if (Q4) {
            const v16f fz = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(T.mn, o.bm16, fz, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pl[r] = ph[r] << (Q4 ? 3 : 6);
#pragma unroll
        for (int u = 0; u < 8; ++u) pl = TK_MFMA32(A[u], o.bl[u], pl, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < MTW) load_atile32<Q4>(T, p, t + 1);
        after_mfmas(t); /* a quarter of this wave's ring staging for the next block: LDS-DMA issue costs 60-180 cycles a piece, here they pass while
                         * the tile's MFMAs are still in the matrix pipe */
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) /* accumulator register 4 b + i = row 8 b + 4 h + i of the M-tile */
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * b + i;
                acc[t][r] = tk_fmaf(o.dw * da[b][i], (float)pl[r], acc[t][r]);
                if (Q4) acc[t][r] = tk_fmaf(-(o.dmin * da[b][i]), cm[r], acc[t][r]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool Q4> struct G32Frag { typedef FragQ4 type; };
template <> struct G32Frag<false> { typedef FragQ6 type; };
/* `tile` is wave-uniform (an SGPR pair); the per-lane offsets are 32-bit and opaque per call, so the loads take the scalar-base form and no
 * 64-bit per-lane address is hoisted out of the K loop and held across it (load_q4 / load_q6 with a lane pointer cost 12 registers there) */
template <bool Q4>
__device__ __forceinline__ typename G32Frag<Q4>::type g32_load(const uint8_t* tile, int lane) {
    unsigned lo = (unsigned)lane * 16u, ho = (unsigned)(lane & 15) * 16u;
    asm volatile("" : "+v"(lo), "+v"(ho));
    if constexpr (Q4) {
        FragQ4 f;
        f.q0 = ldg_nt(tile + lo);
        f.q1 = ldg_nt(tile + 1024 + lo);
        f.h = ldg_nt(tile + 2048 + ho);
        return f;
    } else {
        FragQ6 f;
        f.q0 = ldg_nt(tile + lo);
        f.q1 = ldg_nt(tile + 1024 + lo);
        f.qh = ldg_nt(tile + 2048 + lo);
        f.sc = ldg_nt(tile + 3072 + ho);
        f.d = *(const uint16_t*)(tile + 3328 + (ho >> 3));
        return f;
    }
}
template <bool Q4>
__device__ __forceinline__ void g32_unpack(const typename G32Frag<Q4>::type& f0, const typename G32Frag<Q4>::type& f1, int lane, Ops32& o) {
    if constexpr (Q4) unpack_q4_x32(f0, f1, lane, o);
    else unpack_q6_x32(f0, f1, lane, o);
}


/* The K loop of one wave: its two weight tiles (32 weight rows) against its four 32-row M-tiles, block by block through the ring. */
template <bool Q4, int MTW, typename StageSmall, typename StagePart>
__device__ __forceinline__ void g32_k_loop(const uint8_t* tile, size_t tile_bytes, ptrdiff_t tile_pitch, int nb, const uint8_t* ring, int slot_bytes, int lane,
                                           float (&acc)[MTW][16], StageSmall&& stage_small, StagePart&& stage_part) {
    typedef typename G32Frag<Q4>::type F;
    F f0 = g32_load<Q4>(tile, lane), f1 = g32_load<Q4>(tile + tile_pitch, lane);
    stage_small(0, 0);
    for (int part = 0; part < 4; ++part) stage_part(0, 0, part);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const bool more = b + 1 < nb;
        { /* a CU's two workgroups share its SIMDs under oldest-first arbitration: the older one runs near its solo rate, the younger one on what
           * is left and finishes alone (K loop 58 / 74 us for gate|up).  Issue priority that falls with progress lets the one that is behind
           * catch up: 63 / 71 us, launch set - 1..3 % (profiles/r03_gemm32_balance.txt) */
            const int q = 4 * b / nb;
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (more) stage_small(b + 1, (b + 1) & 1);
        const uint8_t* next = tile + (size_t)(more ? b + 1 : b) * tile_bytes; /* the last step re-requests its own tile: no branch around a load */
        const Ptrs32 bp = block_ptrs32(ring + (b & 1) * slot_bytes, lane);
        ATile32 T;
        load_atile32<Q4>(T, bp, 0); /* tile 0's operands: their LDS latency hides under the unpack */
        __builtin_amdgcn_sched_barrier(0);
        Ops32 o;
        g32_unpack<Q4>(f0, f1, lane, o);
        __builtin_amdgcn_sched_barrier(0);
        f0 = g32_load<Q4>(next, lane);
        f1 = g32_load<Q4>(next + tile_pitch, lane);
        __builtin_amdgcn_sched_barrier(0);
        /* the last block restages itself into the slot nobody reads any more: no branch around the DMA issue */
        /* the ring's four staging parts ride on the tiles' MFMA phases: one per tile, the rest with the LAST tile when there are fewer than four
         * (with the first tile the same code needs 148 bytes of spill per lane) */
        gemm_block32<Q4, MTW>(o, T, bp, acc, [&](int t) {
            stage_part(more ? b + 1 : b, (b + 1) & 1, t);
            if (MTW < 4 && t == MTW - 1)
#pragma unroll
                for (int q = MTW; q < 4; ++q) stage_part(more ? b + 1 : b, (b + 1) & 1, q);
        });
        __builtin_amdgcn_sched_barrier(0);
    }
}

/* 193..256 rows (129..192 as MTW = 3 under TK_MI355X_G32_FROM; and one row half alone).  A workgroup = 4 waves, one per SIMD = four pairs of weight tiles (128 weight rows) x ONE half
 * of the pass's rows (128 rows = four 32-row M-tiles per wave); the two row halves of the same weights are two workgroups that share a CU
 * (73 KiB of ring each) and nothing else.  The SIMD's two resident waves therefore belong to DIFFERENT workgroups: no barrier couples
 * them, so one's weight unpack (190 VALU instructions, no MFMA) drifts under the other's MFMA phases.  With both halves in one 8-wave
 * workgroup the block barrier re-aligned the partners every 256 k — both unpacked at once, then the later one finished alone:
 * 8750 cycles per block against 4352 of matrix pipe (profiles/r03_gemm32_segments.txt).
 * blockIdx -> (half, unit): consecutive workgroup ids go round the 8 XCDs, so the two halves of a unit are 8 ids apart: same XCD, same
 * L2 — the second half's weight requests hit the lines the first one brought in. */
template <int TYPES, int MTW>
__global__ __launch_bounds__(256, 2) void k_gemm32_w4a8(TkGemvArgs a, int groups, int total_row_tiles, int n_halves) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int MT = TK_G32_MT;
    constexpr int CH = MT * TK_RING_TILE_BYTES;   /* one block of the ring */
    constexpr int OFF_AMN = MT * 4096, OFF_AD = MT * 4096 + MT * 512;
    const int tid = threadIdx.x, lane = tid & 63;
    const int pair = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = a.K / a.ks / 256;
    const int nblk_total = a.K / 256;
    int half = 0, unit = blockIdx.x;
    if (n_halves == 2) {
        const int units = gridDim.x >> 1, body = (units >> 3) << 4; /* ids below `body`: XCD-aligned pairs; the rest (units not a multiple of 8) adjacent */
        if ((int)blockIdx.x < body) {
            const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
            half = idx & 1;
            unit = (idx >> 1) * 8 + x;
        } else {
            const int r = blockIdx.x - body;
            half = r & 1;
            unit = (body >> 1) + (r >> 1);
        }
    }
    const int ksi = unit % a.ks;
    const int blk0 = ksi * nb;

    /* a.swiglu (gate | up launches with one K range): the wave's 32 weight rows are gate tile i and up tile i — the same 16 hidden columns —
     * so its epilogue can form silu(gate) * up; the four waves of a workgroup take adjacent tiles (their stores join to 256 B per row) */
    int rt = a.swiglu ? 4 * (unit / a.ks) + pair : 2 * (unit / a.ks + pair * groups); /* first of this wave's two row tiles */
    const bool active = rt < (a.swiglu ? a.seg[0].row_tiles : total_row_tiles);
    if (!active) rt = 0;
    int seg = 0, row_base = 0;
    while (!a.swiglu && seg < a.nseg - 1 && rt >= a.seg[seg].row_tiles) {
        rt -= a.seg[seg].row_tiles;
        row_base += a.seg[seg].row_tiles * TK_TILE_ROWS;
        ++seg;
    }
    const int type = a.seg[seg].type;
    constexpr bool HAS4 = (TYPES & 1) != 0, HAS6 = (TYPES & 2) != 0;
    const bool is4 = HAS4 && (!HAS6 || type == TK_TYPE_Q4_K);
    const size_t tile_bytes = is4 ? (size_t)TK_Q4K_TILE_BYTES : (size_t)TK_Q6K_TILE_BYTES;
    const ptrdiff_t tile_pitch = a.swiglu ? a.seg[1].tiles - a.seg[0].tiles : (ptrdiff_t)((size_t)nblk_total * tile_bytes);
    const uint8_t* tile = a.seg[seg].tiles + ((size_t)rt * nblk_total + blk0) * tile_bytes;

    float acc[MTW][16];
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    /* ring staging: of the half's eight 16-row M-tiles, `pair` and `pair + 4` belong to this wave.  The int8 image (4 x 1 KiB pieces per
     * tile, every lane takes part: no exec masking, so the issue can sit between the MFMA phases of the tile loop) in four parts of two
     * pieces; the f16 sub-block sums (32 lanes) and the block scales (4 lanes) separately at the top of a block */
    /* rows of the pass as 32-row tiles: each half takes MTW of them (rows beyond the pass are computed on whatever the image holds and never stored) */
    const int t32_0 = half * MTW; /* this half's first 32-row tile */
    const int m0 = 2 * t32_0;     /* the half's first 16-row M-tile in the pass's activation images */
    /* the ring is staged whole (eight 16-row M-tiles per block) whatever MTW: the images of all sixteen M-tiles exist (TK_MAX_ROWS), a tile nobody
     * reads costs L2 -> LDS bytes only, and a branch around the DMA issue inside the tile loop costs registers (148 bytes of spill per lane measured) */
    /* addresses = wave-uniform base (SGPR pair) + this lane's 32-bit offset, formed where they are used: the compiler otherwise hoists five
     * 64-bit per-lane addresses out of the K loop and holds ten registers for them across it (the asm keeps the offset opaque per use) */
    auto stage_part = [&](int c, int slot, int part) {
        unsigned lane16 = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lane16));
        const int m = pair + 4 * (part >> 1);
        const uint8_t* src = (const uint8_t*)a.aq + (size_t)(m0 + m) * a.aq_ts + (size_t)(blk0 + c) * 4096 + (part & 1) * 2048;
        const auto gs = (const __attribute__((address_space(1))) void*)(src + lane16);
        const auto ls = (__attribute__((address_space(3))) void*)(lds + slot * CH + m * 4096 + (part & 1) * 2048);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 1024, 0);
    };
    auto stage_small = [&](int c, int slot) {
        uint8_t* dst = lds + slot * CH;
        unsigned lane16 = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lane16));
        for (int m = pair; m < MT; m += 4) {
            const uint8_t* sm = (const uint8_t*)(a.abs16 + (size_t)(m0 + m) * a.abs_ts + (size_t)(blk0 + c) * 256);
            if (lane < 32)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sm + lane16),
                                                 (__attribute__((address_space(3))) void*)(dst + OFF_AMN + m * 512), 16, 0, 0);
            const uint8_t* sd = (const uint8_t*)(a.ad + (size_t)(m0 + m) * a.ad_ts + (size_t)(blk0 + c) * TK_ROW_SLOTS);
            if (lane < 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sd + lane16),
                                                 (__attribute__((address_space(3))) void*)(dst + OFF_AD + m * 64), 16, 0, 0);
        }
    };

    if (!active) { /* a workgroup's spare wave slots (row tiles beyond the matrix) still stage their share of the ring */
        stage_small(0, 0);
        for (int part = 0; part < 4; ++part) stage_part(0, 0, part);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (b + 1 < nb) { stage_small(b + 1, (b + 1) & 1); for (int part = 0; part < 4; ++part) stage_part(b + 1, (b + 1) & 1, part); }
        }
        return;
    }
    if (HAS4 && is4) g32_k_loop<true, MTW>(tile, tile_bytes, tile_pitch, nb, lds, CH, lane, acc, stage_small, stage_part);
    if (HAS6 && !is4) g32_k_loop<false, MTW>(tile, tile_bytes, tile_pitch, nb, lds, CH, lane, acc, stage_small, stage_part);

    /* Epilogue: 64 accumulator registers per lane.  Stored as they stand, a store instruction writes one dword per lane (two 128-byte row
     * segments): 64 store instructions per wave, and the tail of the launch is store-ISSUE bound (exit - loop end 4 us of gate|up's 76).
     * So each 32 x 32 tile goes through LDS once — into the 4 KiB ring regions THIS wave staged itself (M-tiles `pair` and `pair + 4` of
     * the slot nobody reads any more: after the wave's own vmcnt(0) no DMA can land there, and LDS operations of one wave execute in order,
     * so no barrier is needed) — and comes back as 16 bytes per lane: lane (row r8 = l >> 3, column group l & 7) holds four consecutive
     * columns of one row, a store instruction writes eight whole 128-byte row segments, 16 instructions per wave instead of 64. */
    {
        __builtin_amdgcn_s_waitcnt(0);
        uint8_t* scr[2] = {lds + (nb & 1) * CH + pair * 4096, lds + (nb & 1) * CH + (pair + 4) * 4096};
        const int h = lane >> 5, n32 = lane & 31;
        const int n0 = a.col0 + row_base + rt * TK_TILE_ROWS + 4 * (lane & 7);
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            float* sp = (float*)scr[t & 1];
#pragma unroll
            for (int r = 0; r < 16; ++r) sp[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n32] = acc[t][r];
            if (a.swiglu) { /* columns 0..15 of the tile are gate, 16..31 up, of hidden columns 16 rt ..: h = silu(gate) * up, what k_swiglu_q8 evaluates */
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rl = 16 * i + (lane >> 2), cg = 4 * (lane & 3);
                    const v4f g = *(const v4f*)(sp + rl * 32 + cg), u = *(const v4f*)(sp + rl * 32 + 16 + cg);
                    v4f hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = tk_siluf(g[j]) * u[j];
                    const int row = (t32_0 + t) * 32 + rl;
                    if (row < a.nrows) *(v4f*)&a.out[(size_t)row * a.n_total + rt * TK_TILE_ROWS + cg] = hv;
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 8 * i + (lane >> 3);
                const v4f v = *(const v4f*)(sp + rl * 32 + 4 * (lane & 7));
                const int row = (t32_0 + t) * 32 + rl;
                if (row < a.nrows) *(v4f*)&a.out[((size_t)ksi * TK_MAX_ROWS + row) * a.n_total + n0] = v;
            }
        }
    }
}

/* compute units of the calling thread's current device (every launcher runs with the session's device current): read once per device */
static int tk_num_cu() {
    static std::atomic<int> cached[64]; /* launchers run from several host threads (sessions, schedulers) */
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        v = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
#define TK_NUM_CU tk_num_cu()

/* Dynamic LDS above 64 KiB is an opt-in that HIP keeps per (function, DEVICE): a process that drives several GPUs (the ABI takes a
 * device ordinal per handle, SURVEY 8b "Threading") must raise it on each of them.  tk_llm_prepare_device() does that once per device
 * for every instantiation, at session creation — outside any stream capture and before any host thread launches — so the launchers
 * below never touch function attributes. */
#define TK_MAX_DYN_LDS (160 * 1024)
#include <mutex>
template <typename F>
static hipError_t opt_in_lds(F* fn) { return hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, TK_MAX_DYN_LDS); }

#ifndef TK_G32_MIN_ROWS
#define TK_G32_MIN_ROWS (8 * TK_ROW_SLOTS + 1)
#endif
/* rows per pass from which the 32x32x32 kernel takes over (narrower passes run k_gemm_w4a8 at 4 .. 12 M-tiles); TK_MI355X_G32_FROM=129 .. 257
 * moves it (257 = never), read once */
static int tk_g32_from() {
    static const int v = [] { const char* e = getenv("TK_MI355X_G32_FROM"); const int x = e ? atoi(e) : 0; return x >= TK_G32_MIN_ROWS && x <= 257 ? x : 12 * TK_ROW_SLOTS + 1; }();
    return v;
}

void tk_launch_gemv(const TkGemvArgs& a, hipStream_t s) {
    int row_tiles = 0;
    for (int i = 0; i < a.nseg; ++i) row_tiles += a.seg[i].row_tiles;
    int groups = TK_NUM_CU / a.ks;            /* workgroups per K-range */
    if (groups < 1) groups = 1;
    if (groups > row_tiles) groups = row_tiles;
    int waves = (row_tiles + groups - 1) / groups;
    while (waves > 8) { groups *= 2; waves = (row_tiles + groups - 1) / groups; } /* tall matrices: more than one WG per CU */
    int types = 0;
    for (int i = 0; i < a.nseg; ++i) types |= a.seg[i].type == TK_TYPE_Q4_K ? 1 : 2;
#ifndef TK_G32_MIN_ROWS
#define TK_G32_MIN_ROWS (8 * TK_ROW_SLOTS + 1)
#endif
    /* 129..192 rows: the 16x16x64 kernel at ten / twelve M-tiles (its time grows by ~0.4 ms per 32 rows: 3.3 / 3.7 ms per decode step's launch set
     * against 4.2 of the 32x32x32 kernel at three tiles per half, profiles/r06_width_curve.txt); TK_MI355X_G32_FROM=129 keeps the round-5 split */
    if (a.nrows >= tk_g32_from()) {
        /* tk_g32_from() .. 256 rows: the 32x32x32 kernel, one type per wave (a mixed q / k / v launch needs no split): four (weight-tile pair) slots per
         * workgroup, one workgroup per row half */
        const int pairs = a.swiglu ? a.seg[0].row_tiles : row_tiles / 2; /* every segment holds a multiple of 4 row tiles: pairs never straddle segments */
        const int g32 = (pairs + 3) / 4;
        const int n_halves = a.nrows > 8 * TK_ROW_SLOTS ? 2 : 1;
        const int tiles32 = (a.nrows + 31) / 32; /* 5..8 at 129..256 rows: two halves of 3 (<= 192 rows) or 4 tiles */
        const size_t ldsb = (size_t)2 * TK_G32_MT * TK_RING_TILE_BYTES;
        const dim3 grid(g32 * a.ks * n_halves);
#define TK_G32_LAUNCH_M(TYV, MTWV) hipLaunchKernelGGL((k_gemm32_w4a8<TYV, MTWV>), grid, dim3(256), ldsb, s, a, g32, row_tiles, n_halves)
#define TK_G32_LAUNCH(TYV) do { if (n_halves == 1 || tiles32 > 6) TK_G32_LAUNCH_M(TYV, 4); else TK_G32_LAUNCH_M(TYV, 3); } while (0)
        if (types == 1) TK_G32_LAUNCH(1);
        else if (types == 2) TK_G32_LAUNCH(2);
        else TK_G32_LAUNCH(3);
#undef TK_G32_LAUNCH
#undef TK_G32_LAUNCH_M
        return;
    }
    if (a.nrows > 2 * TK_ROW_SLOTS) { /* batched passes of 33 rows and more: K-streamed activations, as many M-tiles per weight tile as the pass's rows fill */
        /* 16-row M-tiles a weight tile is multiplied against: 65..96 rows walk six, not eight */
        const int mtb = a.nrows > 14 * TK_ROW_SLOTS ? 16 : a.nrows > 12 * TK_ROW_SLOTS ? 14 : a.nrows > 10 * TK_ROW_SLOTS ? 12 : a.nrows > 8 * TK_ROW_SLOTS ? 10 : a.nrows > 6 * TK_ROW_SLOTS ? 8 : a.nrows > 4 * TK_ROW_SLOTS ? 6 : 4;
        const size_t ldsb = (size_t)2 * TK_RING_BLOCKS * mtb * TK_RING_TILE_BYTES;
        /* one weight tile per wave (two adjacent tiles per wave halve the LDS operand stream but leave one wave per SIMD: 25 % slower on
         * MI355X, profiles/r01_gemm_batched.txt) */
#define TK_GEMM_LAUNCH(MTV, TYV) hipLaunchKernelGGL((k_gemm_w4a8<MTV, TYV, 1>), dim3(groups * a.ks), dim3(64 * waves), ldsb, s, a, groups, row_tiles)
#define TK_GEMM_TY(MTV) do { if (types == 1) TK_GEMM_LAUNCH(MTV, 1); else if (types == 2) TK_GEMM_LAUNCH(MTV, 2); else TK_GEMM_LAUNCH(MTV, 3); } while (0)
        if (mtb == 4) TK_GEMM_TY(4);
        else if (mtb == 6) TK_GEMM_TY(6);
        else if (mtb == 8) TK_GEMM_TY(8);
        else if (mtb == 10) TK_GEMM_TY(10);
        else if (mtb == 12) TK_GEMM_TY(12);
        else if (mtb == 14) TK_GEMM_TY(14);
        else TK_GEMM_TY(16);
#undef TK_GEMM_TY
#undef TK_GEMM_LAUNCH
        return;
    }
    const int mt = a.nrows > TK_ROW_SLOTS ? 2 : 1;
    const size_t lds = tk_gemv_lds_bytes(a.K, a.ks, mt);
    const int nb = a.K / a.ks / 256;
    /* weight tiles in flight per wave: 2, the largest depth every (M-tiles, tile types) variant holds without spilling (compiler resource
     * report + timings on MI355X, profiles/r01_gemv_variants.txt).  A Q6-only launch with two M-tiles spills as a single-type kernel at that
     * depth; the two-type kernel does not. */
    if (mt == 2 && types == 2) types = 3;
    const int pf = nb % 2 == 0 ? 2 : 1;
#define TK_GEMV_LAUNCH(PFV, MTV, TYV, FUV) hipLaunchKernelGGL((k_gemv_w4a8<PFV, MTV, TYV, FUV>), dim3(groups * a.ks), dim3(64 * waves), lds + fuse_lds, s, a, groups, row_tiles)
#define TK_GEMV_TY(PFV, MTV, FUV) do { if (types == 1) TK_GEMV_LAUNCH(PFV, MTV, 1, FUV); else if (types == 2) TK_GEMV_LAUNCH(PFV, MTV, 2, FUV); else TK_GEMV_LAUNCH(PFV, MTV, 3, FUV); } while (0)
    const size_t fuse_lds = a.fuse == 1 ? ((size_t)a.K + 4) * sizeof(float) : 0; /* the finished row and the canonical sum's four partials */
    if (a.fuse && mt == 1 && pf == 2) { /* tk_gemv_fuses_producer() admits only such launches */
        if (a.fuse == 1) TK_GEMV_TY(2, 1, 1); else TK_GEMV_TY(2, 1, 2);
    } else if (mt == 1) { if (pf == 2) TK_GEMV_TY(2, 1, 0); else TK_GEMV_TY(1, 1, 0); }
    else { if (pf == 2) TK_GEMV_TY(2, 2, 0); else TK_GEMV_TY(1, 2, 0); }
#undef TK_GEMV_TY
#undef TK_GEMV_LAUNCH
}

bool tk_gemv_fuses_producer(int nrows, int K, int ks, int fks) {
    return nrows >= 1 && nrows <= TK_GEMV_FUSE_MAX_ROWS && ks >= 1 && K % (512 * ks) == 0 && fks >= 0 && fks <= TK_RMS_MAX_KS;
}

/* ------------------------------------------------------------------------------------------
 * q/k/v: sum K-split partials, RoPE (adjacent pairs), append K/V (f16) to the cache.
 * grid (n_kv_head, nrows); 256 threads = one q pair each (4 q heads x 64 pairs per kv head).
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ float sum_partials(const float* partial, int ks, int n_total, int row, int col) {
    float o = partial[(int64_t)row * n_total + col];
    for (int s = 1; s < ks; ++s) o = o + partial[((int64_t)s * TK_MAX_ROWS + row) * n_total + col];
    return o;
}

__global__ void k_qkv_rope_append(const float* partial, int ks, int n_total, int n_head, int n_kv_head, int head_dim, const float* rope_cos,
                                  const float* rope_sin, const int32_t* seq, const int32_t* pos, float* qbuf, uint16_t* kcache,
                                  uint16_t* vcache, int layer, int max_seq, int max_ctx) {
    const int kvh = blockIdx.x, r = blockIdx.y;
    const int half = head_dim / 2, grp = n_head / n_kv_head;
    const int QD = n_head * head_dim, KVD = n_kv_head * head_dim;
    const int p = pos[r], sq = seq[r];
    const float* cs = rope_cos + (int64_t)p * half;
    const float* sn = rope_sin + (int64_t)p * half;
    const int64_t cbase = ((((int64_t)layer * max_seq + sq) * n_kv_head + kvh) * max_ctx + p) * head_dim; /* [layer][seq][kv head][ctx][dim] */
    for (int idx = threadIdx.x; idx < (grp + 1) * half; idx += blockDim.x) {
        const int hsel = idx / half, i = idx % half;
        if (hsel < grp) {
            const int col = (kvh * grp + hsel) * head_dim + 2 * i;
            const float a = sum_partials(partial, ks, n_total, r, col);
            const float b = sum_partials(partial, ks, n_total, r, col + 1);
            qbuf[(int64_t)r * QD + col] = tk_fmaf(-b, sn[i], a * cs[i]);
            qbuf[(int64_t)r * QD + col + 1] = tk_fmaf(a, sn[i], b * cs[i]);
        } else {
            const int col = QD + kvh * head_dim + 2 * i;
            const float a = sum_partials(partial, ks, n_total, r, col);
            const float b = sum_partials(partial, ks, n_total, r, col + 1);
            kcache[cbase + 2 * i] = tk_f32_to_f16(tk_fmaf(-b, sn[i], a * cs[i]));
            kcache[cbase + 2 * i + 1] = tk_f32_to_f16(tk_fmaf(a, sn[i], b * cs[i]));
            const int vcol = QD + KVD + kvh * head_dim + 2 * i;
            vcache[cbase + 2 * i] = tk_f32_to_f16(sum_partials(partial, ks, n_total, r, vcol));
            vcache[cbase + 2 * i + 1] = tk_f32_to_f16(sum_partials(partial, ks, n_total, r, vcol + 1));
        }
    }
}

void tk_launch_qkv_rope_append(const float* partial, int ks, int n_total, int n_head, int n_kv_head, int head_dim, const float* rope_cos,
                               const float* rope_sin, const int32_t* seq, const int32_t* pos, int nrows, float* qbuf, uint16_t* kcache,
                               uint16_t* vcache, int layer, int max_seq, int max_ctx, hipStream_t s) {
    hipLaunchKernelGGL(k_qkv_rope_append, dim3(n_kv_head, nrows), dim3(256), 0, s, partial, ks, n_total, n_head, n_kv_head, head_dim, rope_cos,
                       rope_sin, seq, pos, qbuf, kcache, vcache, layer, max_seq, max_ctx);
}

/* ------------------------------------------------------------------------------------------
 * causal GQA attention.  One workgroup per (row, KV head): it serves all GQ = n_head / n_kv_head query heads that share the KV head, so
 * the head's K and V rows are read ONCE per row, and they are read as what they are in the cache layout
 * [layer][sequence][kv head][position][dim]: one contiguous run, streamed through a two-slot LDS ring by LDS-DMA in chunks of 64
 * positions (16 KiB), the next chunk in flight under the current chunk's arithmetic.  Algorithmic bytes per launch = rows x cached
 * positions x n_kv_head x head_dim x 2 B x 2 (bench.py: roofline_attention).
 *
 * Arithmetic = the oracle's order: a score is one fma chain over head_dim (thread = one position of one head, its key row read from
 * LDS); max; exp; PV and the softmax denominator as 4 interleaved partial sums over positions (t mod 4, wave j = partial j), each in
 * ascending position order, combined in j order; then the Q8 quantisation the o-projection consumes (GQ * head_dim is a multiple of
 * 256: whole Q8 blocks).  The key rows sit in LDS with their 16-byte pieces XOR-swizzled by the row number — the swizzle is applied to
 * the per-lane SOURCE address of the DMA — so 16 lanes that read piece i of 16 consecutive rows hit 16 different bank groups.
 * FUSED (decode passes: every sequence appears once): the workgroup first finishes its own q / k / v — K-split partial sums, RoPE,
 * f16 rounding, cache append — and patches its own position into the ring from LDS instead of reading its store back.
 * Non-fused (passes that hold several positions of one sequence): the cache was appended by k_qkv_rope_append before.
 * ------------------------------------------------------------------------------------------ */
#define TK_ATT_MAX_GRP 4
#define TK_ATT_TSPLIT 4 /* canonical: 4 interleaved partial sums over positions (t mod 4), added in order */
/* ring slots (template parameter SLOTS, 2 everywhere): the next chunk lands while the current one is used.  Wide passes (256 rows x 128
 * cached positions, profiles/r02_attention_variants.txt: 2 x 64 rows 53.5 us, 8 x 32 rows 86 us) want a small LDS footprint — workgroups per
 * CU pay, not bytes in flight per workgroup; narrow ones gain nothing from a deeper ring either (profiles/r03_attention_slots.txt) */
__device__ __forceinline__ float sum_partials_wide(const float* partial, int ks, int n_total, int row, int col) {
    float p[TK_RMS_MAX_KS];
#pragma unroll
    for (int s = 0; s < TK_RMS_MAX_KS; ++s)
        if (s < ks) p[s] = partial[((int64_t)s * TK_MAX_ROWS + row) * n_total + col];
    float o = p[0];
#pragma unroll
    for (int s = 1; s < TK_RMS_MAX_KS; ++s)
        if (s < ks) o = o + p[s];
    return o;
}

/* stage rows [row0, row0 + 32) of one (sequence, kv head) cache run into an LDS slot; rows past `last_row` are clamped (their
 * content is never used).  swz: XOR-swizzle the 16-byte pieces of a row by the row number (K); plain copy otherwise (V).
 * Every wave issues exactly chunk_bytes / 4096 one-KiB pieces: the counted waits rely on it. */
/* row_end (< 0: off): pieces whose rows all lie at or past it are not requested at all — only where every wait is a vmcnt(0) (two ring
 * slots), which a 128-position chunk of a short context needs (its unused three quarters would cost a 64-position context 1.5 us) */
__device__ __forceinline__ void att_stage(const uint16_t* run, int row0, int last_row, int rb /* row bytes */, uint8_t* slot, bool swz, int wave, int lane, int chunk,
                                          int row_end = -1) {
    const int pieces = chunk * rb / 1024, ppr = rb / 16; /* 1 KiB pieces per chunk; 16-byte pieces per row */
    for (int pc = wave; pc < pieces; pc += 4) {
        if (row_end >= 0 && row0 + pc * 64 / ppr >= row_end) continue; /* wave-uniform: a piece's first row (no early exit: the loop stays unrolled) */
        const int idx = pc * 64 + lane;           /* 16-byte piece index inside the chunk */
        const int r = idx / ppr, cs = idx % ppr;  /* row inside the chunk, piece slot inside the row */
        int gr = row0 + r;
        gr = gr < last_row ? gr : last_row;
        const int src = swz ? (cs ^ (r & (ppr - 1))) : cs;
        const auto gs = (const __attribute__((address_space(1))) void*)((const uint8_t*)run + (int64_t)gr * rb + src * 16);
        const auto ls = (__attribute__((address_space(3))) void*)(slot + pc * 1024);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
    }
}

template <int GQ, bool FUSED, int HD /* head_dim when it is 64 or 128 (loops unroll, LDS reads batch), 0 = any */, int CH /* positions per ring slot */, int SLOTS = 2>
#ifndef TK_ATT_WAVES
#define TK_ATT_WAVES 1
#endif
__global__ __launch_bounds__(256, TK_ATT_WAVES) void k_attention(const float* __restrict__ qbuf, const float* __restrict__ partial, int ks, int n_total,
                                                    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                    uint16_t* __restrict__ kcache, uint16_t* __restrict__ vcache, const int32_t* __restrict__ seq,
                                                    const int32_t* __restrict__ pos, int n_head, int n_kv_head, int head_dim_rt, int layer, int max_seq,
                                                    int max_ctx, TkActQ8 out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t att_lds[];
    const int head_dim = HD ? HD : head_dim_rt;
    /* blockIdx.x = block of GQ consecutive query heads; GQ is the whole KV group (its K / V rows are read once per row), or half of it
     * in passes with few rows, where more, shorter workgroups beat the second read (the launcher decides) */
    const int hb = blockIdx.x, r = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int kvh = hb * GQ / (n_head / n_kv_head);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = pos[r], T = p + 1, sq = seq[r];
    const int W = GQ * head_dim;               /* outputs of this WG */
    const int rb = head_dim * 2;               /* bytes of one cache row */
    const int half = head_dim / 2, QD = n_head * head_dim, KVD = n_kv_head * head_dim;
    const int slot_bytes = CH * rb;
    /* LDS: the ring (K chunks, then V chunks: one stream), then the float arrays */
    uint8_t* ring = att_lds;
    float* qs = (float*)(att_lds + SLOTS * slot_bytes); /* [GQ][head_dim] */
    float* sc = qs + W;                                   /* [GQ][max_ctx] */
    float* red = sc + (size_t)GQ * max_ctx;               /* [4 waves][GQ] */
    uint16_t* own = (uint16_t*)(red + 4 * TK_ATT_MAX_GRP); /* [2][head_dim]: this row's own K and V (FUSED) */
    /* the epilogue's arrays live in the ring, which is idle once the last V chunk is consumed (a barrier separates the two uses): 10 KiB
     * less LDS per workgroup = four workgroups per CU instead of three at the bench's context length */
    float* part = (float*)ring;                           /* [TSPLIT][W] partial outputs */
    float* lpart = part + TK_ATT_TSPLIT * W;              /* [TSPLIT][GQ] partial denominators */
    float* obuf = lpart + TK_ATT_TSPLIT * TK_ATT_MAX_GRP; /* [W] */
    const int64_t run0 = (((int64_t)layer * max_seq + sq) * n_kv_head + kvh) * (int64_t)max_ctx * head_dim;
    const uint16_t* krun = kcache + run0;
    const uint16_t* vrun = vcache + run0;
    const int nchunk = (T + CH - 1) / CH;
    const int total = 2 * nchunk;              /* the stream: K chunks 0 .. nchunk - 1, then V chunks 0 .. nchunk - 1 */
    const int last_row = max_ctx - 1;
    const int ppw = slot_bytes / 4096;         /* DMA pieces per wave and chunk */
    const int ppr = rb / 16;
    auto issue = [&](int j) { /* chunk j of the stream into slot j % SLOTS; always issued (a chunk with nothing cached yet re-reads clamped rows) */
        const bool is_k = j < nchunk;
        const int c = is_k ? j : j - nchunk;
        att_stage(is_k ? krun : vrun, c * CH, last_row, rb, ring + (j % SLOTS) * slot_bytes, is_k, wave, lane, CH, (SLOTS == 2 && CH > 64) ? T : -1);
    };
    /* before touching chunk j: all but the chunks issued after it have landed; everybody is done with chunk j - 1, whose slot takes
     * chunk j + SLOTS - 1 */
    auto acquire = [&](int j) {
        const int ahead = total - 1 - j < SLOTS - 2 ? total - 1 - j : SLOTS - 2;
        wait_vmcnt(ahead * ppw);
        __syncthreads();
        if (j + SLOTS - 1 < total) issue(j + SLOTS - 1);
    };

    for (int j = 0; j < SLOTS - 1 && j < total; ++j) issue(j); /* in flight under the q / k / v prologue */
    if (FUSED) {
        const float* cs = rope_cos + (int64_t)p * half;
        const float* sn = rope_sin + (int64_t)p * half;
        for (int idx = t; idx < (GQ + 2) * half; idx += 256) {
            const int hsel = idx / half, i = idx % half;
            if (hsel < GQ) {
                const int col = (hb * GQ + hsel) * head_dim + 2 * i;
                const float a = sum_partials_wide(partial, ks, n_total, r, col), b = sum_partials_wide(partial, ks, n_total, r, col + 1);
                qs[hsel * head_dim + 2 * i] = tk_fmaf(-b, sn[i], a * cs[i]);
                qs[hsel * head_dim + 2 * i + 1] = tk_fmaf(a, sn[i], b * cs[i]);
            } else if (hsel == GQ) {
                const int col = QD + kvh * head_dim + 2 * i;
                const float a = sum_partials_wide(partial, ks, n_total, r, col), b = sum_partials_wide(partial, ks, n_total, r, col + 1);
                const uint16_t k0 = tk_f32_to_f16(tk_fmaf(-b, sn[i], a * cs[i])), k1 = tk_f32_to_f16(tk_fmaf(a, sn[i], b * cs[i]));
                const uint32_t kk = (uint32_t)k0 | ((uint32_t)k1 << 16);
                *(uint32_t*)(kcache + run0 + (int64_t)p * head_dim + 2 * i) = kk;
                *(uint32_t*)(own + 2 * i) = kk;
            } else {
                const int col = QD + KVD + kvh * head_dim + 2 * i;
                const uint16_t v0 = tk_f32_to_f16(sum_partials_wide(partial, ks, n_total, r, col)), v1 = tk_f32_to_f16(sum_partials_wide(partial, ks, n_total, r, col + 1));
                const uint32_t vv = (uint32_t)v0 | ((uint32_t)v1 << 16);
                *(uint32_t*)(vcache + run0 + (int64_t)p * head_dim + 2 * i) = vv;
                *(uint32_t*)(own + head_dim + 2 * i) = vv;
            }
        }
    } else {
        for (int i = t; i < W; i += 256) qs[i] = qbuf[(int64_t)r * QD + hb * W + i];
    }
    const float att_scale = tk_divf(1.0f, tk_sqrtf((float)head_dim));

    /* ---- scores: thread = (head, position of the chunk); GQ * CHUNK threads work, one fma chain over head_dim each ---- */
    float mx = -INFINITY;
    for (int c = 0; c < nchunk; ++c) {
        acquire(c); /* the first barrier also publishes qs and own */
        uint8_t* slot = ring + (c % SLOTS) * slot_bytes;
        if (FUSED && c == nchunk - 1) { /* the row's own key (position p, always in the last chunk) comes from LDS, swizzled like the rest */
            const int rr = p - c * CH;
            if (t < ppr) *(uint4*)(slot + rr * rb + ((t ^ (rr & (ppr - 1))) * 16)) = *(const uint4*)((const uint8_t*)own + t * 16);
            __syncthreads();
        }
        for (int idx = t; idx < GQ * CH; idx += 256) {
            const int h = idx / CH, rr = idx % CH, tt = c * CH + rr;
            if (tt < T) {
                const uint8_t* kr = slot + rr * rb;
                const float* qh = qs + h * head_dim;
                float a = 0.0f;
                constexpr int KB = HD ? HD / 8 : 1; /* key pieces read per batch: the whole row when head_dim is known */
                for (int i0 = 0; i0 < ppr; i0 += KB) {
                    uint4 kv[KB];
#pragma unroll
                    for (int u = 0; u < KB; ++u) kv[u] = *(const uint4*)(kr + (((i0 + u) ^ (rr & (ppr - 1))) * 16));
#pragma unroll
                    for (int u = 0; u < KB; ++u) {
                        const v4f q0 = *(const v4f*)(qh + 8 * (i0 + u)), q1 = *(const v4f*)(qh + 8 * (i0 + u) + 4);
                        const uint32_t kw[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float kf = f16bits_to_f32((kw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                            a = tk_fmaf(e < 4 ? q0[e] : q1[e - 4], kf, a);
                        }
                    }
                }
                const float s = a * att_scale;
                sc[(size_t)h * max_ctx + tt] = s;
                mx = tk_fmaxf(mx, s);
            }
        }
    }
    /* row maximum per head: thread idx = h * CHUNK + rr served head h in every chunk (GQ * CHUNK <= 256 threads: one idx per thread) */
    if (CH <= 64) {
        float m = mx;
        for (int s = CH / 2; s >= 1; s >>= 1) m = tk_fmaxf(m, wave_xor_f(m, s)); /* over the CHUNK lanes of one head */
        if ((lane & (CH - 1)) == 0 && t < GQ * CH) red[t / CH] = m;
    } else { /* a head's CHUNK threads span CH / 64 whole waves: per-wave maxima, joined below (a maximum does not depend on the order) */
        float m = mx;
        for (int s = 32; s >= 1; s >>= 1) m = tk_fmaxf(m, wave_xor_f(m, s));
        if (lane == 0) red[TK_ATT_MAX_GRP + wave] = m; /* the second row of red[4][GQ <= 4]: free until the epilogue's lpart, which lives in the ring */
    }
    __syncthreads(); /* every score is written */
    if (CH > 64 && t < GQ) {
        float m = red[TK_ATT_MAX_GRP + t * (CH / 64)];
        for (int w = 1; w < CH / 64; ++w) m = tk_fmaxf(m, red[TK_ATT_MAX_GRP + t * (CH / 64) + w]);
        red[t] = m;
    }
    if (CH > 64) __syncthreads();
    for (int h = 0; h < GQ; ++h) {
        const float m = red[h];
        for (int tt = t; tt < T; tt += 256) sc[(size_t)h * max_ctx + tt] = tk_expf(sc[(size_t)h * max_ctx + tt] - m);
    }
    /* ---- PV: wave j takes positions t = j (mod 4); lane owns dims (2 lane, 2 lane + 1) [+ 128 k] of every head ---- */
    float acc[GQ][2][2], l[GQ]; /* [head][dim block of 128][pair] */
#pragma unroll
    for (int h = 0; h < GQ; ++h) { l[h] = 0.0f; acc[h][0][0] = acc[h][0][1] = acc[h][1][0] = acc[h][1][1] = 0.0f; }
    for (int c = 0; c < nchunk; ++c) {
        acquire(nchunk + c); /* the first barrier also publishes the probabilities */
        uint8_t* slot = ring + ((nchunk + c) % SLOTS) * slot_bytes;
        if (FUSED && c == nchunk - 1) {
            const int rr = p - c * CH;
            if (t < ppr) *(uint4*)(slot + rr * rb + t * 16) = *(const uint4*)((const uint8_t*)(own + head_dim) + t * 16);
            __syncthreads();
        }
        const int t_end = T - c * CH < CH ? T - c * CH : CH;
        /* positions in batches of PB: every LDS read of a batch is issued before its first fma (out-of-range slots of the last batch
         * read a clamped row and enter with probability 0: fma(0, v, acc) == acc and l + 0 == l exactly, the canonical order is untouched) */
        constexpr int PB = 4;
        for (int rr0 = wave; rr0 < t_end; rr0 += PB * TK_ATT_TSPLIT) {
            float pr[PB][GQ];
            uint32_t vv[PB][2];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int rr = rr0 + u * TK_ATT_TSPLIT;
                const bool live = rr < t_end;
                const int rc = live ? rr : t_end - 1;
#pragma unroll
                for (int h = 0; h < GQ; ++h) pr[u][h] = live ? sc[(size_t)h * max_ctx + c * CH + rc] : 0.0f;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int d0 = 2 * lane + 128 * db;
                    vv[u][db] = d0 < head_dim ? *(const uint32_t*)(slot + rc * rb + d0 * 2) : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    if (2 * lane + 128 * db < head_dim) {
                        const float v0 = f16bits_to_f32(vv[u][db] & 0xffffu), v1 = f16bits_to_f32(vv[u][db] >> 16);
#pragma unroll
                        for (int h = 0; h < GQ; ++h) {
                            acc[h][db][0] = tk_fmaf(pr[u][h], v0, acc[h][db][0]);
                            acc[h][db][1] = tk_fmaf(pr[u][h], v1, acc[h][db][1]);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < GQ; ++h) l[h] = l[h] + pr[u][h];
            }
        }
    }
    __syncthreads(); /* every wave is done with the ring: it now holds the epilogue's arrays */
#pragma unroll
    for (int h = 0; h < GQ; ++h) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int d0 = 2 * lane + 128 * db;
            if (d0 < head_dim) {
                part[wave * W + h * head_dim + d0] = acc[h][db][0];
                part[wave * W + h * head_dim + d0 + 1] = acc[h][db][1];
            }
        }
        if (lane == 0) lpart[wave * TK_ATT_MAX_GRP + h] = l[h];
    }
    __syncthreads();
    for (int i = t; i < W; i += 256) {
        const int h = i / head_dim;
        const float a = ((part[i] + part[W + i]) + part[2 * W + i]) + part[3 * W + i];
        const float ll = ((lpart[h] + lpart[TK_ATT_MAX_GRP + h]) + lpart[2 * TK_ATT_MAX_GRP + h]) + lpart[3 * TK_ATT_MAX_GRP + h];
        obuf[i] = tk_divf(a, ll);
    }
    __syncthreads();
    for (int c = t; c < W / 8; c += 256) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = obuf[8 * c + i];
        quantize_chunk8(v, hb * (W / 8) + c, r, out);
    }
}

/* ------------------------------------------------------------------------------------------
 * Narrow decode passes (at most one workgroup per CU: 1 .. 16 rows of Mistral-7B): the same attention, arithmetic for arithmetic, as
 * k_attention<2, true, 128, ...>, laid out for LATENCY.  There the launch is a chain of ~11 barrier-separated phases of four waves (two
 * K chunks, two V chunks, their own-row patches, max, exp, three epilogue steps: 15 us per layer of a one-runner decode step, a quarter of
 * the token for under 1 % of its bytes).  Here a workgroup has 16 waves and the CU's whole LDS: every cached key and value row of the
 * context (up to `cap` positions, ~290 at a 512-position window; longer contexts go through the same code chunk by chunk) is requested by
 * LDS-DMA in the first instructions, under the q / k / v prologue; then ONE score phase (wave = 64 positions of one head), max, exp, ONE
 * value phase (wave = one head x one of the four canonical interleaved partial sums), combine, quantise: six barriers.
 * Workgroup = (row, two query heads of one KV group); grid (n_head / 2, rows).
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(1024) void k_attention_narrow(const float* __restrict__ partial, int ks, int n_total, const float* __restrict__ rope_cos,
                                                            const float* __restrict__ rope_sin, uint16_t* __restrict__ kcache, uint16_t* __restrict__ vcache,
                                                            const int32_t* __restrict__ seq, const int32_t* __restrict__ pos, int n_head, int n_kv_head, int layer,
                                                            int max_seq, int max_ctx, int cap, TkActQ8 out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t att_lds[];
    constexpr int HD = 128, GQ = 2, W = GQ * HD, RB = HD * 2, PPR = RB / 16; /* outputs per workgroup, bytes per cache row, 16-byte pieces per row */
    const int hb = blockIdx.x, r = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kvh = hb * GQ / (n_head / n_kv_head);
    const int p = pos[r], T = p + 1, sq = seq[r];
    const int half = HD / 2, QD = n_head * HD, KVD = n_kv_head * HD;
    uint8_t* kbuf = att_lds;                                  /* [cap][256 B], 16-byte pieces XOR-swizzled by the row number */
    uint8_t* vbuf = att_lds + (size_t)cap * RB;               /* [cap][256 B] */
    float* qs = (float*)(att_lds + (size_t)2 * cap * RB);     /* [GQ][HD] */
    float* sc = qs + W;                                       /* [GQ][max_ctx] */
    float* red = sc + (size_t)GQ * ((max_ctx + 3) & ~3);      /* [16] wave maxima, [GQ] head maxima; 16-byte aligned */
    uint16_t* own = (uint16_t*)(red + 32);                    /* [2][HD]: this row's own K and V */
    float* part = (float*)kbuf;                               /* epilogue, aliases the key rows: [4][W] partial outputs, [4][GQ] denominators, [W] outputs */
    float* lpart = part + TK_ATT_TSPLIT * W;
    float* obuf = lpart + TK_ATT_TSPLIT * TK_ATT_MAX_GRP;
    const int64_t run0 = (((int64_t)layer * max_seq + sq) * n_kv_head + kvh) * (int64_t)max_ctx * HD;
    const int nchunk = (T + cap - 1) / cap;
    /* rows [row0, row0 + cap) of a cache run into a buffer: 1 KiB pieces (four rows) dealt to the 16 waves; pieces wholly past the context are
     * not requested, rows past it inside a piece are clamped (never used) */
    auto stage = [&](const uint16_t* run, int row0, uint8_t* buf, bool swz) {
        const int pieces = cap * RB / 1024;
        for (int pc = wave; pc < pieces; pc += 16) {
            if (row0 + pc * 4 >= T) break; /* wave-uniform */
            const int rr = pc * 4 + (lane >> 4), cs = lane & 15;
            int gr = row0 + rr;
            gr = gr < max_ctx - 1 ? gr : max_ctx - 1;
            const int src = swz ? (cs ^ (rr & (PPR - 1))) : cs;
            const auto gs = (const __attribute__((address_space(1))) void*)((const uint8_t*)run + (int64_t)gr * RB + src * 16);
            const auto ls = (__attribute__((address_space(3))) void*)(buf + pc * 1024);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
        }
    };
    stage(kcache + run0, 0, kbuf, true);
    stage(vcache + run0, 0, vbuf, false);
    /* prologue: this row's q (two heads), k and v — K-split sums, RoPE, f16 rounding, cache append (k_attention's FUSED prologue) */
    if (t < (GQ + 2) * half) {
        const float* cs = rope_cos + (int64_t)p * half;
        const float* sn = rope_sin + (int64_t)p * half;
        const int hsel = t / half, i = t % half;
        if (hsel < GQ) {
            const int col = (hb * GQ + hsel) * HD + 2 * i;
            const float a = sum_partials_wide(partial, ks, n_total, r, col), b = sum_partials_wide(partial, ks, n_total, r, col + 1);
            qs[hsel * HD + 2 * i] = tk_fmaf(-b, sn[i], a * cs[i]);
            qs[hsel * HD + 2 * i + 1] = tk_fmaf(a, sn[i], b * cs[i]);
        } else if (hsel == GQ) {
            const int col = QD + kvh * HD + 2 * i;
            const float a = sum_partials_wide(partial, ks, n_total, r, col), b = sum_partials_wide(partial, ks, n_total, r, col + 1);
            const uint16_t k0 = tk_f32_to_f16(tk_fmaf(-b, sn[i], a * cs[i])), k1 = tk_f32_to_f16(tk_fmaf(a, sn[i], b * cs[i]));
            const uint32_t kk = (uint32_t)k0 | ((uint32_t)k1 << 16);
            if ((hb * GQ) % (n_head / n_kv_head) == 0) *(uint32_t*)(kcache + run0 + (int64_t)p * HD + 2 * i) = kk; /* one workgroup of the KV group appends */
            *(uint32_t*)(own + 2 * i) = kk;
        } else {
            const int col = QD + KVD + kvh * HD + 2 * i;
            const uint16_t v0 = tk_f32_to_f16(sum_partials_wide(partial, ks, n_total, r, col)), v1 = tk_f32_to_f16(sum_partials_wide(partial, ks, n_total, r, col + 1));
            const uint32_t vv = (uint32_t)v0 | ((uint32_t)v1 << 16);
            if ((hb * GQ) % (n_head / n_kv_head) == 0) *(uint32_t*)(vcache + run0 + (int64_t)p * HD + 2 * i) = vv;
            *(uint32_t*)(own + HD + 2 * i) = vv;
        }
    }
    const float att_scale = tk_divf(1.0f, tk_sqrtf((float)HD));
    /* ---- scores: wave w = head w & 1, positions (w >> 1) * 64 + lane of the chunk; one fma chain over head_dim per (head, position) ---- */
    const int sh = wave & 1;
    float mx = -INFINITY;
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0) { __syncthreads(); stage(kcache + run0, c * cap, kbuf, true); } /* everybody is done with the previous chunk's key rows */
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): this wave's pieces (and its prologue loads) */
        __syncthreads();                     /* everybody's; qs and own are written */
        if (c == nchunk - 1) { /* the row's own key (position p, in the last chunk) comes from LDS, swizzled like the rest */
            const int rr = p - c * cap;
            if (t < PPR) *(uint4*)(kbuf + rr * RB + ((t ^ (rr & (PPR - 1))) * 16)) = *(const uint4*)((const uint8_t*)own + t * 16);
            if (t >= 64 && t < 64 + PPR && nchunk == 1) *(uint4*)(vbuf + rr * RB + (t - 64) * 16) = *(const uint4*)((const uint8_t*)(own + HD) + (t - 64) * 16);
            __syncthreads();
        }
        const int tc = T - c * cap < cap ? T - c * cap : cap;
        for (int rr = (wave >> 1) * 64 + lane; rr < tc; rr += 512) {
            const uint8_t* kr = kbuf + rr * RB;
            const float* qh = qs + sh * HD;
            uint4 kv[PPR];
#pragma unroll
            for (int u = 0; u < PPR; ++u) kv[u] = *(const uint4*)(kr + ((u ^ (rr & (PPR - 1))) * 16));
            float a = 0.0f;
#pragma unroll
            for (int u = 0; u < PPR; ++u) {
                const v4f q0 = *(const v4f*)(qh + 8 * u), q1 = *(const v4f*)(qh + 8 * u + 4);
                const uint32_t kw[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float kf = f16bits_to_f32((kw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                    a = tk_fmaf(e < 4 ? q0[e] : q1[e - 4], kf, a);
                }
            }
            const float sv = a * att_scale;
            sc[(size_t)sh * max_ctx + c * cap + rr] = sv;
            mx = tk_fmaxf(mx, sv);
        }
    }
    for (int s2 = 32; s2 >= 1; s2 >>= 1) mx = tk_fmaxf(mx, wave_xor_f(mx, s2));
    if (lane == 0) red[wave] = mx;
    __syncthreads(); /* every score and every wave maximum is written */
    if (t < GQ) {
        float m = red[t];
        for (int w = 1; w < 8; ++w) m = tk_fmaxf(m, red[t + 2 * w]); /* a maximum does not depend on the order */
        red[16 + t] = m;
    }
    __syncthreads();
    for (int i = t; i < GQ * T; i += 1024) {
        const int h = i / T, tt = i - h * T;
        sc[(size_t)h * max_ctx + tt] = tk_expf(sc[(size_t)h * max_ctx + tt] - red[16 + h]);
    }
    /* ---- PV: wave w < 8 = head w >> 2, partial sum w & 3 (positions = w & 3 mod 4, ascending); lane owns dims 2 lane, 2 lane + 1 ---- */
    const int ph = wave >> 2, pj = wave & 3;
    float acc0 = 0.0f, acc1 = 0.0f, lsum = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0 || nchunk > 1) { /* long contexts: the value rows arrive chunk by chunk too (chunk 0 was requested at the start) */
            if (c > 0) { __syncthreads(); stage(vcache + run0, c * cap, vbuf, false); }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if (c == nchunk - 1) {
                const int rr = p - c * cap;
                if (t < PPR) *(uint4*)(vbuf + rr * RB + t * 16) = *(const uint4*)((const uint8_t*)(own + HD) + t * 16);
            }
        }
        __syncthreads(); /* the probabilities (and a patched value row) are written */
        if (wave < 8) {
            const int tc = T - c * cap < cap ? T - c * cap : cap;
            const float* prow = sc + (size_t)ph * max_ctx + c * cap;
            constexpr int PB = 8;
            for (int rr0 = pj; rr0 < tc; rr0 += PB * TK_ATT_TSPLIT) {
                float pr[PB];
                uint32_t vv[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) { /* slots past the context read the last live row and enter with probability 0: exact no-ops */
                    const int rr = rr0 + u * TK_ATT_TSPLIT;
                    const bool live = rr < tc;
                    const int rc = live ? rr : tc - 1;
                    pr[u] = live ? prow[rc] : 0.0f;
                    vv[u] = *(const uint32_t*)(vbuf + rc * RB + lane * 4);
                }
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    acc0 = tk_fmaf(pr[u], f16bits_to_f32(vv[u] & 0xffffu), acc0);
                    acc1 = tk_fmaf(pr[u], f16bits_to_f32(vv[u] >> 16), acc1);
                    lsum = lsum + pr[u];
                }
            }
        }
    }
    __syncthreads(); /* every wave is done with the key and value rows: the key buffer now holds the epilogue's arrays */
    if (wave < 8) {
        part[pj * W + ph * HD + 2 * lane] = acc0;
        part[pj * W + ph * HD + 2 * lane + 1] = acc1;
        if (lane == 0) lpart[pj * TK_ATT_MAX_GRP + ph] = lsum;
    }
    __syncthreads();
    if (t < W) {
        const int h = t / HD;
        const float a = ((part[t] + part[W + t]) + part[2 * W + t]) + part[3 * W + t];
        const float ll = ((lpart[h] + lpart[TK_ATT_MAX_GRP + h]) + lpart[2 * TK_ATT_MAX_GRP + h]) + lpart[3 * TK_ATT_MAX_GRP + h];
        obuf[t] = tk_divf(a, ll);
    }
    __syncthreads();
    if (t < W / 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = obuf[8 * t + i];
        quantize_chunk8(v, hb * (W / 8) + t, r, out);
    }
}

/* ------------------------------------------------------------------------------------------
 * Multi-position passes (prompt chunks: several positions of one sequence in a pass; the reference feeds whole prompts —
 * src/ai_models/tk_runner_streaming.c:20-40 — and the cortex's context strings run to hundreds of tokens, budget 2 048:
 * src/cortex/tk_cortex_main.c:1334).  k_attention gives every row a workgroup of its own that streams the sequence's keys and values and
 * walks them with one fma chain per thread: 256 rows of ONE sequence read the same rows 256 times and spend 256 VALU fmas per (row, head,
 * position) — 177 us per layer for 256 rows over 450 positions (profiles/r05_prefill_attention.txt).  Here 16 consecutive rows of a
 * sequence share a workgroup and the arithmetic moves to the fp32 matrix pipe WITHOUT changing a bit of it:
 *   - v_mfma_f32_16x16x4_f32 is a k-ascending fma chain per output (what tk_gemm_tiled relies on), so a score — one fma chain over
 *     head_dim from 0.0f, then * scale — is 32 chained MFMAs of a [16 rows] x [16 positions] tile; an output element depends only on its
 *     own row and column, so which rows share a tile changes nothing;
 *   - the canonical PV order — four interleaved partial sums over positions (t mod 4 = j), each an ascending fma chain, joined in j order
 *     — is wave j's job: one MFMA step takes positions (t0 + j, + 4, + 8, + 12) of a 16-position tile, ascending, for all 16 rows;
 *     probabilities of positions a row does not see (causal mask, padding) enter as 0: fma(0, v, acc) == acc and l + 0 == l exactly;
 *   - no score storage: pass 1 walks the keys for the row maxima, pass 2 recomputes the (identical) scores block by block, so LDS does
 *     not grow with the context.
 * Workgroup = one 16-row tile of one sequence (table built on the device by k_att_tiles from the pass's sequence ids) x one 256-wide
 * block of the attention output (256 / head_dim query heads of one KV head: whole Q8 blocks for the o-projection); 4 waves per head:
 * wave w computes score tile w of every 64-position block and owns PV class j = w.  Keys sit in LDS in B-operand order (lane group g's
 * dims g, g + 4, ... contiguous), values row-major with padded rows; both are fetched one block ahead into registers.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_att_tiles(const int32_t* __restrict__ seq, int nrows, int32_t* __restrict__ tiles) {
    __shared__ int sq[TK_MAX_ROWS], run0[TK_MAX_ROWS], cnt[TK_MAX_ROWS];
    const int t = threadIdx.x;
    sq[t] = t < nrows ? seq[t] : -1;
    __syncthreads();
    run0[t] = (t < nrows && (t == 0 || sq[t] != sq[t - 1])) ? t : 0; /* start of the run a row belongs to: running maximum of the starts */
    __syncthreads();
    for (int d = 1; d < TK_MAX_ROWS; d <<= 1) {
        const int v = t >= d ? (run0[t] > run0[t - d] ? run0[t] : run0[t - d]) : run0[t];
        __syncthreads();
        run0[t] = v;
        __syncthreads();
    }
    const bool head = t < nrows && ((t - run0[t]) & 15) == 0; /* a tile starts every 16 rows of a run */
    int len = 0;
    if (head) { len = 1; while (len < 16 && t + len < nrows && sq[t + len] == sq[t]) ++len; }
    cnt[t] = head ? 1 : 0;
    __syncthreads();
    for (int d = 1; d < TK_MAX_ROWS; d <<= 1) { /* inclusive prefix sum: the tile's index */
        const int v = t >= d ? cnt[t] + cnt[t - d] : cnt[t];
        __syncthreads();
        cnt[t] = v;
        __syncthreads();
    }
    if (head) tiles[cnt[t]] = t | (len << 16); /* tiles[1 ..] */
    if (t == TK_MAX_ROWS - 1) tiles[0] = cnt[t];
}

template <int HD>
struct TkPrefillLds {
    static constexpr int HPB = 256 / HD;                 /* query heads per workgroup */
    static constexpr int KCH = HD / 2 + 16;              /* bytes of one (position, lane group) key run: HD / 4 halves + 16 B of padding */
    static constexpr int VROW = HD * 2 + 16;             /* bytes of a padded value row */
    static constexpr int EROW = 20;                      /* floats of a padded probability row (16 rows of the tile per position) */
    static constexpr size_t k_off = 0;
    static constexpr size_t v_off = (size_t)64 * 4 * KCH;
    static constexpr size_t e_off = v_off + (size_t)64 * VROW;
    static constexpr size_t o_off = e_off + (size_t)HPB * 64 * EROW * 4;
    static constexpr size_t l_off = o_off + (size_t)16 * 256 * 4;     /* [HPB][16] denominators */
    static constexpr size_t m_off = l_off + (size_t)HPB * 16 * 4;     /* [HPB][4][16] per-wave maxima */
    static constexpr size_t bytes = m_off + (size_t)HPB * 4 * 16 * 4;
};

template <int HD>
__global__ __launch_bounds__((256 / HD) * 256) void k_attention_prefill(const float* __restrict__ qbuf, const uint16_t* __restrict__ kcache,
                                                                      const uint16_t* __restrict__ vcache, const int32_t* __restrict__ seq,
                                                                      const int32_t* __restrict__ pos, const int32_t* __restrict__ tiles, int n_head,
                                                                      int n_kv_head, int layer, int max_seq, int max_ctx, TkActQ8 out) {
    using L = TkPrefillLds<HD>;
    constexpr int HPB = L::HPB, NT = HPB * 256, KCH = L::KCH, VROW = L::VROW, EROW = L::EROW;
    constexpr int PIECES = 64 * HD / 8, PPT = (PIECES + NT - 1) / NT, PPR = HD / 8; /* 16-byte pieces of a 64-position block, per thread, per row */
    extern __shared__ __attribute__((aligned(16))) uint8_t pf_lds[];
    const int n_tiles = tiles[0];
    /* the grid holds rows / 16 + 8 tile slots (a graph fixes it at capture); a pass of many short runs has more tiles: they are walked */
    for (int tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
    const int tw = tiles[1 + tile];
    const int row0 = tw & 0xffff, nr = tw >> 16;
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hsel = wid >> 2, w = wid & 3;
    const int n16 = lane & 15, G = lane >> 4;
    const int head = blockIdx.x * HPB + hsel, QD = n_head * HD;
    const int kvh = (blockIdx.x * HPB) / (n_head / n_kv_head);
    const int sq = seq[row0];
    uint8_t* kbuf = pf_lds + L::k_off;
    uint8_t* vbuf = pf_lds + L::v_off;
    float* ebuf = (float*)(pf_lds + L::e_off) + (size_t)hsel * 64 * EROW;
    float* obuf = (float*)(pf_lds + L::o_off);
    float* lbuf = (float*)(pf_lds + L::l_off) + hsel * 16;
    float* mred = (float*)(pf_lds + L::m_off) + hsel * 64;
    const int64_t run0 = (((int64_t)layer * max_seq + sq) * n_kv_head + kvh) * (int64_t)max_ctx * HD;
    const uint16_t* krun = kcache + run0;
    const uint16_t* vrun = vcache + run0;

    int Trow[4], Tmax = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) Trow[i] = 4 * G + i < nr ? pos[row0 + 4 * G + i] + 1 : 0;
    for (int m = 0; m < nr; ++m) { const int T = pos[row0 + m] + 1; Tmax = T > Tmax ? T : Tmax; }
    const int NB = (Tmax + 63) / 64;
    const float att_scale = tk_divf(1.0f, tk_sqrtf((float)HD));

    /* this head's queries as the A operand: lane (row n16, group G) holds q[4 s + G], s = 0 .. HD / 4 - 1 */
    float qa[HD / 4];
    {
        const float* qr = qbuf + (int64_t)(row0 + (n16 < nr ? n16 : 0)) * QD + head * HD + G;
#pragma unroll
        for (int s = 0; s < HD / 4; ++s) qa[s] = n16 < nr ? qr[4 * s] : 0.0f;
    }

    uint4 kreg[PPT], vreg[PPT];
    auto fetch = [&](const uint16_t* run, int b, uint4* dst) { /* block b's rows (clamped to the last one any row of the tile sees) into registers */
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int pid = t + u * NT;
            if (PIECES % NT == 0 || pid < PIECES) {
                int pr = 64 * b + pid / PPR;
                pr = pr < Tmax ? pr : Tmax - 1;
                dst[u] = *(const uint4*)(run + (int64_t)pr * HD + 8 * (pid % PPR));
            }
        }
    };
    auto put_k = [&]() { /* dims 8 q + e of position pn: group g = e & 3 takes (e = g, e = 4 + g) = its steps 2 q, 2 q + 1 */
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int pid = t + u * NT;
            if (PIECES % NT == 0 || pid < PIECES) {
                const int pn = pid / PPR, q8 = pid % PPR;
                const uint32_t wv[4] = {kreg[u].x, kreg[u].y, kreg[u].z, kreg[u].w};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t lo = (wv[g >> 1] >> (16 * (g & 1))) & 0xffffu, hi = (wv[2 + (g >> 1)] >> (16 * (g & 1))) & 0xffffu;
                    *(uint32_t*)(kbuf + (size_t)(pn * 4 + g) * KCH + 4 * q8) = lo | (hi << 16);
                }
            }
        }
    };
    auto put_v = [&]() {
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int pid = t + u * NT;
            if (PIECES % NT == 0 || pid < PIECES) *(uint4*)(vbuf + (size_t)(pid / PPR) * VROW + 16 * (pid % PPR)) = vreg[u];
        }
    };
    auto score_tile = [&]() -> v4f { /* rows x positions 16 w .. 16 w + 15 of the staged block: one fma chain over head_dim per element */
        v4f d = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint8_t* kp = kbuf + (size_t)((16 * w + n16) * 4 + G) * KCH;
#pragma unroll
        for (int q = 0; q < HD / 32; ++q) {
            const uint4 kk = *(const uint4*)(kp + 16 * q);
            const uint32_t wv[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[8 * q + e], f16bits_to_f32((wv[e >> 1] >> (16 * (e & 1))) & 0xffffu), d, 0, 0, 0);
        }
        return d;
    };

    /* ---- pass 1: row maxima ---- */
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    v4f d_one = {0.0f, 0.0f, 0.0f, 0.0f}; /* a context of one block (<= 64 positions): its scores and its staged keys serve pass 2 as they are */
    fetch(krun, 0, kreg);
    if (NB == 1) fetch(vrun, 0, vreg);
    for (int b = 0; b < NB; ++b) {
        __syncthreads(); /* the previous block's (or tile's) readers are done */
        put_k();
        if (b + 1 < NB) fetch(krun, b + 1, kreg);
        __syncthreads();
        const v4f d = score_tile();
        d_one = d;
        const int tp = 64 * b + 16 * w + n16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tp < Trow[i]) mx[i] = tk_fmaxf(mx[i], d[i] * att_scale);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        for (int s = 8; s >= 1; s >>= 1) mx[i] = tk_fmaxf(mx[i], wave_xor_f(mx[i], s));
        if (n16 == 0) mred[w * 16 + 4 * G + i] = mx[i];
    }
    __syncthreads();
    float rowmax[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        rowmax[i] = tk_fmaxf(tk_fmaxf(mred[4 * G + i], mred[16 + 4 * G + i]), tk_fmaxf(mred[32 + 4 * G + i], mred[48 + 4 * G + i]));

    /* ---- pass 2: probabilities and PV, class j = w ---- */
    v4f acc[HD / 16];
#pragma unroll
    for (int c = 0; c < HD / 16; ++c) acc[c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    float l = 0.0f; /* lanes G == 0: row n16's denominator of class w */
    if (NB > 1) { fetch(krun, 0, kreg); fetch(vrun, 0, vreg); }
    for (int b = 0; b < NB; ++b) {
        __syncthreads(); /* the previous block's keys, values and probabilities are consumed */
        if (NB > 1) put_k();
        put_v();
        if (b + 1 < NB) { fetch(krun, b + 1, kreg); fetch(vrun, b + 1, vreg); }
        __syncthreads();
        {
            const v4f d = NB > 1 ? score_tile() : d_one;
            const int tp = 64 * b + 16 * w + n16;
            v4f e;
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = tp < Trow[i] ? tk_expf(d[i] * att_scale - rowmax[i]) : 0.0f;
            *(v4f*)(ebuf + (size_t)(16 * w + n16) * EROW + 4 * G) = e;
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int prow = 16 * st + w + 4 * G; /* this lane group's position of the step: t0 + j + 4 g */
            const float a = ebuf[(size_t)prow * EROW + n16];
            const uint8_t* vp = vbuf + (size_t)prow * VROW + 4 * n16;
#pragma unroll
            for (int c = 0; c < HD / 32; ++c) {
                const uint32_t vv = *(const uint32_t*)(vp + 64 * c);
                acc[2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f16bits_to_f32(vv & 0xffffu), acc[2 * c], 0, 0, 0);
                acc[2 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f16bits_to_f32(vv >> 16), acc[2 * c + 1], 0, 0, 0);
            }
            if (G == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) l = l + ebuf[(size_t)(16 * st + w + 4 * g) * EROW + n16];
            }
        }
    }
    /* ---- the four classes joined in order: ((p0 + p1) + p2) + p3, divided by ((l0 + l1) + l2) + l3 ---- */
    for (int step = 0; step < 4; ++step) {
        __syncthreads();
        if (w == step) {
            if (G == 0) lbuf[n16] = step == 0 ? l : lbuf[n16] + l;
            if (step < 3) {
#pragma unroll
                for (int c = 0; c < HD / 16; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* o = obuf + (size_t)(4 * G + i) * 256 + hsel * HD + 32 * (c >> 1) + 2 * n16 + (c & 1);
                        *o = step == 0 ? acc[c][i] : *o + acc[c][i];
                    }
            }
        }
    }
    __syncthreads();
    if (w == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ll = lbuf[4 * G + i];
#pragma unroll
            for (int c = 0; c < HD / 16; ++c) {
                float* o = obuf + (size_t)(4 * G + i) * 256 + hsel * HD + 32 * (c >> 1) + 2 * n16 + (c & 1);
                *o = tk_divf(*o + acc[c][i], ll);
            }
        }
    }
    __syncthreads();
    /* the o-projection's input: whole Q8 blocks (256 outputs of a row = this workgroup's heads), 32 lanes per block */
    for (int idx = t; idx < nr * 32; idx += NT) {
        const int m = idx >> 5, ch = idx & 31;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = obuf[(size_t)m * 256 + 8 * ch + i];
        quantize_chunk8(v, blockIdx.x * 32 + ch, row0 + m, out);
    }
    } /* tiles */
}

/* ------------------------------------------------------------------------------------------
 * Decode passes of few rows over LONG contexts (the reference runs a 4 096-position window with a 2 048-token prompt budget:
 * src/ai_models/tk_runner_lifecycle.c:48, src/cortex/tk_cortex_main.c:1334).  k_attention_narrow walks a long context chunk by chunk inside
 * one workgroup per pair of heads — ten to eighteen barrier-separated DMA round trips at 2 048 .. 4 000 positions: 51 / 96 us per layer at one
 * row (164 GB/s: a latency chain), as much as the whole step's mat-vec launches.  Here the same arithmetic is spread over the chip in three
 * launches, none of which waits on a chunk loop:
 *   k_att_scores_long   workgroup = (row, KV head, 64-position block): first the group's queries (K-split sums, RoPE) and — in the
 *                       context's last block — this row's own key and value (f16 rounding, cache append, the key patched into the staged
 *                       block from LDS): k_qkv_rope_append's arithmetic; then the block's scores of the group's query heads on
 *                       v_mfma_f32_16x16x4_f32 (the heads are the tile's rows; a score is the canonical fma chain over head_dim, * scale),
 *                       written to a [row][head][position] fp32 buffer;
 *   k_att_pv_chain      workgroup = one wave = (row, head, class j): row maximum (order-free), e = exp(s - m), then the canonical class
 *                       chain — positions j, j + 4, ... ascending, acc = fma(e, v, acc), l = l + e — as ONE sequential chain per lane (a lane
 *                       owns two dims), 32 value rows requested ahead; partial outputs and denominators to a scratch buffer;
 *   k_att_pv_join       the four classes joined in order, division, Q8 quantisation.
 * Bit-identical to k_attention / k_attention_narrow (tests/test_llm_attention_gpu.py: the decode cases past 2 048 positions run both).
 * The session takes this form for passes of at most TK_LONG_ATT_MAX_ROWS rows that reach tk_long_att_min_pos(rows) (tk_llm_kernels.h).
 * ------------------------------------------------------------------------------------------ */
template <int HD>
__global__ __launch_bounds__(256) void k_att_scores_long(const float* __restrict__ partial, int ks, int n_total, const float* __restrict__ rope_cos,
                                                         const float* __restrict__ rope_sin, uint16_t* __restrict__ kcache, uint16_t* __restrict__ vcache,
                                                         const int32_t* __restrict__ seq, const int32_t* __restrict__ pos, int n_head, int n_kv_head, int layer,
                                                         int max_seq, int max_ctx, float* __restrict__ scores) {
    constexpr int KCH = HD / 2 + 16, PPR = HD / 8, PIECES = 64 * PPR, PPT = (PIECES + 255) / 256, HALF = HD / 2;
    __shared__ __attribute__((aligned(16))) uint8_t kbuf[64 * 4 * KCH];
    __shared__ float qs[4 * HD];      /* the group's rotated queries */
    __shared__ uint16_t own_k[HD];    /* this row's own key (the workgroup of the context's last block appends it, and the value) */
    const int kvh = blockIdx.x, b = blockIdx.y, r = blockIdx.z;
    const int p = pos[r], T = p + 1;
    if (64 * b >= T) return; /* the grid covers the whole window: blocks past this row's context leave at once */
    const int t = threadIdx.x, lane = t & 63, n16 = lane & 15, G = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int GQ = n_head / n_kv_head, QD = n_head * HD, KVD = n_kv_head * HD;
    const bool last = 64 * b + 64 >= T; /* this block holds position p */
    uint16_t* krun = kcache + (((int64_t)layer * max_seq + seq[r]) * n_kv_head + kvh) * (int64_t)max_ctx * HD;
    {   /* q (every block) and k, v of this row (the last block): K-split sums, RoPE on adjacent pairs, f16 rounding — k_qkv_rope_append's
         * arithmetic, inside this launch (a launch of its own cost 4.8 us per layer) */
        const float* cs = rope_cos + (int64_t)p * HALF;
        const float* sn = rope_sin + (int64_t)p * HALF;
        for (int idx = t; idx < (GQ + (last ? 2 : 0)) * HALF; idx += 256) {
            const int hsel = idx / HALF, i = idx % HALF;
            const int col = hsel < GQ ? (kvh * GQ + hsel) * HD + 2 * i : (hsel == GQ ? QD : QD + KVD) + kvh * HD + 2 * i;
            const float a = sum_partials(partial, ks, n_total, r, col), bb = sum_partials(partial, ks, n_total, r, col + 1);
            if (hsel < GQ) {
                qs[hsel * HD + 2 * i] = tk_fmaf(-bb, sn[i], a * cs[i]);
                qs[hsel * HD + 2 * i + 1] = tk_fmaf(a, sn[i], bb * cs[i]);
            } else if (hsel == GQ) {
                const uint16_t k0 = tk_f32_to_f16(tk_fmaf(-bb, sn[i], a * cs[i])), k1 = tk_f32_to_f16(tk_fmaf(a, sn[i], bb * cs[i]));
                own_k[2 * i] = k0; own_k[2 * i + 1] = k1;
                *(uint32_t*)(krun + (int64_t)p * HD + 2 * i) = (uint32_t)k0 | ((uint32_t)k1 << 16);
            } else {
                uint16_t* vrow = vcache + (((int64_t)layer * max_seq + seq[r]) * n_kv_head + kvh) * (int64_t)max_ctx * HD + (int64_t)p * HD;
                *(uint32_t*)(vrow + 2 * i) = (uint32_t)tk_f32_to_f16(a) | ((uint32_t)tk_f32_to_f16(bb) << 16);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) { /* the block's key rows into B-operand order (k_attention_prefill: put_k) */
        const int pid = t + u * 256;
        if (PIECES % 256 == 0 || pid < PIECES) {
            const int pn = pid / PPR, q8 = pid % PPR;
            int pr = 64 * b + pn;
            pr = pr < p ? pr : (p > 0 ? p - 1 : 0); /* cached rows only: position p itself is patched in from LDS below (its store may still be on its way) */
            const uint4 kk = *(const uint4*)(krun + (int64_t)pr * HD + 8 * q8);
            const uint32_t wv[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t lo = (wv[g >> 1] >> (16 * (g & 1))) & 0xffffu, hi = (wv[2 + (g >> 1)] >> (16 * (g & 1))) & 0xffffu;
                *(uint32_t*)(kbuf + (size_t)(pn * 4 + g) * KCH + 4 * q8) = lo | (hi << 16);
            }
        }
    }
    __syncthreads(); /* the staged block, qs and own_k are written */
    if (last && t < HD) { /* this row's own key into its slot of the block: dim t = 8 q8 + e sits at group e & 3, step 2 q8 + (e >> 2) */
        const int pn = p - 64 * b, q8 = t >> 3, e = t & 7;
        *(uint16_t*)(kbuf + (size_t)(pn * 4 + (e & 3)) * KCH + 2 * (2 * q8 + (e >> 2))) = own_k[t];
    }
    float qa[HD / 4]; /* A operand: tile row = query head of the group (rows GQ .. 15 are zero), lane group G holds dims 4 s + G */
#pragma unroll
    for (int s = 0; s < HD / 4; ++s) qa[s] = n16 < GQ ? qs[n16 * HD + 4 * s + G] : 0.0f;
    if (last) __syncthreads();
    v4f d = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint8_t* kp = kbuf + (size_t)((16 * w + n16) * 4 + G) * KCH;
#pragma unroll
    for (int q = 0; q < HD / 32; ++q) {
        const uint4 kk = *(const uint4*)(kp + 16 * q);
        const uint32_t wv[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[8 * q + e], f16bits_to_f32((wv[e >> 1] >> (16 * (e & 1))) & 0xffffu), d, 0, 0, 0);
    }
    const int tp = 64 * b + 16 * w + n16;
    if (G == 0 && tp < T) { /* lane group 0 holds tile rows 0 .. 3 = the group's heads */
        const float att_scale = tk_divf(1.0f, tk_sqrtf((float)HD));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < GQ) scores[((int64_t)r * n_head + kvh * GQ + i) * max_ctx + tp] = d[i] * att_scale;
    }
}

template <int HD>
__global__ __launch_bounds__(64) void k_att_pv_chain(const float* __restrict__ scores, const uint16_t* __restrict__ vcache, const int32_t* __restrict__ seq,
                                                     const int32_t* __restrict__ pos, int n_head, int n_kv_head, int layer, int max_seq, int max_ctx,
                                                     float* __restrict__ pv_part, float* __restrict__ l_part) {
    /* one wave = one (row, head, class) chain: 128 x rows workgroups, so the value rows arrive through as many CUs' memory pipes (a CU
     * pulls ~25 - 40 GB/s: sixteen workgroups of eight chains each — the first form of this launch — took 29 us at 2 048 positions) */
    const int j = blockIdx.x, head = blockIdx.y, r = blockIdx.z, lane = threadIdx.x;
    const int kvh = head / (n_head / n_kv_head);
    const int T = pos[r] + 1;
    const float* s = scores + ((int64_t)r * n_head + head) * max_ctx;
    const uint16_t* vrun = vcache + (((int64_t)layer * max_seq + seq[r]) * n_kv_head + kvh) * (int64_t)max_ctx * HD;
    float m = -INFINITY;
    for (int t0 = 0; t0 < T; t0 += 16 * 64) { /* sixteen loads in flight per lane: a load per iteration would be a chain of cache latencies */
        float sv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int tt = t0 + 64 * u + lane; sv[u] = tt < T ? s[tt] : -INFINITY; }
#pragma unroll
        for (int u = 0; u < 16; ++u) m = tk_fmaxf(m, sv[u]);
    }
    for (int sft = 32; sft >= 1; sft >>= 1) m = tk_fmaxf(m, wave_xor_f(m, sft));
    const int nj = T > j ? (T - j + 3) / 4 : 0; /* positions j, j + 4, ... of this class */
    /* The chain: a lane owns two dims (head_dim 64: the upper half-wave mirrors the lower).  Chunks of 32 class positions, two register
     * sets: a chunk's value rows are requested a chunk before the chain reaches them; the scores of a chunk are requested two chunks ahead
     * and BEFORE the value rows of the chunk in between (vmcnt counts in order: waiting for a score requested after 32 row loads would
     * drain them).  Probabilities reach every lane through LDS broadcast reads issued before the chunk's chain starts.  Measured and
     * dropped: two rows per request with v_permlane32_swap (64 positions in flight; 19.1 us against 15.8 at 2 048 positions: the swaps and
     * the four-dim owners' longer steps cost more than the deeper prefetch saves), v_readlane per step (16.7 us). */
    const int dl = lane % (HD / 2);
    float acc0 = 0.0f, acc1 = 0.0f, l = 0.0f;
    uint32_t va[32], vb[32];
    auto fetch = [&](uint32_t (&v)[32], int c) { /* value rows of class positions 32 c .. 32 c + 31 */
        if (32 * c + 32 <= nj) { /* a whole chunk: one base address, rows 4 * head_dim halves apart (two scalar adds per request instead of six) */
            const uint16_t* base = vrun + (int64_t)(j + 128 * c) * HD + 2 * dl;
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = *(const uint32_t*)(base + (int64_t)k * 4 * HD);
            return;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) { /* the last chunk, or past the end: clamped rows (their probability is 0) */
            int tt = j + 4 * (32 * c + k);
            tt = tt < T ? tt : T - 1;
            v[k] = *(const uint32_t*)(vrun + (int64_t)tt * HD + 2 * dl);
        }
    };
    auto score_of = [&](int c) -> float { /* lane i < 32: score of class position 32 c + i */
        const int i = 32 * c + (lane & 31);
        return i < nj ? s[j + 4 * i] : -INFINITY;
    };
    auto probs = [&](float sc, int c) -> float { return 32 * c + (lane & 31) < nj ? tk_expf(sc - m) : 0.0f; };
    __shared__ float ebuf[2][32];
    int eslot = 0;
    auto eat = [&](const uint32_t (&v)[32], float e32) {
        if (lane < 32) ebuf[eslot][lane] = e32;
        __syncthreads();
        float ek[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) ek[k] = ebuf[eslot][k];
        eslot ^= 1;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            acc0 = tk_fmaf(ek[k], f16bits_to_f32(v[k] & 0xffffu), acc0);
            acc1 = tk_fmaf(ek[k], f16bits_to_f32(v[k] >> 16), acc1);
            l = l + ek[k];
        }
    };
    const int nch = (nj + 31) / 32;
    float sa = score_of(0), sb = score_of(1);
    if (nch > 0) fetch(va, 0);
    for (int c = 0; c < nch; c += 2) {
        const float ea = probs(sa, c);
        sa = score_of(c + 2);
        fetch(vb, c + 1); /* past the end: clamped rows, probabilities 0 */
        eat(va, ea);
        if (c + 1 >= nch) break;
        const float eb = probs(sb, c + 1);
        sb = score_of(c + 3);
        if (c + 2 < nch) fetch(va, c + 2);
        eat(vb, eb);
    }
    float* pp = pv_part + (((int64_t)r * n_head + head) * TK_ATT_TSPLIT + j) * HD;
    if (lane < HD / 2) { pp[2 * dl] = acc0; pp[2 * dl + 1] = acc1; }
    if (lane == 0) l_part[((int64_t)r * n_head + head) * TK_ATT_TSPLIT + j] = l;
}

/* the four classes of every head joined in order, divided, quantised: workgroup = (row, the 256 / head_dim heads of a Q8 block) */
template <int HD>
__global__ __launch_bounds__(256) void k_att_pv_join(const float* __restrict__ pv_part, const float* __restrict__ l_part, int n_head, TkActQ8 out) {
    __shared__ float obuf[256];
    constexpr int HPB = 256 / HD;
    const int r = blockIdx.y, t = threadIdx.x;
    const int head = blockIdx.x * HPB + t / HD, dd = t % HD;
    const float* pp = pv_part + ((int64_t)r * n_head + head) * TK_ATT_TSPLIT * HD + dd;
    const float* lp = l_part + ((int64_t)r * n_head + head) * TK_ATT_TSPLIT;
    const float a = ((pp[0] + pp[HD]) + pp[2 * HD]) + pp[3 * HD];
    const float ll = ((lp[0] + lp[1]) + lp[2]) + lp[3];
    obuf[t] = tk_divf(a, ll);
    __syncthreads();
    if (t < 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = obuf[8 * t + i];
        quantize_chunk8(v, blockIdx.x * 32 + t, r, out);
    }
}

void tk_launch_attention_long(const float* partial, int ks, int n_total, const float* rope_cos, const float* rope_sin, uint16_t* kcache, uint16_t* vcache,
                              const int32_t* seq, const int32_t* pos, int nrows, int n_head, int n_kv_head, int head_dim, int layer, int max_seq, int max_ctx,
                              float* scores, TkActQ8 out, hipStream_t s) {
    const dim3 gs(n_kv_head, (max_ctx + 63) / 64, nrows);
    /* the scratch buffer: scores, then the classes' partial outputs and denominators (tk_attention_long_scratch_floats) */
    float* pv_part = scores + (size_t)TK_LONG_ATT_MAX_ROWS * n_head * max_ctx;
    float* l_part = pv_part + (size_t)TK_LONG_ATT_MAX_ROWS * n_head * TK_ATT_TSPLIT * head_dim;
    const dim3 gc(TK_ATT_TSPLIT, n_head, nrows);
    if (head_dim == 128) {
        hipLaunchKernelGGL((k_att_scores_long<128>), gs, dim3(256), 0, s, partial, ks, n_total, rope_cos, rope_sin, kcache, vcache, seq, pos, n_head, n_kv_head, layer,
                           max_seq, max_ctx, scores);
        hipLaunchKernelGGL((k_att_pv_chain<128>), gc, dim3(64), 0, s, scores, vcache, seq, pos, n_head, n_kv_head, layer, max_seq, max_ctx, pv_part, l_part);
        hipLaunchKernelGGL((k_att_pv_join<128>), dim3(n_head / 2, nrows), dim3(256), 0, s, pv_part, l_part, n_head, out);
    } else {
        hipLaunchKernelGGL((k_att_scores_long<64>), gs, dim3(256), 0, s, partial, ks, n_total, rope_cos, rope_sin, kcache, vcache, seq, pos, n_head, n_kv_head, layer,
                           max_seq, max_ctx, scores);
        hipLaunchKernelGGL((k_att_pv_chain<64>), gc, dim3(64), 0, s, scores, vcache, seq, pos, n_head, n_kv_head, layer, max_seq, max_ctx, pv_part, l_part);
        hipLaunchKernelGGL((k_att_pv_join<64>), dim3(n_head / 4, nrows), dim3(256), 0, s, pv_part, l_part, n_head, out);
    }
}

size_t tk_attention_long_scratch_floats(int n_head, int head_dim, int max_ctx) {
    return (size_t)TK_LONG_ATT_MAX_ROWS * n_head * ((size_t)max_ctx + (size_t)TK_ATT_TSPLIT * head_dim + TK_ATT_TSPLIT);
}

bool tk_attention_long_applies(int nrows, int n_head, int n_kv_head, int head_dim) {
    /* TK_MI355X_NO_LONG_ATT=1: decode passes keep the fused kernels whatever the context (A/B timing, parity tests of both) */
    const char* np = getenv("TK_MI355X_NO_LONG_ATT");
    if (np && np[0] == '1') return false;
    if (nrows < 1 || nrows > TK_LONG_ATT_MAX_ROWS || (head_dim != 64 && head_dim != 128)) return false;
    const int hpb = 256 / head_dim, grp = n_kv_head > 0 ? n_head / n_kv_head : 0;
    return grp >= 1 && grp <= 4 && grp % hpb == 0 && n_head % hpb == 0;
}

bool tk_attention_prefill_applies(int n_head, int n_kv_head, int head_dim) {
    /* TK_MI355X_NO_PREFILL_ATT=1: multi-position passes keep k_attention's one-workgroup-per-row form (A/B timing, parity tests of both) */
    const char* np = getenv("TK_MI355X_NO_PREFILL_ATT");
    if (np && np[0] == '1') return false;
    if (head_dim != 64 && head_dim != 128) return false;
    const int hpb = 256 / head_dim, grp = n_kv_head > 0 ? n_head / n_kv_head : 0;
    return grp > 0 && grp % hpb == 0 && n_head % hpb == 0;
}

void tk_launch_att_tiles(const int32_t* seq, int nrows, int32_t* tiles, hipStream_t s) {
    hipLaunchKernelGGL(k_att_tiles, dim3(1), dim3(TK_MAX_ROWS), 0, s, seq, nrows, tiles);
}

void tk_launch_attention_prefill(const float* qbuf, const uint16_t* kcache, const uint16_t* vcache, const int32_t* seq, const int32_t* pos,
                                 const int32_t* tiles, int nrows, int n_head, int n_kv_head, int head_dim, int layer, int max_seq, int max_ctx, TkActQ8 out,
                                 hipStream_t s) {
    if (head_dim == 128)
        hipLaunchKernelGGL((k_attention_prefill<128>), dim3(n_head / 2, nrows / 16 + 8), dim3(512), TkPrefillLds<128>::bytes, s, qbuf, kcache, vcache, seq, pos, tiles, n_head,
                           n_kv_head, layer, max_seq, max_ctx, out);
    else
        hipLaunchKernelGGL((k_attention_prefill<64>), dim3(n_head / 4, nrows / 16 + 8), dim3(1024), TkPrefillLds<64>::bytes, s, qbuf, kcache, vcache, seq, pos, tiles, n_head,
                           n_kv_head, layer, max_seq, max_ctx, out);
}

static size_t tk_attention_narrow_fixed_lds(int max_ctx) { /* everything but the key / value rows: q, scores, maxima, own row */
    return (size_t)(2 * 128 + 2 * (size_t)((max_ctx + 3) & ~3) + 32) * sizeof(float) + 2 * 128 * 2;
}
/* decode sessions alive per device: a narrow-attention workgroup takes a whole CU (16 waves, most of its LDS), so while it runs no other
 * stream's mat-vec workgroup can share the CU and the launch itself waits until whole CUs drain.  Alone that is the fastest form; with
 * several decode streams on the device the ring form (4 waves, <= 70 KiB) overlaps with their launches and the step is shorter
 * (3 groups x 16 rows: 3.92 -> 3.75 ms per step and group, profiles/r06_northstar_explore.txt).  Both forms are bit-identical. */
static std::atomic<int> g_sessions_alive[64];
void tk_attention_note_session(int device, int delta) {
    if (device >= 0 && device < 64) g_sessions_alive[device].fetch_add(delta, std::memory_order_relaxed);
}
static bool tk_device_is_shared_by_sessions() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    return g_sessions_alive[dev].load(std::memory_order_relaxed) > 1;
}
/* positions the narrow kernel keeps resident per chunk (a multiple of 32), 0 when it does not apply */
static int tk_attention_narrow_cap(int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, bool fused) {
    const int grp = n_kv_head > 0 ? n_head / n_kv_head : 0;
    if (!fused || head_dim != 128 || (grp != 2 && grp != 4) || (n_head / 2) * nrows > TK_NUM_CU) return 0;
    /* TK_MI355X_NO_NARROW_ATT=1: narrow passes keep k_attention's ring form (A/B timing: a narrow workgroup takes a whole CU — 16 waves and
     * most of its LDS — so it cannot share the CU with another stream's mat-vec workgroups) */
    const char* nn = getenv("TK_MI355X_NO_NARROW_ATT"); /* =0: narrow even when the device is shared */
    if (nn && nn[0] == '1') return 0;
    if (!(nn && nn[0] == '0') && tk_device_is_shared_by_sessions()) return 0;
    const size_t fixed = tk_attention_narrow_fixed_lds(max_ctx);
    if (fixed + 64 * 512 > (size_t)TK_MAX_DYN_LDS) return 0;
    int cap = (int)(((size_t)TK_MAX_DYN_LDS - fixed) / 512);
    const int want = (max_ctx + 31) / 32 * 32;
    cap = cap > want ? want : cap;
    return cap / 32 * 32;
}

size_t tk_attention_lds_bytes(int gq, int head_dim, int max_ctx, int chunk, int slots) {
    const size_t W = (size_t)gq * head_dim;
    const size_t ring = (size_t)slots * chunk * head_dim * 2;
    const size_t epilogue = ((size_t)TK_ATT_TSPLIT * W + TK_ATT_TSPLIT * TK_ATT_MAX_GRP + W) * sizeof(float); /* aliases the ring */
    return (ring > epilogue ? ring : epilogue) + (W + (size_t)gq * max_ctx + 4 * TK_ATT_MAX_GRP) * sizeof(float) + (size_t)2 * head_dim * 2;
}

#ifndef TK_ATT_NARROW_CHUNK
#define TK_ATT_NARROW_CHUNK 128 /* 64: the round-3 form */
#endif
#ifndef TK_ATT_WIDE_SLOTS
#define TK_ATT_WIDE_SLOTS 2 /* ring depth of the many-workgroup (32-position chunk) form */
#endif
/* Which attention launch a pass takes on the calling thread's device (the choice depends on its CU count): the one place that decides,
 * used by the launcher and exported (tk_mi355x_attention_plan) so tests assert what they run from the launcher's own answer. */
TkAttentionPlan tk_attention_plan(int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, bool fused) {
    TkAttentionPlan pl{};
#ifndef TK_ATT_NO_NARROW
    if (const int cap = tk_attention_narrow_cap(nrows, n_head, n_kv_head, head_dim, max_ctx, fused)) {
        pl.kernel = 1; pl.gq = 2; pl.chunk = cap; pl.slots = 1;
        pl.lds_bytes = tk_attention_narrow_fixed_lds(max_ctx) + (size_t)cap * 512;
        return pl;
    }
#endif
    int gq = n_head / n_kv_head; /* 1, 2 or 4 (TkLlmModel::init) */
    /* few rows: two workgroups per KV head (two query heads each) — twice the workgroups, half the dependent work in each; K / V are then
     * read twice, which costs nothing while the launch is latency-bound (16 rows: 15.9 -> see profiles/r02_attention_variants.txt) */
    if (gq == 4 && (2 * head_dim) % 256 == 0 && nrows * n_kv_head < 2 * TK_NUM_CU) gq = 2;
#ifdef TK_ATT_GQ1_ROWS /* diagnostic: one query head per workgroup below this many rows */
    if (gq == 2 && head_dim % 256 == 0 && nrows <= TK_ATT_GQ1_ROWS) gq = 1;
#endif
    /* positions per ring slot: 64 while the launch is a latency chain of few workgroups (fewer, longer phases), 32 once several
     * workgroups per CU are resident (16 KiB less LDS each: more of them fit; 256 rows: 47.9 -> 43.3 us, 16 rows would lose 24 %;
     * profiles/r02_attention_variants.txt) */
    /* at most one workgroup per CU and <= 2 query heads per workgroup: 128 positions per slot — the score phase (one fma chain per (head,
     * position): 2 x 64 threads of a 64-position chunk leave two of the four waves idle) runs on all four waves and a context has half as
     * many barrier-separated phases */
    const bool narrow = TK_ATT_NARROW_CHUNK == 128 && gq * 128 <= 256 && (n_head / gq) * nrows <= TK_NUM_CU && (head_dim == 128 || head_dim == 64) &&
                        tk_attention_lds_bytes(gq, head_dim, max_ctx, 128, 2) <= (size_t)TK_MAX_DYN_LDS;
    const int chunk = narrow ? 128 : (n_head / gq) * nrows > 4 * TK_NUM_CU ? 32 : 64;
    /* ring depth 2 at every pass width: with few workgroups a five-slot ring (every K and V chunk of a 128-position context in flight at
     * once) measured no faster — 16 rows x 128 positions 10.7 us against 11.1, decode step unchanged (profiles/r03_attention_slots.txt): a
     * one-row launch already takes 9.8 us, the launch is a chain of ~10 barrier-separated phases, not of DMA latencies */
    pl.kernel = 0; pl.gq = gq; pl.chunk = chunk; pl.slots = chunk == 32 ? TK_ATT_WIDE_SLOTS : 2;
    pl.lds_bytes = tk_attention_lds_bytes(gq, head_dim, max_ctx, chunk, pl.slots);
    /* a pass that is not fused holds several positions of a sequence: the session runs k_attention_prefill (kernel 2) where it applies; the
     * other fields keep describing the k_attention form it replaces (TK_MI355X_NO_PREFILL_ATT=1, or a caller of tk_launch_attention itself) */
    if (!fused && tk_attention_prefill_applies(n_head, n_kv_head, head_dim)) pl.kernel = 2;
    return pl;
}

void tk_launch_attention(const float* qbuf, const float* partial, int ks, int n_total, const float* rope_cos, const float* rope_sin,
                         uint16_t* kcache, uint16_t* vcache, const int32_t* seq, const int32_t* pos, int nrows, int n_head, int n_kv_head,
                         int head_dim, int layer, int max_seq, int max_ctx, TkActQ8 out, bool fused, hipStream_t s) {
    const TkAttentionPlan pl = tk_attention_plan(nrows, n_head, n_kv_head, head_dim, max_ctx, fused);
    if (pl.kernel == 1) {
        hipLaunchKernelGGL(k_attention_narrow, dim3(n_head / 2, nrows), dim3(1024), pl.lds_bytes, s, partial, ks, n_total, rope_cos, rope_sin, kcache,
                           vcache, seq, pos, n_head, n_kv_head, layer, max_seq, max_ctx, pl.chunk, out);
        return;
    }
    const int gq = pl.gq, chunk = pl.chunk;
    const size_t lds = pl.lds_bytes;
#define TK_ATT_LAUNCH_CH(G, F, H, C, S)                                                                                                       \
    hipLaunchKernelGGL((k_attention<G, F, H, C, S>), dim3(n_head / gq, nrows), dim3(256), lds, s, qbuf, partial, ks, n_total, rope_cos, rope_sin, kcache, \
                       vcache, seq, pos, n_head, n_kv_head, head_dim, layer, max_seq, max_ctx, out)
#define TK_ATT_LAUNCH_HD(G, F, H) do { if (chunk == 32) TK_ATT_LAUNCH_CH(G, F, H, 32, TK_ATT_WIDE_SLOTS); else if (chunk == 128 && G <= 2 && H != 0) TK_ATT_LAUNCH_CH((G <= 2 ? G : 2), F, (H ? H : 128), 128, 2); else TK_ATT_LAUNCH_CH(G, F, H, 64, 2); } while (0)
#define TK_ATT_LAUNCH(G, F) do { if (head_dim == 128) TK_ATT_LAUNCH_HD(G, F, 128); else if (head_dim == 64) TK_ATT_LAUNCH_HD(G, F, 64); else TK_ATT_LAUNCH_HD(G, F, 0); } while (0)
    if (fused) {
        if (gq == 4) TK_ATT_LAUNCH(4, true); else if (gq == 2) TK_ATT_LAUNCH(2, true); else TK_ATT_LAUNCH(1, true);
    } else {
        if (gq == 4) TK_ATT_LAUNCH(4, false); else if (gq == 2) TK_ATT_LAUNCH(2, false); else TK_ATT_LAUNCH(1, false);
    }
#undef TK_ATT_LAUNCH_CH
#undef TK_ATT_LAUNCH_HD
#undef TK_ATT_LAUNCH
}

/* ------------------------------------------------------------------------------------------
 * SwiGLU + Q8 quantise:  act = silu(gate) * up ; gate = cols [0,FF), up = cols [FF,2FF)
 * ------------------------------------------------------------------------------------------ */
/* KSM = the most K-split slabs an instantiation handles: the slab loads are held in registers before the first add, and the gate | up
 * projection of the canonical plan has ONE slab — with room for eight the kernel needs 130 registers (3 waves per SIMD), with one 40 */
template <int KSM>
__global__ __launch_bounds__(256) void k_swiglu_q8(const float* partial, int ks, int FF, TkActQ8 out) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= FF / 8) return; /* FF/8 is a multiple of 32: whole half-waves drop out together */
    /* 8 gate and 8 up values as 16-byte loads, every K-split slab requested before the first add (slabs added in ascending order) */
    v4f g[2], u[2];
    {
        v4f pg[KSM][2], pu[KSM][2];
#pragma unroll
        for (int s = 0; s < KSM; ++s)
            if (s < ks) {
                const float* row = partial + ((int64_t)s * TK_MAX_ROWS + r) * (2 * (int64_t)FF);
                pg[s][0] = *(const v4f*)(row + 8 * c); pg[s][1] = *(const v4f*)(row + 8 * c + 4);
                pu[s][0] = *(const v4f*)(row + FF + 8 * c); pu[s][1] = *(const v4f*)(row + FF + 8 * c + 4);
            }
        g[0] = pg[0][0]; g[1] = pg[0][1]; u[0] = pu[0][0]; u[1] = pu[0][1];
#pragma unroll
        for (int s = 1; s < KSM; ++s)
            if (s < ks) { g[0] = g[0] + pg[s][0]; g[1] = g[1] + pg[s][1]; u[0] = u[0] + pu[s][0]; u[1] = u[1] + pu[s][1]; }
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tk_siluf(g[i >> 2][i & 3]) * u[i >> 2][i & 3];
    quantize_chunk8(v, c, r, out);
}

/* the Q8 quantisation alone, for h = silu(gate) * up that the wide gate | up launch has formed in its epilogue (TkGemvArgs::swiglu) */
__global__ __launch_bounds__(256) void k_quant_q8(const float* __restrict__ hbuf, int FF, TkActQ8 out) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= FF / 8) return; /* FF/8 is a multiple of 32: whole half-waves drop out together */
    const float* row = hbuf + (int64_t)r * FF + 8 * c;
    const v4f a = *(const v4f*)row, b = *(const v4f*)(row + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    quantize_chunk8(v, c, r, out);
}

void tk_launch_quant_q8(const float* hbuf, int FF, int nrows, TkActQ8 out, hipStream_t s) {
    hipLaunchKernelGGL(k_quant_q8, dim3((FF / 8 + 255) / 256, nrows), dim3(256), 0, s, hbuf, FF, out);
}

bool tk_gemv_fuses_swiglu(int nrows, int ks, int type_gate, int type_up) {
    return nrows >= tk_g32_from() && ks == 1 && type_gate == type_up && (type_gate == TK_TYPE_Q4_K || type_gate == TK_TYPE_Q6_K); /* the 32x32x32 kernel's epilogue */
}

void tk_launch_swiglu_q8(const float* partial, int ks, int FF, int nrows, TkActQ8 out, hipStream_t s) {
    if (ks == 1) hipLaunchKernelGGL((k_swiglu_q8<1>), dim3((FF / 8 + 255) / 256, nrows), dim3(256), 0, s, partial, ks, FF, out);
    else hipLaunchKernelGGL((k_swiglu_q8<TK_RMS_MAX_KS>), dim3((FF / 8 + 255) / 256, nrows), dim3(256), 0, s, partial, ks, FF, out);
}

/* ------------------------------------------------------------------------------------------
 * sampling.  Greedy (the default, the parity definition of SURVEY §0 F8 and the bench's mode): first index of the maximum.  Stochastic
 * (a row whose TkSampleRow::temp > 0): the chain the reference installs with llama_sampling_default_params()
 * (src/ai_models/tk_runner_lifecycle.c:76-77, sampled in tk_runner_streaming.c:60-61) — top-k, top-p, min-p, temperature, one draw —
 * restated with ONE canonical arithmetic order (oracle: orc_sample_row) so a seed gives the same ids on both sides:
 *   candidates = the K = min(top_k or 64, 64, allowed) tokens of largest logit, ties to the lower id, in that order (l_0 >= l_1 >= ...)
 *   p_i = exp(l_i - l_0) / sum (sum over i ascending);  top-p: the shortest prefix whose running sum reaches top_p;  min-p: drop the tail
 *   with p_i < min_p * p_0;  w_i = exp((l_i - l_0) / temp), W = sum ascending;  u = 24 random bits * 2^-24 from splitmix64(seed, counter);
 *   the first i whose running sum of w exceeds u * W.
 * Either way the result feeds the next step on-device (tok <- id, pos <- pos + 1, history append, counter + 1) so a decode loop is a pure
 * graph replay.  The selection is a radix select over (order-preserving logit key, ~id): six 8-bit rounds of an LDS histogram.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(1024) void k_argmax(const float* logits, int vocab, const uint32_t* allow_base, const int32_t* allow_row, TkSampleRow* samp,
                                                  int32_t* tok, int32_t* pos, int32_t* nsteps, int32_t* hist, int hist_stride) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ uint32_t sm[256 + 8 + 4 * TK_SAMPLE_MAX_K];
    const int r = blockIdx.x, t = threadIdx.x;
    const float* lg = logits + (int64_t)r * vocab;
    /* per-row token masks: allow_row[r] = index of this row's mask ((vocab + 31) / 32 words each) or -1 = unconstrained, so rows sampling
     * under different grammars — and rows under none — share one pass */
    const int mi = allow_row ? allow_row[r] : -1;
    const uint32_t* allow = mi >= 0 ? allow_base + (size_t)mi * ((vocab + 31) / 32) : nullptr;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    if (samp && samp[r].temp > 0.0f) { /* wave-uniform per workgroup: a row is sampled by its whole workgroup */
        __shared__ int32_t picked;
        const TkSampleRow sp = samp[r];
        sample_row(lg, vocab, allow, sp, &picked, sm);
        __syncthreads();
        if (t == 0) { idx = picked; samp[r].counter = sp.counter + 1; }
    } else {
    /* a thread's candidates i = t, t + 1024, ... in ascending order, eight loads in flight at a time: a load under a test (the mask's
     * `continue`) waits for the one before it — 32 latencies in a row were 14 us of a one-row decode step.  Indices past the vocabulary
     * load a clamped element and are skipped by the test. */
    constexpr int NU = 8;
    if (allow) { /* grammar-constrained sampling: arg max over the allowed tokens */
        for (int i0 = t; i0 < vocab; i0 += 1024 * NU) {
            float v[NU];
            uint32_t m[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = i0 + 1024 * u, ic = i < vocab ? i : vocab - 1;
                v[u] = lg[ic];
                m[u] = allow[ic >> 5];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = i0 + 1024 * u;
                if (i < vocab && ((m[u] >> (i & 31)) & 1u) && (v[u] > best || idx == 0x7fffffff)) { best = v[u]; idx = i; } /* the first allowed token wins ties and -inf logits */
            }
        }
    } else {
        for (int i0 = t; i0 < vocab; i0 += 1024 * NU) {
            float v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = i0 + 1024 * u;
                v[u] = lg[i < vocab ? i : vocab - 1];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = i0 + 1024 * u;
                if (i < vocab && (v[u] > best || idx == 0x7fffffff)) { best = v[u]; idx = i; }
            }
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s, TK_WAVE);
        const int oi = __shfl_xor(idx, s, TK_WAVE);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    }
    }
    if (t == 0) {
        if (tok) tok[r] = idx;
        if (pos) pos[r] = pos[r] + 1;
        if (hist) {
            const int n = nsteps[r];
            hist[(int64_t)n * hist_stride + r] = idx;
            nsteps[r] = n + 1;
        }
    }
}

void tk_launch_argmax(const float* logits, int vocab, int nrows, const uint32_t* allow_base, const int32_t* allow_row, TkSampleRow* samp, int32_t* tok,
                      int32_t* pos, int32_t* nsteps, int32_t* hist, int hist_stride, hipStream_t s) {
    hipLaunchKernelGGL(k_argmax, dim3(nrows), dim3(1024), 0, s, logits, vocab, allow_base, allow_row, samp, tok, pos, nsteps, hist, hist_stride);
}

/* ------------------------------------------------------------------------------------------
 * per-device opt-in to > 64 KiB of dynamic LDS for every kernel above (see the comment at TK_MAX_DYN_LDS)
 * ------------------------------------------------------------------------------------------ */
const char* tk_llm_prepare_device(int device) {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return "device ordinal out of range";
    if (done[device]) return nullptr;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) return "tk_llm_prepare_device: the calling thread's current device must be `device`";
    hipError_t e = hipSuccess;
#define TK_OPT(fn) do { if (e == hipSuccess) e = opt_in_lds(fn); } while (0)
#define TK_OPT_GEMM(MTV) do { TK_OPT((k_gemm_w4a8<MTV, 1, 1>)); TK_OPT((k_gemm_w4a8<MTV, 2, 1>)); TK_OPT((k_gemm_w4a8<MTV, 3, 1>)); } while (0)
    TK_OPT_GEMM(4); TK_OPT_GEMM(6); TK_OPT_GEMM(8); TK_OPT_GEMM(10); TK_OPT_GEMM(12); TK_OPT_GEMM(14); TK_OPT_GEMM(16);
#define TK_OPT_G32(TYV) do { TK_OPT((k_gemm32_w4a8<TYV, 4>)); TK_OPT((k_gemm32_w4a8<TYV, 3>)); } while (0)
    TK_OPT_G32(1); TK_OPT_G32(2); TK_OPT_G32(3);
#undef TK_OPT_G32
#define TK_OPT_GEMV_F(PFV, MTV, FUV) do { TK_OPT((k_gemv_w4a8<PFV, MTV, 1, FUV>)); TK_OPT((k_gemv_w4a8<PFV, MTV, 2, FUV>)); TK_OPT((k_gemv_w4a8<PFV, MTV, 3, FUV>)); } while (0)
#define TK_OPT_GEMV(PFV, MTV) TK_OPT_GEMV_F(PFV, MTV, 0)
    TK_OPT_GEMV(1, 1); TK_OPT_GEMV(2, 1); TK_OPT_GEMV(1, 2); TK_OPT_GEMV(2, 2); TK_OPT_GEMV_F(2, 1, 1); TK_OPT_GEMV_F(2, 1, 2);
#define TK_OPT_ATT_C(H, C, S) do { TK_OPT((k_attention<1, true, H, C, S>)); TK_OPT((k_attention<2, true, H, C, S>)); TK_OPT((k_attention<4, true, H, C, S>)); \
                                 TK_OPT((k_attention<1, false, H, C, S>)); TK_OPT((k_attention<2, false, H, C, S>)); TK_OPT((k_attention<4, false, H, C, S>)); } while (0)
#define TK_OPT_ATT(H) do { TK_OPT_ATT_C(H, 32, TK_ATT_WIDE_SLOTS); TK_OPT_ATT_C(H, 64, 2); } while (0)
    TK_OPT_ATT(0); TK_OPT_ATT(64); TK_OPT_ATT(128);
#define TK_OPT_ATT_N(H) do { TK_OPT((k_attention<1, true, H, 128, 2>)); TK_OPT((k_attention<2, true, H, 128, 2>)); TK_OPT((k_attention<1, false, H, 128, 2>)); TK_OPT((k_attention<2, false, H, 128, 2>)); } while (0)
    TK_OPT_ATT_N(64); TK_OPT_ATT_N(128);
    TK_OPT(k_attention_narrow);
    TK_OPT((k_attention_prefill<128>));
    TK_OPT((k_attention_prefill<64>));
#undef TK_OPT_ATT_N
#undef TK_OPT_ATT
#undef TK_OPT_GEMV
#undef TK_OPT_GEMM
#undef TK_OPT
    if (e != hipSuccess) return hipGetErrorString(e);
    done[device] = true;
    return nullptr;
}
