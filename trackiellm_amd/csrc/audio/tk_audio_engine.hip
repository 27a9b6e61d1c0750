#include "tk_audio_engine.h"
#include "tk_whisper_ggml.h"

#include <stdio.h>
#include <string.h>

#include "../nn/tk_gemm_tiled.h"
#include "../nn/tk_nn_kernels.h"

/* Perception streams run at the highest stream priority: their kernels are small and many, and behind the LLM's large GEMM
 * launches they would otherwise queue for a free CU at every step (the fused cycle's critical path becomes queueing, not work). */
static inline hipError_t tk_create_perception_stream(hipStream_t* s) {
    int lo = 0, hi = 0; /* numerically lower = higher priority */
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

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

/* ---------------------------------------------------------------- log-mel front end kernels */

/* frame t, tap n: s16 -> f32 (/32768, as the reference's VAD does, src/sensors/tk_vad_silero.c:78-82), reflect padding of
 * n_fft/2 at both ends of the zero-padded 30 s window, times the Hann window */
__global__ void k_frames(const int16_t* pcm, int n_samples, int pcm_stride, int n_total, int T, const float* window, float* out) {
    const int b = blockIdx.z;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * TK_WH_NFFT) return;
    const int t = (int)(i / TK_WH_NFFT), n = (int)(i % TK_WH_NFFT);
    int j = t * TK_WH_HOP + n - TK_WH_NFFT / 2;
    if (j < 0) j = -j;
    if (j >= n_total) j = 2 * (n_total - 1) - j;
    const float x = j < n_samples ? tk_divf((float)pcm[(int64_t)b * pcm_stride + j], 32768.0f) : 0.0f;
    out[((int64_t)b * T + t) * TK_WH_NFFT + n] = x * window[n];
}

__global__ void k_power(const float* ri, int64_t rows, int nb, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nb) return;
    const int64_t r = i / nb;
    const int f = (int)(i % nb);
    const float re = ri[r * 2 * nb + f], im = ri[r * 2 * nb + nb + f];
    out[i] = tk_fmaf(re, re, im * im);
}

/* log10(max(x, 1e-10)); clamp to (max over the utterance) - 8; (x + 4) / 4.  One workgroup per batch element. */
__global__ __launch_bounds__(1024) void k_logmel_finish(float* mel, int per_b) {
    __shared__ float red[16];
    float* m = mel + (int64_t)blockIdx.x * per_b;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < per_b; i += 1024) {
        const float v = tk_log10f(tk_fmaxf(m[i], 1e-10f));
        m[i] = v;
        mx = tk_fmaxf(mx, v);
    }
    for (int s = 32; s >= 1; s >>= 1) mx = tk_fmaxf(mx, __shfl_xor(mx, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; ++w) mx = tk_fmaxf(mx, red[w]);
    const float floor_v = mx - 8.0f;
    for (int i = threadIdx.x; i < per_b; i += 1024) m[i] = (tk_fmaxf(m[i], floor_v) + 4.0f) * 0.25f;
}


/* ---------------------------------------------------------------- decoder-step attention, fused

 * One query row per (sequence, head): scores, softmax and the value contraction in one workgroup, with exactly the arithmetic of the
 * three-launch form it replaces (tk_whisper_graph.h: attention()): score = (fp32 fma chain over the head dimension, ascending) * scale
 * + 0.0f (the GEMM epilogue's bias slot); softmax as k_softmax_rows evaluates it (256 strided partial maxima / sums, shuffle tree,
 * ((w0 + w1) + w2) + w3); output = one fp32 fma chain over the keys in ascending order, + 0.0f.  The decoder steps of 32 utterances were
 * sixteen 64x64-tile GEMM launches of one live row each per step; this is eight short launches. */
#define TK_AT1_CK 64 /* keys per ring slot of the value stream */
#define TK_AT1_NS 3  /* ring slots */
/* s_waitcnt vmcnt(n) alone */
__device__ __forceinline__ void at1_wait_vmcnt(int n) {
#define TK_VMW(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
    switch (n) {
        TK_VMW(0) TK_VMW(1) TK_VMW(2) TK_VMW(3) TK_VMW(4) TK_VMW(5) TK_VMW(6) TK_VMW(7) TK_VMW(8) TK_VMW(9) TK_VMW(10) TK_VMW(11) TK_VMW(12)
        TK_VMW(13) TK_VMW(14) TK_VMW(15) TK_VMW(16)
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
#undef TK_VMW
}
/* The value contraction is ONE fp32 chain over the keys per output (the canonical order), carried by hd threads; what the other threads
 * can do is move the values: all four waves stream V through a three-slot LDS ring by LDS-DMA (64 keys x hd floats per slot, 1 KiB pieces
 * of four 256-byte rows), the first two slots requested before the scores are even computed, so the chain reads LDS instead of waiting
 * for 32 global rows at a time (cross-attention over 1500 keys: the chain wave alone had 384 KB to pull through 32 loads in flight). */
__global__ __launch_bounds__(256) void k_attend1(const float* q, const float* k, const float* v, float* out, int Tk, int64_t q_bstride, int64_t kv_bstride, int d,
                                                 int hd, float scale, float* img /* optional: out also as the tiled GEMM's operand image [B rows][K = d] */) {
    extern __shared__ __attribute__((aligned(16))) float att1_lds[];
    __shared__ float red[4];
    float* vring = att1_lds;                                   /* [NS][CK][hd] */
    float* sc = att1_lds + (size_t)TK_AT1_NS * TK_AT1_CK * hd; /* [Tk] */
    float* qs = sc + Tk;                                       /* [hd] */
    const int h = blockIdx.x, b = blockIdx.y, t0 = threadIdx.x, lane = t0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
    const float* qp = q + (int64_t)b * q_bstride + (int64_t)h * hd;
    const float* kp = k + (int64_t)b * kv_bstride + (int64_t)h * hd;
    const float* vp = v + (int64_t)b * kv_bstride + (int64_t)h * hd;
    const int nchunk = (Tk + TK_AT1_CK - 1) / TK_AT1_CK;
    const int upr = hd / 4;                       /* 16-byte units per row */
    const int ppw = TK_AT1_CK * hd / 256 / 4;     /* DMA pieces per wave and chunk (the launcher admits hd % 16 == 0 only) */
    auto issue = [&](int c) { /* rows past the last key re-read it: never used by the chain */
        float* slot = vring + (size_t)(c % TK_AT1_NS) * TK_AT1_CK * hd;
        for (int i = 0; i < ppw; ++i) {
            const int p = wave + 4 * i, u = p * 64 + lane;
            int row = c * TK_AT1_CK + u / upr;
            row = row < Tk ? row : Tk - 1;
            const auto gs = (const __attribute__((address_space(1))) void*)(vp + (int64_t)row * d + 4 * (u % upr));
            const auto ls = (__attribute__((address_space(3))) void*)((uint8_t*)slot + p * 1024);
            __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
        }
    };
    for (int j = 0; j < TK_AT1_NS - 1 && j < nchunk; ++j) issue(j);
    for (int i = t0; i < hd; i += 256) qs[i] = qp[i];
    __syncthreads();
    float m = -INFINITY;
    for (int t = t0; t < Tk; t += 256) {
        const float* kr = kp + (int64_t)t * d;
        float a = 0.0f;
        for (int i = 0; i < hd; i += 4) {
            const float4 kk = *(const float4*)(kr + i);
            a = tk_fmaf(qs[i], kk.x, a); a = tk_fmaf(qs[i + 1], kk.y, a); a = tk_fmaf(qs[i + 2], kk.z, a); a = tk_fmaf(qs[i + 3], kk.w, a);
        }
        a = a * scale;
        a = a + 0.0f;
        sc[t] = a;
        m = tk_fmaxf(m, a);
    }
    for (int s = 32; s >= 1; s >>= 1) m = tk_fmaxf(m, __shfl_xor(m, s, 64));
    if ((t0 & 63) == 0) red[t0 >> 6] = m;
    __syncthreads();
    m = tk_fmaxf(tk_fmaxf(red[0], red[1]), tk_fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int t = t0; t < Tk; t += 256) { const float e = tk_expf(sc[t] - m); sc[t] = e; s = s + e; }
    for (int w = 32; w >= 1; w >>= 1) s = s + __shfl_xor(s, w, 64);
    if ((t0 & 63) == 0) red[t0 >> 6] = s;
    __syncthreads();
    const float tot = ((red[0] + red[1]) + red[2]) + red[3];
    for (int t = t0; t < Tk; t += 256) sc[t] = tk_divf(sc[t], tot);
    float a = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
        /* chunk c has landed once all but the chunks requested after it have; the barrier also says everybody is done with chunk c - 1,
         * whose slot takes chunk c + NS - 1 (and, the first time, publishes the probabilities) */
        const int ahead = nchunk - 1 - c < TK_AT1_NS - 2 ? nchunk - 1 - c : TK_AT1_NS - 2;
        at1_wait_vmcnt(ahead * ppw);
        __syncthreads();
        if (c + TK_AT1_NS - 1 < nchunk) issue(c + TK_AT1_NS - 1);
        if (t0 < hd) {
            const float* vs = vring + (size_t)(c % TK_AT1_NS) * TK_AT1_CK * hd + t0;
            const float* pc = sc + c * TK_AT1_CK;
            const int n = Tk - c * TK_AT1_CK < TK_AT1_CK ? Tk - c * TK_AT1_CK : TK_AT1_CK;
            int u = 0;
            for (; u + 8 <= n; u += 8) {
                float vv[8], pp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { vv[e] = vs[(u + e) * hd]; pp[e] = pc[u + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) a = tk_fmaf(pp[e], vv[e], a);
            }
            for (; u < n; ++u) a = tk_fmaf(pc[u], vs[u * hd], a);
        }
    }
    if (t0 < hd) {
        const float r = a + 0.0f;
        out[(int64_t)b * q_bstride + (int64_t)h * hd + t0] = r;
        if (img) {
            const int kk = h * hd + t0;
            img[(int64_t)(b >> 4) * 16 * d + (int64_t)(kk >> 4) * 256 + (kk & 3) * 64 + (b & 15) * 4 + ((kk & 15) >> 2)] = r;
        }
    }
}

/* ---------------------------------------------------------------- GPU ops for the shared graph */

/* shapes the tiled GEMM takes: a plain linear layer C = act(A W^T + b) + R with K a multiple of 128 */
static inline bool tk_tiled_gemm_applies(const TkGemm& g) {
    return g.b_kn == 0 && g.batch <= 1 && g.batch_inner == 0 && g.alpha == 1.0f && !g.b_f16 && g.ldb == g.K && g.K % 128 == 0 && g.M > 0;
}
/* floats of the activation image: whole row blocks (16 M-tiles, or the M-tile count of a short pass) */
static inline size_t tk_tiled_gemm_scratch(const TkGemm& g) {
    const int rows_blk = g.M < TK_TW_MAX_BLOCK_ROWS ? g.M : TK_TW_MAX_BLOCK_ROWS;
    const int mt = rows_blk > 128 ? 16 : rows_blk > 64 ? 8 : rows_blk > 32 ? 4 : rows_blk > 16 ? 2 : 1;
    const int64_t blk = (int64_t)mt * 16, padded = ((int64_t)g.M + blk - 1) / blk * blk;
    return (size_t)padded * (size_t)g.K;
}

struct TkAudioGpuOps {
    TkAsr* a;
    hipStream_t s;
    /* set by the engine around the per-utterance long passes (log-mel, encoder, cross keys / values): there the opt-in fast contraction applies.
     * The decode steps never take it — a choice by the row count of a launch would make an utterance's result depend on its batch */
    bool long_ok = false;
    /* the operand image of the last packed activation matrix: consecutive linear layers on the same input (q / k / v, cross k / v) pack it
     * once.  It sits on top of the arena and is dropped — its space handed back — by every other op, by any allocation of the graph and
     * by a GEMM that writes into the matrix it was made from. */
    struct { const float* A = nullptr; int M = 0, K = 0, lda = 0; float* img = nullptr; size_t mark = 0; bool live = false; } pk;
    /* linear layers on the SAME packed input (q / k / v of a layer, the cross keys and values) are collected — up to three — and go out as ONE
     * launch with a destination per segment (q | k | v: the query buffer and row p of the two caches); anything else flushes the batch first */
    struct { TkTiledGemm t{}; int n = 0; } pend;
    void flush() {
        if (pend.n == 0) return;
        pend.t.nseg = pend.n;
        int cols = 0;
        for (int i = 0; i < pend.n; ++i) cols += pend.t.row_tiles[i] * 16;
        pend.t.n_valid = cols;
        if (!tk_launch_gemm_tiled(pend.t, s)) a->launch_error = "tiled GEMM launch refused (batched decoder layers)";
        pend.n = 0;
    }
    void drop_image() {
        flush();
        if (pk.live) { a->arena_used = pk.mark; pk.live = false; }
    }
    float* alloc(size_t n) {
        drop_image();
        return raw_alloc(n);
    }
    float* raw_alloc(size_t n) {
        n = (n + 63) & ~(size_t)63;
        float* p = a->arena + a->arena_used;
        a->arena_used += n;
        return p;
    }
    int32_t* alloc_i32(size_t n) { return (int32_t*)alloc(n); }
    static bool short_pass(int M) { return M <= TK_TW_MAX_BLOCK_ROWS; }
    /* the packed image of a matrix a producer kernel is about to write (layer norm, decoder attention): lives on top of the arena like
     * the image gemm() packs itself, and is found there by the linear layers that follow */
    float* image_for(const float* A, int M, int K, int lda) {
        pk.mark = a->arena_used;
        TkGemm g{};
        g.M = M; g.K = K;
        pk.img = raw_alloc(tk_tiled_gemm_scratch(g));
        pk.A = A; pk.M = M; pk.K = K; pk.lda = lda; pk.live = true;
        return pk.img;
    }
    /* a linear layer whose weights have tiles runs on the tiled GEMM (same k-ascending fp32 chain, bit-identical, ~3x the rate of the
     * LDS-staged kernel on these shapes and one short launch instead of a latency chain for the decoder's few rows) */
    void gemm(const TkGemm& g) {
        if (a->fast && long_ok) { /* opt-in fast contraction of the long passes (TkAsr::fast): row-major operands, split-f16 MFMA where the shape allows */
            drop_image();
            TkGemm f = g;
            f.fast = 1;
            tk_launch_gemm(f, s);
            return;
        }
        const int idx = tk_tiled_gemm_applies(g) ? a->model->tensor_of(g.B) : -1;
        if (idx < 0 || !a->model->wt[(size_t)idx]) { drop_image(); tk_launch_gemm(g, s); return; }
        if (!(pk.live && pk.A == g.A && pk.M == g.M && pk.K == g.K && pk.lda == g.lda)) {
            drop_image();
            pk.mark = a->arena_used;
            pk.img = raw_alloc(tk_tiled_gemm_scratch(g));
            tk_launch_pack_a(g.A, g.M, g.K, g.lda, 0, pk.img, s);
            pk.A = g.A; pk.M = g.M; pk.K = g.K; pk.lda = g.lda; pk.live = true;
        }
        /* does the output overlap the packed matrix (an in-place layer)?  Then the image is stale once the launch has run */
        const float* c0 = g.C;
        const float* c1 = g.C + (size_t)(g.M - 1) * g.ldc + g.N;
        const float* a0 = g.A;
        const float* a1 = g.A + (size_t)(g.M - 1) * g.lda + g.K;
        const bool in_place = c0 < a1 && a0 < c1;
        const bool plain = !g.residual && !g.c_feeds_linear && g.N % 16 == 0 && !in_place; /* an in-place layer never joins a batch: it runs alone below */
        if (plain) { /* joins (or opens) the batch of layers on this input */
            if (pend.n > 0 && (pend.t.a_img != pk.img || pend.t.K != g.K || pend.t.nrows != g.M || pend.t.act != g.act)) flush();
            TkTiledGemm& t = pend.t;
            if (pend.n == 0) {
                t = TkTiledGemm{};
                t.wbytes = 4; t.K = g.K; t.ks = 1; t.nrows = g.M; t.slab_rows = 0; t.a_img = pk.img; t.a_ts = (size_t)g.K * 16;
                t.act = g.act; t.add_zero_bias = 1; t.per_seg = 1;
            }
            const int i = pend.n++;
            t.tiles[i] = a->model->wt[(size_t)idx]; t.row_tiles[i] = g.N / 16;
            t.seg_out[i] = g.C; t.seg_ldc[i] = g.ldc; t.seg_bias[i] = g.bias; t.seg_n[i] = g.N;
            if (pend.n == 3) flush();
            return; /* none of these layers writes into its own input: `plain` excludes in-place layers */
        }
        flush();
        TkTiledGemm t{};
        t.tiles[0] = a->model->wt[(size_t)idx]; t.row_tiles[0] = (g.N + 15) / 16; t.nseg = 1; t.wbytes = 4;
        t.K = g.K; t.ks = 1; t.ldc = g.ldc; t.n_valid = g.N; t.nrows = g.M; t.slab_rows = 0;
        t.a_img = pk.img; t.a_ts = (size_t)g.K * 16; t.out = g.C;
        t.bias = g.bias; t.residual = g.residual; t.ldr = g.ldr; t.act = g.act; t.add_zero_bias = 1;
        /* fc1: its output is only ever the input of fc2 — written as fc2's operand image as well (on top of fc1's own), on long (encoder) passes
         * too: there the second image replaces a k_pack_a launch over B x 1500 x 4 d values (9.8 -> 9.4 ms per call, profiles/r04_perception.txt) */
        const bool emit_img = g.c_feeds_linear && g.N % 128 == 0 && g.ldc == g.N && !g.residual && !in_place;
        float* cimg = nullptr;
        if (emit_img) {
            TkGemm nx{};
            nx.M = g.M; nx.K = g.N;
            cimg = raw_alloc(tk_tiled_gemm_scratch(nx));
            t.c_img = cimg;
        }
        if (!tk_launch_gemm_tiled(t, s)) tk_launch_gemm(g, s); /* not reached: tk_tiled_gemm_applies() admits only shapes the launcher takes */
        if (emit_img) { /* the image pair stays allocated from pk.mark up; the upper one is now the live image */
            pk.img = cimg; pk.A = g.C; pk.M = g.M; pk.K = g.N; pk.lda = g.ldc; pk.live = true;
            return;
        }
        if (in_place) drop_image(); /* the output overwrote the packed matrix: the image is stale */
    }
    void im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, float* col) { drop_image(); tk_launch_im2col1d(x, B, T, C, ldx, kw, stride, pad, col, s); }
    void layernorm(const float* x, int rows, int D, const float* w, const float* b, float* y) {
        drop_image();
        float* img = D % 128 == 0 && !(a->fast && long_ok) ? image_for(y, rows, D, D) : nullptr; /* a norm feeds linear layers: packed on the way out */
        tk_launch_layernorm(x, rows, D, w, b, TK_WH_LN_EPS, y, s, img);
    }
    void softmax_rows(float* x, int rows, int cols, int ld) { drop_image(); tk_launch_softmax_rows(x, rows, cols, ld, s); }
    bool attend1(const float* q, const float* k, const float* v, float* out, int B, int Tk, int64_t q_bstride, int64_t kv_bstride, int d, int nh) {
        drop_image();
        const int hd = d / nh;
        const size_t ldsb = ((size_t)TK_AT1_NS * TK_AT1_CK * hd + (size_t)Tk + (size_t)hd) * 4;
        if (hd > 256 || (hd & 15) || (d & 3) || Tk < 1 || ldsb > 60 * 1024) return false; /* the three-launch form takes what does not fit */
        float* img = (short_pass(B) && d % 128 == 0 && q_bstride == d) ? image_for(out, B, d, d) : nullptr; /* the output projection's input, packed on the way out */
        hipLaunchKernelGGL(k_attend1, dim3(nh, B), dim3(256), ldsb, s, q, k, v, out, Tk, q_bstride, kv_bstride, d, hd, tk_divf(1.0f, tk_sqrtf((float)hd)), img);
        return true;
    }
    /* the opt-in fast contraction's attention (TkAsr::fast): Q K^T, softmax and P V of a long pass in one kernel, no score matrix in HBM */
    bool attend_fused(const float* q, const float* k, const float* v, float* out, int B, int Tq, int Tk, int64_t q_bstride, int64_t kv_bstride, int d, int nh) {
        if (!a->fast || !long_ok || Tq < 2 || nh < 1 || d != nh * 64) return false;
        drop_image();
        return tk_launch_attention_h3(q, k, v, out, B, nh, Tq, Tk, q_bstride, kv_bstride, d, tk_divf(1.0f, tk_sqrtf(64.0f)), s);
    }
    void add_rows(float* x, const float* add, int rows, int D, int add_rows) { drop_image(); tk_launch_add_rows(x, add, rows, D, add_rows, s); }
    void embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int rows, int D, float* out) {
        drop_image();
        tk_launch_embed_rows(table, pos, idx, pos_idx, rows, D, out, s);
    }
    void argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out) {
        drop_image();
        if (!a->pick.on) { tk_launch_argmax_rows(x, rows, cols, ld, out, s); return; }
        /* whisper.cpp's decoding policy: the step's token by temperature, its log-probability noted (step p of the decode loop, row b: slot p * B + b) */
        TkPick pk{a->pick.temp, a->pick.seed, (uint32_t)(a->pick.step * rows), a->pick.logprob ? a->pick.logprob + (size_t)a->pick.step * rows : nullptr};
        if (a->pick.filtered) {
            /* prompt positions before the last one produce logits nobody samples from: the rows' decode state must not move there */
            if (a->pick.step >= a->pick.first_step) {
                TkWhFilter f{a->pick.suppress, a->pick.state, a->pick.beg, a->pick.eot, a->pick.tid0};
                tk_launch_pick_rows_filtered(x, rows, cols, ld, out, pk, f, s);
            }
        } else {
            tk_launch_pick_rows(x, rows, cols, ld, out, pk, s);
        }
        a->pick.step++;
    }
    void frames(const int16_t* pcm, int B, int n_samples, int pcm_stride, int n_total, int T, const float* window, float* out) {
        drop_image();
        const int64_t per = (int64_t)T * TK_WH_NFFT;
        hipLaunchKernelGGL(k_frames, dim3((unsigned)((per + 255) / 256), 1, B), dim3(256), 0, s, pcm, n_samples, pcm_stride, n_total, T, window, out);
    }
    void power(const float* ri, int rows, int nb, float* out) {
        drop_image();
        const int64_t n = (int64_t)rows * nb;
        hipLaunchKernelGGL(k_power, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ri, (int64_t)rows, nb, out);
    }
    void logmel_finish(float* mel, int B, int per_b) { drop_image(); hipLaunchKernelGGL(k_logmel_finish, dim3(B), dim3(1024), 0, s, mel, per_b); }
};

/* sizing twin of the ops (counts arena floats for a given batch) */
struct TkAudioSizeOps {
    size_t used = 0, peak = 0;
    float* alloc(size_t n) { used += (n + 63) & ~(size_t)63; if (used > peak) peak = used; return nullptr; }
    int32_t* alloc_i32(size_t n) { return (int32_t*)alloc(n); }
    void gemm(const TkGemm& g) { /* the tiled path's activation image is scratch on top of the live allocations; fc1 of a short pass adds fc2's image */
        if (!tk_tiled_gemm_applies(g)) return;
        size_t top = used + ((tk_tiled_gemm_scratch(g) + 63) & ~(size_t)63);
        if (g.c_feeds_linear) { TkGemm nx{}; nx.M = g.M; nx.K = g.N; top += (tk_tiled_gemm_scratch(nx) + 63) & ~(size_t)63; }
        if (top > peak) peak = top;
    }
    void image(int M, int K) { /* a producer's packed output (layer norm, decoder attention) */
        TkGemm g{};
        g.M = M; g.K = K;
        const size_t top = used + ((tk_tiled_gemm_scratch(g) + 63) & ~(size_t)63);
        if (top > peak) peak = top;
    }
    void im2col1d(const float*, int, int, int, int, int, int, int, float*) {}
    void layernorm(const float*, int rows, int D, const float*, const float*, float*) { image(rows, D); }
    void softmax_rows(float*, int, int, int) {}
    bool attend1(const float*, const float*, const float*, float*, int B, int, int64_t, int64_t, int d, int) { image(B, d); return true; }
    bool attend_fused(const float*, const float*, const float*, float*, int, int, int, int64_t, int64_t, int, int) { return false; } /* sized for the exact form */
    void add_rows(float*, const float*, int, int, int) {}
    void embed_rows(const float*, const float*, const int32_t*, const int32_t*, int, int, float*) {}
    void argmax_rows(const float*, int, int, int, int32_t*) {}
    void frames(const int16_t*, int, int, int, int, int, const float*, float*) {}
    void power(const float*, int, int, float*) {}
    void logmel_finish(float*, int, int) {}
};

/* ---------------------------------------------------------------- model */

TkWhisperModel::~TkWhisperModel() {
    (void)hipSetDevice(device);
    for (auto p : w) if (p) (void)hipFree(p);
    for (auto p : wt) if (p) (void)hipFree(p);
}

int TkWhisperModel::tensor_of(const float* dev_ptr) const {
    for (size_t i = 0; i < w.size(); ++i) if (w[i] == dev_ptr) return (int)i;
    return -1;
}

bool TkWhisperModel::prepare_tiles() {
    HIPQ(hipSetDevice(device));
    if (wt.size() != man.t.size()) wt.assign(man.t.size(), nullptr);
    for (size_t i = 0; i < man.t.size(); ++i) {
        const bool linear = man.t[i].kind == TK_WK_LINEAR_W || (int)i == man.tok_emb; /* the token embedding is also the logits matrix */
        if (!linear || man.t[i].cols % 128 || man.t[i].rows < 16 || wt[i]) continue;
        HIPQ(hipMalloc((void**)&wt[i], tk_tiled_weight_bytes(man.t[i].rows, man.t[i].cols, 4)));
        tk_launch_tile_weights(w[i], 4, man.t[i].rows, man.t[i].cols, wt[i], nullptr);
    }
    HIPQ(hipDeviceSynchronize());
    return true;
}

bool TkWhisperModel::init(const TkWhisperHP& h, int dev) {
    hp = h;
    device = dev;
    if (h.n_audio_state % h.n_audio_head || h.n_text_state % h.n_text_head || h.n_audio_state % 2 || h.n_mels <= 0 || h.n_audio_ctx <= 0 ||
        h.n_text_ctx <= 0 || h.n_vocab <= 0) { error = "invalid Whisper hyper-parameters"; return false; }
    man = tk_whisper_manifest(h);
    HIPQ(hipSetDevice(device));
    w.assign(man.t.size(), nullptr);
    for (size_t i = 0; i < man.t.size(); ++i) HIPQ(hipMalloc((void**)&w[i], (size_t)man.t[i].rows * man.t[i].cols * 4));
    return true;
}

bool TkWhisperModel::set_tensor(int idx, const float* host, size_t n) {
    if (idx < 0 || idx >= (int)man.t.size() || n != (size_t)man.t[idx].rows * man.t[idx].cols) { error = "tensor index / size mismatch"; return false; }
    HIPQ(hipSetDevice(device));
    HIPQ(hipMemcpy(w[idx], host, n * 4, hipMemcpyHostToDevice));
    if ((size_t)idx < wt.size() && wt[(size_t)idx]) { /* keep the tiled twin current */
        tk_launch_tile_weights(w[idx], 4, man.t[idx].rows, man.t[idx].cols, wt[(size_t)idx], nullptr);
        HIPQ(hipDeviceSynchronize());
    }
    return true;
}

bool TkWhisperModel::fill_synthetic(uint64_t seed) {
    std::vector<float> buf;
    for (size_t i = 0; i < man.t.size(); ++i) {
        buf.resize((size_t)man.t[i].rows * man.t[i].cols);
        tk_whisper_fill_tensor(hp, man, (int)i, seed, buf.data());
        if (!set_tensor((int)i, buf.data(), buf.size())) return false;
    }
    return true;
}

bool TkWhisperModel::load_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { error = std::string("cannot open ") + path; return false; }
    char magic[8];
    int32_t n = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "TKWHSP1", 8) == 0 && fread(&n, 4, 1, f) == 1 && n == (int)man.t.size();
    std::vector<float> buf;
    /* the three frontend tables and the sinusoidal positions may be omitted (rows == 0 in the file): they are recomputed */
    for (int i = 0; ok && i < n; ++i) {
        int64_t d[2];
        ok = fread(d, 8, 2, f) == 2;
        if (!ok) break;
        if (d[0] == 0) {
            buf.resize((size_t)man.t[i].rows * man.t[i].cols);
            tk_whisper_fill_tensor(hp, man, i, 0, buf.data());
        } else {
            ok = d[0] == man.t[i].rows && d[1] == man.t[i].cols;
            buf.resize((size_t)d[0] * d[1]);
            ok = ok && fread(buf.data(), 4, buf.size(), f) == buf.size();
        }
        ok = ok && set_tensor(i, buf.data(), buf.size());
    }
    fclose(f);
    if (!ok && error.empty()) error = "not a TKWHSP1 container for this geometry";
    return ok;
}

bool TkWhisperModel::load_ggml(TkWhisperGgml& g) {
    std::vector<float> buf;
    for (int i = 0; i < (int)man.t.size(); ++i) {
        bool found = false;
        if (!g.read(man, i, &buf, &found)) {
            if (found) { error = g.error; return false; }
            if (i != man.hann && i != man.dft) { error = "tensor " + man.t[i].name + " is missing from the checkpoint"; return false; }
            buf.resize((size_t)man.t[i].rows * man.t[i].cols);
            tk_whisper_fill_tensor(hp, man, i, 0, buf.data()); /* analysis window and DFT twiddles are computed, not stored */
        }
        if (!set_tensor(i, buf.data(), buf.size())) return false;
    }
    return true;
}

/* ---------------------------------------------------------------- ASR */

TkAsr::~TkAsr() {
    if (model) (void)hipSetDevice(model->device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (arena) (void)hipFree(arena);
    if (pcm_dev) (void)hipFree(pcm_dev);
    if (suppress_dev) (void)hipFree(suppress_dev);
    if (lp_dev) (void)hipFree(lp_dev);
    if (wh_state) (void)hipFree(wh_state);
    if (stream) (void)hipStreamDestroy(stream);
}

bool TkAsr::init(TkWhisperModel* m, int mb) {
    model = m;
    max_batch = mb;
    if (!m || mb < 1 || mb > 256) { error = "max_batch must be in [1,256]"; return false; }
    HIPQ(hipSetDevice(m->device));
    HIPQ(tk_create_perception_stream(&stream));
    TkAudioSizeOps so;
    std::vector<float*> nullw(m->man.t.size(), nullptr);
    TkWhisperGraph<TkAudioSizeOps> g{m->hp, m->man, nullw.data()};
    float* ml = g.mel(so, nullptr, mb, 0, 0);
    float* enc = g.encode(so, ml, mb);
    (void)g.begin_decode(so, enc, mb);
    arena_floats = so.peak + 4096 + (size_t)3 * m->hp.n_text_ctx * mb + 64; /* + the decode loop's token / position / arg-max tables */
    if (!m->prepare_tiles()) { error = m->error; return false; }
    if (!tk_gemm_tiled_prepare_device()) { error = "LDS opt-in of the tiled GEMM failed"; return false; }
    HIPQ(hipMalloc((void**)&arena, arena_floats * 4));
    return true;
}

bool TkAsr::transcribe_policy(int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps, float temperature, uint64_t seed,
                              int32_t* tokens_out, float* logprobs_out) {
    if (!(temperature >= 0.0f) || model->hp.n_vocab > 65536) { error = "temperature must be >= 0 and the vocabulary at most 65536 tokens"; return false; }
    const int total = n_prompt + n_steps - 1;
    if (B < 1 || total < 1) { error = "nothing to decode"; return false; }
    if (hipSetDevice(model->device) != hipSuccess) { error = "hipSetDevice failed"; return false; }
    if (!ensure_decode_buffers()) return false;
    if (total > model->hp.n_text_ctx || B > max_batch) { error = "prompt + steps exceed the text context"; return false; }
    float* d_lp = lp_dev;
    pick.on = true; pick.temp = temperature; pick.seed = seed; pick.logprob = d_lp; pick.step = 0;
    const bool ok = transcribe(B, pcm, n_samples, prompt, n_prompt, n_steps, tokens_out, nullptr, nullptr, nullptr); /* synchronises the stream */
    pick.on = false;
    bool copied = true;
    if (ok && logprobs_out) {
        pick_logprobs.resize((size_t)total * B);
        copied = hipMemcpyAsync(pick_logprobs.data(), d_lp, pick_logprobs.size() * sizeof(float), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
        for (int step = 0; copied && step < n_steps; ++step)
            for (int b = 0; b < B; ++b) logprobs_out[(size_t)b * n_steps + step] = pick_logprobs[(size_t)(n_prompt - 1 + step) * B + b];
    }
    pick.logprob = nullptr;
    if (ok && !copied) error = "copy of the log-probabilities failed";
    return ok && copied;
}

/* the decode's small device buffers, allocated ONCE per engine: nothing on a call's path allocates or frees device memory or uses the null stream —
 * hipFree synchronises the whole device and, issued while another host thread captures a hipGraph (the LLM's passes are captured at first use of
 * every row count), was the prime suspect of an intermittent process-wide stall on cold boxes (round 6, 64 cortices) */
bool TkAsr::ensure_decode_buffers() {
    const TkWhisperHP& h = model->hp;
    if (!lp_dev) HIPQ(hipMalloc((void**)&lp_dev, (size_t)max_batch * h.n_text_ctx * sizeof(float)));
    if (!suppress_dev) HIPQ(hipMalloc((void**)&suppress_dev, (size_t)h.n_vocab));
    if (!wh_state) HIPQ(hipMalloc((void**)&wh_state, (size_t)max_batch * TK_WH_STATE_INTS * sizeof(int32_t)));
    return true;
}

bool TkAsr::transcribe_ref(int B, const int16_t* pcm, int n_samples, const int32_t* n_samples_row, const int32_t* prompt, int n_prompt, int n_steps, float temperature,
                           uint64_t seed, const uint8_t* suppress, int32_t token_beg, int32_t token_eot, int32_t* tokens_out, float* logprobs_out, int32_t* result_len,
                           int32_t* status) {
    const TkWhisperHP& h = model->hp;
    if (!(temperature >= 0.0f) || h.n_vocab > 65536) { error = "temperature must be >= 0 and the vocabulary at most 65536 tokens"; return false; }
    if (B < 1 || B > max_batch || !suppress || !n_samples_row || token_beg <= token_eot || token_beg >= h.n_vocab) { error = "bad arguments to the reference-parameter decode"; return false; }
    const int total = n_prompt + n_steps - 1;
    if (total < 1) { error = "nothing to decode"; return false; }
    if (hipSetDevice(model->device) != hipSuccess) { error = "hipSetDevice failed"; return false; }
    if (!ensure_decode_buffers()) return false;
    if (total > h.n_text_ctx) { error = "prompt + steps exceed the text context"; return false; }
    if (suppress_host.size() != (size_t)h.n_vocab || memcmp(suppress_host.data(), suppress, (size_t)h.n_vocab) != 0) {
        HIPQ(hipStreamSynchronize(stream)); /* the previous decode may still read the table */
        suppress_host.assign(suppress, suppress + h.n_vocab);
        HIPQ(hipMemcpyAsync(suppress_dev, suppress_host.data(), (size_t)h.n_vocab, hipMemcpyHostToDevice, stream));
    }
    std::vector<int32_t> st0((size_t)B * TK_WH_STATE_INTS, 0);
    for (int b = 0; b < B; ++b) st0[(size_t)b * TK_WH_STATE_INTS + 7] = n_samples_row[b] / TK_WH_HOP; /* seek_end: the utterance in 10 ms frames */
    HIPQ(hipMemcpyAsync(wh_state, st0.data(), st0.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    float* d_lp = lp_dev;
    HIPQ(hipMemsetAsync(d_lp, 0, (size_t)total * B * sizeof(float), stream));
    pick.on = true; pick.temp = temperature; pick.seed = seed; pick.logprob = d_lp; pick.step = 0;
    pick.filtered = true; pick.first_step = n_prompt - 1; pick.suppress = suppress_dev; pick.state = wh_state; pick.beg = token_beg; pick.eot = token_eot;
    pick.tid0 = 50; /* max_initial_ts 1.0 s / (30 s / 1500 positions) */
    const bool ok = transcribe(B, pcm, n_samples, prompt, n_prompt, n_steps, tokens_out, nullptr, nullptr, nullptr); /* synchronises the stream */
    pick.on = false; pick.filtered = false;
    bool copied = true;
    std::vector<int32_t> st1((size_t)B * TK_WH_STATE_INTS);
    if (ok) {
        pick_logprobs.resize((size_t)total * B);
        copied = hipMemcpyAsync(pick_logprobs.data(), d_lp, pick_logprobs.size() * sizeof(float), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipMemcpyAsync(st1.data(), wh_state, st1.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
        for (int b = 0; copied && b < B; ++b) {
            const int32_t* sb = &st1[(size_t)b * TK_WH_STATE_INTS];
            if (status) status[b] = sb[6];
            /* n_steps reached before the row ended: every sampled token counts (whisper.cpp would go on to n_text_ctx / 2 - 4 tokens) */
            if (result_len) result_len[b] = sb[6] == 0 ? sb[0] : sb[5];
            for (int step = 0; logprobs_out && step < n_steps; ++step) logprobs_out[(size_t)b * n_steps + step] = pick_logprobs[(size_t)(n_prompt - 1 + step) * B + b];
        }
    }
    pick.logprob = nullptr;
    if (ok && !copied) error = "copy of the decode state failed";
    return ok && copied;
}

bool TkAsr::transcribe(int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps, int32_t* tokens_out,
                       std::vector<float>* mel_out, std::vector<float>* enc_out, std::vector<float>* first_logits) {
    const TkWhisperHP& h = model->hp;
    if (B < 1 || B > max_batch) { error = "batch larger than the ASR context was created for"; return false; }
    if (n_samples < 0 || n_samples > h.n_samples()) { error = "audio longer than the model window"; return false; }
    if (n_prompt < 1 || n_prompt + n_steps > h.n_text_ctx) { error = "prompt + steps exceed the text context"; return false; }
    HIPQ(hipSetDevice(model->device));
    size_t need = (size_t)B * (n_samples > 0 ? n_samples : 1);
    if (need > pcm_cap) need = (size_t)max_batch * (size_t)h.n_samples(); /* once: the engine's whole batch of full windows (0.96 MB each) */
    if (need > pcm_cap) {
        if (pcm_dev) (void)hipFree(pcm_dev);
        pcm_dev = nullptr;
        HIPQ(hipMalloc((void**)&pcm_dev, need * 2));
        pcm_cap = need;
    }
    if (n_samples > 0) HIPQ(hipMemcpyAsync(pcm_dev, pcm, (size_t)B * n_samples * 2, hipMemcpyHostToDevice, stream));
    arena_used = 0;
    launch_error.clear();
    TkAudioGpuOps ops{this, stream};
    TkWhisperGraph<TkAudioGpuOps> g{h, model->man, model->w.data()};
    ops.long_ok = true;
    float* ml = g.mel(ops, pcm_dev, B, n_samples, n_samples);
    float* enc = g.encode(ops, ml, B);
    auto st = g.begin_decode(ops, enc, B);
    ops.long_ok = false;
    HIPQ(hipGetLastError());
    if (mel_out) { mel_out->resize((size_t)B * h.n_frames() * h.n_mels); HIPQ(hipMemcpyAsync(mel_out->data(), ml, mel_out->size() * 4, hipMemcpyDeviceToHost, stream)); }
    if (enc_out) { enc_out->resize((size_t)B * h.n_audio_ctx * h.n_audio_state); HIPQ(hipMemcpyAsync(enc_out->data(), enc, enc_out->size() * 4, hipMemcpyDeviceToHost, stream)); }
    /* The whole greedy loop is enqueued without touching the host: prompt tokens and positions sit in device tables uploaded once, step p
     * reads its token from the prompt table or from the previous step's arg-max slot and writes its own arg-max into slot p of an output
     * table; one copy and one synchronisation at the end.  (A synchronisation and three small copies per step made the 16 steps of 32
     * utterances 1.2 ms each on a 0.3 ms GPU workload.) */
    const int total = n_prompt + n_steps - 1; /* the last generated token is not fed back */
    const size_t tab_ints = ((size_t)n_prompt + 2 * (size_t)total) * B;
    ops.drop_image();
    if (arena_used + tab_ints + 64 > arena_floats) { error = "ASR arena too small for the decode tables"; return false; }
    int32_t* tab = ops.alloc_i32(tab_ints); /* [n_prompt][B] prompt tokens, [total][B] positions, [total][B] arg-max slots */
    int32_t* tok_tab = tab;
    int32_t* pos_tab = tab + (size_t)n_prompt * B;
    int32_t* out_tab = pos_tab + (size_t)total * B;
    std::vector<int32_t> host_tab(((size_t)n_prompt + total) * B);
    for (int p = 0; p < n_prompt; ++p)
        for (int b = 0; b < B; ++b) host_tab[(size_t)p * B + b] = prompt[p];
    for (int p = 0; p < total; ++p)
        for (int b = 0; b < B; ++b) host_tab[((size_t)n_prompt + p) * B + b] = p;
    HIPQ(hipMemcpyAsync(tok_tab, host_tab.data(), host_tab.size() * 4, hipMemcpyHostToDevice, stream));
    int steps_run = n_steps; /* sampled positions really enqueued */
    std::vector<int32_t> st_host;
    for (int p = 0; p < total; ++p) {
        st.tok = p < n_prompt ? tok_tab + (size_t)p * B : out_tab + (size_t)(p - 1) * B;
        st.pos = pos_tab + (size_t)p * B;
        st.next = out_tab + (size_t)p * B;
        g.decode_step(ops, st, p);
        if (p == n_prompt - 1 && first_logits) {
            first_logits->resize((size_t)B * h.n_vocab);
            HIPQ(hipMemcpyAsync(first_logits->data(), st.logits, first_logits->size() * 4, hipMemcpyDeviceToHost, stream));
        }
        /* the reference-parameter decode ends where the rows' own bookkeeping says (whisper.cpp: up to n_text_ctx / 2 - 4 tokens): every 16 sampled
         * positions the rows' status words are read; once every row has completed or failed nothing more is enqueued — a finished row emits eot and
         * stands still anyway (k_pick_rows_filtered), so stopping early changes no token */
        const int step = p - (n_prompt - 1);
        if (pick.filtered && step >= 0 && (step + 1) % 16 == 0 && p + 1 < total) {
            ops.flush();
            st_host.resize((size_t)B * TK_WH_STATE_INTS);
            HIPQ(hipMemcpyAsync(st_host.data(), pick.state, st_host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            HIPQ(hipStreamSynchronize(stream));
            bool all_done = true;
            for (int b = 0; b < B; ++b) all_done = all_done && st_host[(size_t)b * TK_WH_STATE_INTS + 6] != 0;
            if (all_done) { steps_run = step + 1; break; }
        }
    }
    ops.flush();
    std::vector<int32_t> outs((size_t)steps_run * B);
    HIPQ(hipMemcpyAsync(outs.data(), out_tab + (size_t)(n_prompt - 1) * B, outs.size() * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream)); /* host_tab and outs live until here */
    for (int step = 0; step < n_steps; ++step)
        for (int b = 0; b < B; ++b) tokens_out[(size_t)b * n_steps + step] = step < steps_run ? outs[(size_t)step * B + b] : pick.eot;
    HIPQ(hipGetLastError());
    if (!launch_error.empty()) { error = launch_error; return false; }
    return true;
}

/* ---------------------------------------------------------------- VAD */

TkVadModel::~TkVadModel() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    float* ptrs[] = {w1, b1, w2, b2, x, hbuf, p};
    for (float* q : ptrs) if (q) (void)hipFree(q);
    if (stream) (void)hipStreamDestroy(stream);
}

bool TkVadModel::init(int dev, int win, int hid) {
    device = dev; window = win; hidden = hid;
    HIPQ(hipSetDevice(device));
    HIPQ(hipMalloc((void**)&w1, (size_t)hid * win * 4));
    HIPQ(hipMalloc((void**)&b1, (size_t)hid * 4));
    HIPQ(hipMalloc((void**)&w2, (size_t)hid * 4));
    HIPQ(hipMalloc((void**)&b2, 4));
    return true;
}

bool TkVadModel::fill_synthetic(uint64_t seed) {
    std::vector<float> a((size_t)hidden * window), b(hidden), c(hidden), d(1);
    tk_vad_synth(seed, window, hidden, a.data(), b.data(), c.data(), d.data());
    HIPQ(hipSetDevice(device));
    HIPQ(hipMemcpy(w1, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(b1, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(w2, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(b2, d.data(), 4, hipMemcpyHostToDevice));
    return true;
}

/* relu is not in TkAct: hidden = max(0, .) is folded into a tiny kernel together with the second layer */
__global__ void k_vad_head(const float* hid, int n, int hidden, const float* w2, const float* b2, float* prob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int k = 0; k < hidden; ++k) acc = tk_fmaf(tk_fmaxf(hid[(int64_t)i * hidden + k], 0.0f), w2[k], acc);
    prob[i] = tk_sigmoidf(acc + b2[0]);
}

bool TkVadModel::infer(const float* windows_host, int n, float* prob_host) {
    if (n <= 0) return true;
    HIPQ(hipSetDevice(device));
    if (!stream) HIPQ(tk_create_perception_stream(&stream));
    if (n > cap) { /* grow-only scratch: no allocation on the steady-state path */
        if (x) (void)hipFree(x);
        if (hbuf) (void)hipFree(hbuf);
        if (p) (void)hipFree(p);
        x = hbuf = p = nullptr;
        cap = 0;
        const int want = n < 128 ? 128 : n;
        HIPQ(hipMalloc((void**)&x, (size_t)want * window * 4));
        HIPQ(hipMalloc((void**)&hbuf, (size_t)want * hidden * 4));
        HIPQ(hipMalloc((void**)&p, (size_t)want * 4));
        cap = want;
    }
    HIPQ(hipMemcpyAsync(x, windows_host, (size_t)n * window * 4, hipMemcpyHostToDevice, stream));
    TkGemm g{};
    g.A = x; g.B = w1; g.C = hbuf; g.bias = b1; g.M = n; g.N = hidden; g.K = window; g.lda = window; g.ldb = window; g.ldc = hidden;
    g.alpha = 1.0f; g.batch = 1;
    tk_launch_gemm(g, stream);
    hipLaunchKernelGGL(k_vad_head, dim3((n + 63) / 64), dim3(64), 0, stream, hbuf, n, hidden, w2, b2, p);
    HIPQ(hipGetLastError());
    HIPQ(hipMemcpyAsync(prob_host, p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    return true;
}
