#include "tk_nn_kernels.h"

#include <atomic>
#include <mutex>

#include "../common/tk_exact_math.h"
#include "../common/tk_sample_device.h"

typedef float v16f __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDS_LD (BK + 1)

/* B element i (in elements) of an f32 or f16 matrix; 4 consecutive elements as one 16- or 8-byte load */
__device__ __forceinline__ float ldb1(const float* B, int f16, int64_t i) {
    return f16 ? (float)((const _Float16*)B)[i] : B[i];
}
__device__ __forceinline__ void ldb4(const float* B, int f16, int64_t i, float* out) {
    if (f16) {
        const uint2 t = *(const uint2*)((const _Float16*)B + i);
        const _Float16* h = (const _Float16*)&t;
        out[0] = (float)h[0]; out[1] = (float)h[1]; out[2] = (float)h[2]; out[3] = (float)h[3];
    } else {
        const float4 t = *(const float4*)(B + i);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case TK_ACT_SILU: return tk_siluf(v);
        case TK_ACT_GELU: return tk_geluf(v);
        case TK_ACT_SIGMOID: return tk_sigmoidf(v);
        default: return v;
    }
}

/* Implicit im2col (TkGemm::im_C > 0): A is a convolution's NHWC input and GEMM row m = (b, oy, ox), column k = (ky kw + kx) C + c are
 * decoded on the fly, so the column matrix (kh kw times the input) is never written or read.  A thread's rows are fixed for the whole
 * launch (decoded once) and its k advances by one slab per load: (c, kx, ky) are carried, no division in the loop.  Same values, same
 * chain: bit-identical to the explicit matrix. */
struct ImRow { int64_t pix0; int iy0, ix0; bool ok; };
struct ImCol { int c, kx, ky; };
__device__ __forceinline__ ImRow im_row(const TkGemm& g, int m) {
    ImRow r;
    r.ok = m < g.M;
    const int mm = r.ok ? m : 0;
    const int ox = mm % g.im_Wo, t = mm / g.im_Wo, oy = t % g.im_Ho, b = t / g.im_Ho;
    r.iy0 = oy * g.im_stride - g.im_pad;
    r.ix0 = ox * g.im_stride - g.im_pad;
    r.pix0 = (int64_t)b * g.im_H * g.im_W;
    return r;
}
__device__ __forceinline__ ImCol im_col(const TkGemm& g, int k) {
    ImCol q;
    const int tap = k / g.im_C;
    q.c = k - tap * g.im_C; q.ky = tap / g.im_kw; q.kx = tap - q.ky * g.im_kw;
    return q;
}
__device__ __forceinline__ void im_advance(const TkGemm& g, ImCol& q, int dk) {
    q.c += dk;
    while (q.c >= g.im_C) { q.c -= g.im_C; if (++q.kx == g.im_kw) { q.kx = 0; ++q.ky; } }
}
__device__ __forceinline__ void im_load4(const TkGemm& g, const float* X, const ImRow& r, const ImCol& q, int k, float out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    const int iy = r.iy0 + q.ky, ix = r.ix0 + q.kx;
    if (r.ok && k < g.K && iy >= 0 && iy < g.im_H && ix >= 0 && ix < g.im_W) {
        const float4 t = *(const float4*)(X + (r.pix0 + (int64_t)iy * g.im_W + ix) * g.im_ldx + q.c);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
    }
}

/*
 * Operand loads of the three GEMM kernels below, 4 consecutive elements (one k-group, or one n-group of a [K][N] matrix) per call.
 * PATH 0 is the general case: any pitch, alignment and extent, f32 or f16 B — element-wise fallbacks under run-time tests, which also
 * keep the compiler from batching a slab's loads (each waits for the previous one: measured 2x on the Whisper Q.K^T launch).
 * PATH 1 ([N][K] weights) and PATH 2 ([K][N]) are the same loads when tk_gemm_fast() holds — every pitch, batch offset and extent along
 * the vector a multiple of 4, bases 16-byte aligned, f32 — as ONE predicated 16-byte load each, so all of a slab's loads are in flight
 * together.  Values and order of the arithmetic are identical on every path.
 */
__device__ __forceinline__ void ld4_if(bool ok, const float* p, float out[4]) {
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (ok) t = *(const float4*)p;
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}
/* A[m][k .. k+3] */
template <int PATH>
__device__ __forceinline__ void load_a4(const TkGemm& g, const float* A, bool a_vec, int m, int k, float out[4]) {
    if (PATH != 0) { ld4_if(m < g.M && k < g.K, A + (int64_t)m * g.lda + k, out); return; }
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (m < g.M) {
        const float* p = A + (int64_t)m * g.lda + k;
        if (a_vec && k + 3 < g.K) { const float4 t = *(const float4*)p; out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (k + i < g.K) out[i] = p[i];
        }
    }
}
/* B as [N][ldb]: B[n][k .. k+3] */
template <int PATH>
__device__ __forceinline__ void load_b4_nk(const TkGemm& g, const float* B, bool b_vec, int n, int k, float out[4]) {
    if (PATH != 0) { ld4_if(n < g.N && k < g.K, B + (int64_t)n * g.ldb + k, out); return; }
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (n < g.N) {
        const int64_t p = (int64_t)n * g.ldb + k;
        if (b_vec && k + 3 < g.K) ldb4(B, g.b_f16, p, out);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (k + i < g.K) out[i] = ldb1(B, g.b_f16, p + i);
        }
    }
}
/* B as [K][ldb]: B[k][n .. n+3] */
template <int PATH>
__device__ __forceinline__ void load_b4_kn(const TkGemm& g, const float* B, bool b_vec, int k, int n, float out[4]) {
    if (PATH != 0) { ld4_if(k < g.K && n < g.N, B + (int64_t)k * g.ldb + n, out); return; }
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (k < g.K) {
        const int64_t p = (int64_t)k * g.ldb + n;
        if (b_vec && n + 3 < g.N) ldb4(B, g.b_f16, p, out);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (n + i < g.N) out[i] = ldb1(B, g.b_f16, p + i);
        }
    }
}

/* epilogue of one 32x32 accumulator tile: v = act(alpha acc + bias[n]) + residual, rows m_base + (r % 4) + 8 (r / 4) + 4 (lane / 32),
 * column n.  The activation is chosen once per tile, not per element. */
/* activations of the opt-in fast contraction: the hardware's exp2 / reciprocal (~1 ulp each) instead of the exact-math sequences, whose
 * ~70 instructions per element outweigh a split-f16 launch's matrix work on short K */
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float fast_geluf(float x) {
    const float inner = 0.797884560802865356f * (x + 0.044715f * ((x * x) * x));
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(inner * 2.8853900817779268f) + 1.0f);
    return (0.5f * x) * (1.0f + th);
}

template <bool FASTACT = false>
__device__ __forceinline__ void store_tile(const TkGemm& g, const v16f& acc, float* C, const float* R, int m_base, int n, int lane) {
    if (n >= g.N) return;
    const float bias = g.bias ? g.bias[n] : 0.0f;
    const float alpha = g.alpha;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    if (alpha != 1.0f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] * alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + bias;
    if (g.act == TK_ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = FASTACT ? v[r] * fast_sigmoidf(v[r]) : tk_siluf(v[r]);
    } else if (g.act == TK_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = FASTACT ? fast_geluf(v[r]) : tk_geluf(v[r]);
    } else if (g.act == TK_ACT_SIGMOID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = FASTACT ? fast_sigmoidf(v[r]) : tk_sigmoidf(v[r]);
    }
    const int mb = m_base + 4 * (lane >> 5);
    float* cp = C + (int64_t)mb * g.ldc + n;
    if (m_base + 32 <= g.M) { /* the whole tile lies inside the matrix (wave-uniform): straight-line stores, no test per row */
        if (R) {
            const float* rp = R + (int64_t)mb * g.ldr + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] + rp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldr];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = v[r];
        return;
    }
    if (R) {
        const float* rp = R + (int64_t)mb * g.ldr + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2);
            if (mb + dm < g.M) v[r] = v[r] + rp[(int64_t)dm * g.ldr];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2);
        if (mb + dm < g.M) cp[(int64_t)dm * g.ldc] = v[r];
    }
}

__device__ __forceinline__ void batch_offsets(const TkGemm& g, int z, int64_t& oA, int64_t& oB, int64_t& oC, int64_t& oR) {
    if (g.batch_inner > 0) {
        const int zo = z / g.batch_inner, zi = z % g.batch_inner;
        oA = zo * g.sA2 + zi * g.sA; oB = zo * g.sB2 + zi * g.sB; oC = zo * g.sC2 + zi * g.sC; oR = zo * g.sR2 + zi * g.sR;
    } else {
        oA = (int64_t)z * g.sA; oB = (int64_t)z * g.sB; oC = (int64_t)z * g.sC; oR = (int64_t)z * g.sR;
    }
}

/*
 * 64x64 output tile per workgroup, 4 waves in 2x2, one 32x32 fp32 MFMA accumulator per wave.
 * A/B k-slabs of 32 are staged through LDS with 16 B coalesced loads (row pitch 33 floats:
 * the 32 lanes of an MFMA operand read hit 32 different banks); the next slab's global loads are
 * issued before the current slab's MFMAs, so small launches (a few workgroups, the Whisper decoder
 * steps) are not a chain of exposed load latencies.
 */
template <bool IM, int PATH>
__global__ __launch_bounds__(256) void k_gemm_f32(TkGemm g) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Bs[BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int64_t oA, oB, oC, oR;
    batch_offsets(g, blockIdx.z, oA, oB, oC, oR);
    const float* A = g.A + oA;
    const float* B = g.b_f16 ? (const float*)((const _Float16*)g.B + oB) : g.B + oB;
    float* C = g.C + oC;
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const bool a_vec = (g.lda & 3) == 0 && (((uintptr_t)A) & 15) == 0;
    const bool b_vec = (g.ldb & 3) == 0 && (((uintptr_t)B) & (g.b_f16 ? 7 : 15)) == 0;
    float ra[2][4], rb[2][4]; /* the next k slab on its way from global memory while the current one feeds the MFMAs */
    ImRow ir[2]; ImCol ic;
    if (IM) {
        ir[0] = im_row(g, m0 + (tid >> 3)); ir[1] = im_row(g, m0 + ((tid + 256) >> 3));
        ic = im_col(g, (tid & 7) * 4);
    }

    /* A: 64 rows x 32 k, thread -> (row = e / 8, 4 consecutive k) for e = tid, tid + 256; B likewise ([N][K]) or, for [K][N],
     * 4 consecutive n of one k (scattered transposed into LDS) */
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = tid + 256 * h, row = e >> 3, k = k0 + (e & 7) * 4;
            if (IM) im_load4(g, A, ir[h], ic, k, ra[h]);
            else load_a4<PATH>(g, A, a_vec, m0 + row, k, ra[h]);
            if (PATH == 2 || (PATH == 0 && g.b_kn)) load_b4_kn<PATH>(g, B, b_vec, k0 + (e >> 4), n0 + (e & 15) * 4, rb[h]);
            else load_b4_nk<PATH>(g, B, b_vec, n0 + row, k, rb[h]);
        }
        if (IM) im_advance(g, ic, BK);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = tid + 256 * h;
            const int row = e >> 3, kc = (e & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) As[row * LDS_LD + kc + i] = ra[h][i];
            if (PATH == 2 || (PATH == 0 && g.b_kn)) {
                const int kk = e >> 4, nc = (e & 15) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[(nc + i) * LDS_LD + kk] = rb[h][i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[row * LDS_LD + kc + i] = rb[h][i];
            }
        }
    };

    gload(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < g.K) gload(k0 + BK); /* in flight under this slab's 16 MFMAs */
        const float* ap = As + (wm * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
        const float* bp = Bs + (wn * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
        __syncthreads();
    }
    store_tile(g, acc, C, g.residual ? g.residual + oR : nullptr, m0 + wm * 32, n0 + wn * 32 + (lane & 31), lane);
}

/*
 * Narrow outputs (N < 96: the detector's high-resolution layers have 16 .. 80 output channels and up to hundreds of thousands of rows):
 * the 64x64 tile above spends MFMAs on columns that do not exist (N = 16, 32, 80) and reads A once per 64 columns.  Here a workgroup
 * covers 128 rows x ALL columns, its four waves stacked in M with NT = ceil(N / 32) accumulators each: A — the expensive operand, above
 * all when it is a convolution input addressed on the fly — is read exactly once, and one operand read feeds NT MFMAs.  Same slabs, same
 * k-ordered chain per output element (bit-identical); A [M][lda] and B [N][ldb] with k contiguous only, f32, no batch.
 */
#define NBM 128
template <bool IM, int PATH, int NT>
__global__ __launch_bounds__(256) void k_gemm_f32_tall(TkGemm g) {
    __shared__ float As[NBM * LDS_LD];
    __shared__ float Bs[NT * 32 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * NBM;
    const float* A = g.A;
    const float* B = g.B;
    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const bool a_vec = (g.lda & 3) == 0 && (((uintptr_t)A) & 15) == 0;
    const bool b_vec = (g.ldb & 3) == 0 && (((uintptr_t)B) & 15) == 0;
    float ra[4][4], rb[NT][4];
    ImRow ir[4]; ImCol ic;
    if (IM) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ir[h] = im_row(g, m0 + ((tid + 256 * h) >> 3));
        ic = im_col(g, (tid & 7) * 4);
    }
    auto gload = [&](int k0) {
        const int k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) { /* A: 128 rows x 32 k = 1024 groups of 4 k */
            if (IM) im_load4(g, A, ir[h], ic, k, ra[h]);
            else load_a4<PATH>(g, A, a_vec, m0 + ((tid + 256 * h) >> 3), k, ra[h]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) load_b4_nk<PATH>(g, B, b_vec, (tid >> 3) + 32 * t, k, rb[t]); /* B: NT x 32 rows x 32 k */
        if (IM) im_advance(g, ic, BK);
    };
    auto lstore = [&]() {
        const int kc = (tid & 7) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = (tid + 256 * h) >> 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) As[row * LDS_LD + kc + i] = ra[h][i];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = (tid >> 3) + 32 * t;
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[row * LDS_LD + kc + i] = rb[t][i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < g.K) gload(k0 + BK);
        const float* ap = As + (wave * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
        const float* bp = Bs + (lane & 31) * LDS_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float av = ap[kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[t * 32 * LDS_LD + kk], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) store_tile(g, acc[t], g.C, g.residual, m0 + wave * 32, 32 * t + (lane & 31), lane);
}

/*
 * Large-shape variant: 128x128 output tile per workgroup, each of the 4 waves owns 64x64 = 2x2 MFMA tiles (4 accumulators),
 * k slabs of 32 double-buffered in LDS: the next slab's global loads are issued before the current slab's 64 MFMAs and
 * written to the other buffer afterwards, one barrier per slab.  Same k-ordered chain per output element as k_gemm_f32.
 */
#define LBM 128
#define LBN 128
template <bool IM, int PATH>
__global__ __launch_bounds__(256) void k_gemm_f32_big(TkGemm g) {
    extern __shared__ float lsm[]; /* [2][LBM*LDS_LD] A, then [2][LBN*LDS_LD] B */
    float* As = lsm;
    float* Bs = lsm + 2 * LBM * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    int64_t oA, oB, oC, oR;
    batch_offsets(g, blockIdx.z, oA, oB, oC, oR);
    const float* A = g.A + oA;
    const float* B = g.b_f16 ? (const float*)((const _Float16*)g.B + oB) : g.B + oB;
    float* C = g.C + oC;
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool a_vec = (g.lda & 3) == 0 && (((uintptr_t)A) & 15) == 0;
    const bool b_vec = (g.ldb & 3) == 0 && (((uintptr_t)B) & (g.b_f16 ? 7 : 15)) == 0;
    float ra[4][4], rb[4][4]; /* register staging of the next slab: 4 float4 per thread per operand */
    ImRow ir[4]; ImCol ic;
    if (IM) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ir[h] = im_row(g, m0 + ((tid + 256 * h) >> 3));
        ic = im_col(g, (tid & 7) * 4);
    }

    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h, row = e >> 3, k = k0 + (e & 7) * 4;
            if (IM) im_load4(g, A, ir[h], ic, k, ra[h]);
            else load_a4<PATH>(g, A, a_vec, m0 + row, k, ra[h]);
            if (PATH == 2 || (PATH == 0 && g.b_kn)) load_b4_kn<PATH>(g, B, b_vec, k0 + (e >> 5), n0 + (e & 31) * 4, rb[h]); /* [K][N]: 4 consecutive n of one k */
            else load_b4_nk<PATH>(g, B, b_vec, n0 + row, k, rb[h]);
        }
        if (IM) im_advance(g, ic, BK);
    };
    auto lstore = [&](int buf) {
        float* as = As + buf * LBM * LDS_LD;
        float* bs = Bs + buf * LBN * LDS_LD;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h;
            const int row = e >> 3, kc = (e & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) as[row * LDS_LD + kc + i] = ra[h][i];
            if (PATH == 2 || (PATH == 0 && g.b_kn)) {
                const int kk = e >> 5, nc = (e & 31) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) bs[(nc + i) * LDS_LD + kk] = rb[h][i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) bs[row * LDS_LD + kc + i] = rb[h][i];
            }
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        const bool more = k0 + BK < g.K;
        if (more) gload(k0 + BK);
        const float* ap = As + buf * LBM * LDS_LD + (wm * 64 + (lane & 31)) * LDS_LD + (lane >> 5);
        const float* bp = Bs + buf * LBN * LDS_LD + (wn * 64 + (lane & 31)) * LDS_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = ap[kk], a1 = ap[32 * LDS_LD + kk], b0 = bp[kk], b1 = bp[32 * LDS_LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const float* R = g.residual ? g.residual + oR : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) store_tile(g, acc[i][j], C, R, m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32 + (lane & 31), lane);
}

/* ------------------------------------------------------------------------------------------
 * Opt-in fast contraction (TkGemm::fast): the same tiles, loaders and epilogue on the f16 matrix pipe.
 * v_mfma_f32_32x32x16_f16 retires 16 k per 8 passes where v_mfma_f32_32x32x2_f32 retires 2 per 16: 16 x the rate, with operands of 11
 * significant bits.  To stay near fp32 every operand is split on its way into LDS: hi = f16(x), lo = f16((x - hi) * 2048) (the scale keeps
 * lo a normal f16 number), so x = hi + lo / 2048 to ~22 bits, and a product is hi.hi + (hi.lo + lo.hi) / 2048 (lo.lo, 2^-22 of it, is
 * dropped): three MFMAs per 16 k into two fp32 accumulators, joined once in the epilogue.  Products of f16 values are exact in fp32; what
 * differs from the exact chain is the ~2^-22 truncation and the order in which the pipe adds 16 products — results agree with the chain to
 * ~1e-6 of their scale, NOT bit for bit, which is why this is never the default and never what a parity test of the exact path runs.
 * Values beyond the f16 range (|x| > 65504) are not representable in hi: the networks of this path stay far below it.
 * LDS rows hold 32 k as halves + 8 of padding (80 B): 16-byte aligned, and the 16 lanes of a ds_read_b128 phase hit 64 distinct banks.
 * ------------------------------------------------------------------------------------------ */
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
#define HLD 40
#define H3_SCALE 2048.0f
__device__ __forceinline__ void h3_split4(const float x[4], _Float16* hi, _Float16* lo) {
    v4h h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 t = (_Float16)x[i];
        h[i] = t;
        l[i] = (_Float16)((x[i] - (float)t) * H3_SCALE);
    }
    *(v4h*)hi = h;
    *(v4h*)lo = l;
}
__device__ __forceinline__ v16f h3_join(const v16f& hh, const v16f& hx) {
    v16f o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = hh[r] + hx[r] * (1.0f / H3_SCALE);
    return o;
}
#define H3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)

/* k_gemm_f32_tall's tiling (128 rows x all columns, N < 96) */
template <bool IM, int NT>
__global__ __launch_bounds__(256) void k_gemm_h3_tall(TkGemm g) {
    __shared__ __attribute__((aligned(16))) _Float16 Ah[NBM * HLD];
    __shared__ __attribute__((aligned(16))) _Float16 Al[NBM * HLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bh[NT * 32 * HLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bl[NT * 32 * HLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * NBM;
    const float* A = g.A;
    const float* B = g.B;
    v16f acc[NT], acx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[t][i] = 0.0f; acx[t][i] = 0.0f; }
    float ra[4][4], rb[NT][4];
    ImRow ir[4]; ImCol ic;
    if (IM) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ir[h] = im_row(g, m0 + ((tid + 256 * h) >> 3));
        ic = im_col(g, (tid & 7) * 4);
    }
    auto gload = [&](int k0) {
        const int k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (IM) im_load4(g, A, ir[h], ic, k, ra[h]);
            else load_a4<1>(g, A, true, m0 + ((tid + 256 * h) >> 3), k, ra[h]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) load_b4_nk<1>(g, B, true, (tid >> 3) + 32 * t, k, rb[t]);
        if (IM) im_advance(g, ic, BK);
    };
    auto lstore = [&]() {
        const int kc = (tid & 7) * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int o = ((tid + 256 * h) >> 3) * HLD + kc;
            h3_split4(ra[h], Ah + o, Al + o);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = ((tid >> 3) + 32 * t) * HLD + kc;
            h3_split4(rb[t], Bh + o, Bl + o);
        }
    };
    gload(0);
    const int ao = (wave * 32 + (lane & 31)) * HLD + (lane >> 5) * 8, bo = (lane & 31) * HLD + (lane >> 5) * 8;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < g.K) gload(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            const v8h ah = *(const v8h*)(Ah + ao + ks), al = *(const v8h*)(Al + ao + ks);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const v8h bh = *(const v8h*)(Bh + bo + t * 32 * HLD + ks), bl = *(const v8h*)(Bl + bo + t * 32 * HLD + ks);
                acc[t] = H3_MFMA(ah, bh, acc[t]);
                acx[t] = H3_MFMA(ah, bl, acx[t]);
                acx[t] = H3_MFMA(al, bh, acx[t]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) store_tile<true>(g, h3_join(acc[t], acx[t]), g.C, g.residual, m0 + wave * 32, 32 * t + (lane & 31), lane);
}

/* k_gemm_f32_big's tiling (128 x 128 per workgroup, 2 x 2 MFMA tiles per wave, double-buffered slabs) */
template <bool IM>
__global__ __launch_bounds__(256, 2) void k_gemm_h3_big(TkGemm g) {
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[]; /* [2 buffers][Ah, Al, Bh, Bl][128 rows][HLD] */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    int64_t oA, oB, oC, oR;
    batch_offsets(g, blockIdx.z, oA, oB, oC, oR);
    const float* A = g.A + oA;
    const float* B = g.B + oB;
    float* C = g.C + oC;
    v16f acc[2][2], acx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; acx[i][j][r] = 0.0f; }
    float ra[4][4], rb[4][4];
    ImRow ir[4]; ImCol ic;
    if (IM) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ir[h] = im_row(g, m0 + ((tid + 256 * h) >> 3));
        ic = im_col(g, (tid & 7) * 4);
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h, row = e >> 3, k = k0 + (e & 7) * 4;
            if (IM) im_load4(g, A, ir[h], ic, k, ra[h]);
            else load_a4<1>(g, A, true, m0 + row, k, ra[h]);
            load_b4_nk<1>(g, B, true, n0 + row, k, rb[h]);
        }
        if (IM) im_advance(g, ic, BK);
    };
    constexpr int PLANE = LBM * HLD; /* LBM == LBN */
    auto lstore = [&](int buf) {
        _Float16* base = hsm + buf * 4 * PLANE;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h, o = (e >> 3) * HLD + (e & 7) * 4;
            h3_split4(ra[h], base + o, base + PLANE + o);
            h3_split4(rb[h], base + 2 * PLANE + o, base + 3 * PLANE + o);
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    const int ao = (wm * 64 + (lane & 31)) * HLD + (lane >> 5) * 8, bo = (wn * 64 + (lane & 31)) * HLD + (lane >> 5) * 8;
    /* (a second slab in flight — requested two iterations before it is split into LDS — measured no faster: 18.7 against 17.8 ms per 32
     * clips; the loop is not waiting on its loads.  profiles/r06_fast_perception.txt) */
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        const bool more = k0 + BK < g.K;
        if (more) gload(k0 + BK);
        const _Float16* base = hsm + buf * 4 * PLANE;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            v8h ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const v8h*)(base + ao + i * 32 * HLD + ks);
                al[i] = *(const v8h*)(base + PLANE + ao + i * 32 * HLD + ks);
                bh[i] = *(const v8h*)(base + 2 * PLANE + bo + i * 32 * HLD + ks);
                bl[i] = *(const v8h*)(base + 3 * PLANE + bo + i * 32 * HLD + ks);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = H3_MFMA(ah[i], bh[j], acc[i][j]);
                    acx[i][j] = H3_MFMA(ah[i], bl[j], acx[i][j]);
                    acx[i][j] = H3_MFMA(al[i], bh[j], acx[i][j]);
                }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const float* R = g.residual ? g.residual + oR : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) store_tile<true>(g, h3_join(acc[i][j], acx[i][j]), C, R, m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32 + (lane & 31), lane);
}
#define H3_BIG_LDS ((size_t)2 * 4 * LBM * HLD * sizeof(_Float16))

/* ------------------------------------------------------------------------------------------
 * Opt-in fused attention for the fast contraction (head_dim 64, many queries per sequence: the Whisper encoder): Q K^T, softmax and P V in
 * one kernel, nothing of the [Tq][Tk] score matrix ever in HBM (the exact path writes it, normalises it in place and reads it back: 8 of the
 * fast ASR's 15.6 ms per 32 clips, profiles/r06_fast_perception.txt).  Online softmax (running maximum and sum per query, the output rescaled
 * when the maximum moves) — a different association than the exact path's two passes, inside the fast path's ~1e-6.
 * A workgroup takes 128 queries of one (sequence, head) — 32 per wave — and walks the keys 32 at a time:
 *   S^T = K Q^T   (keys x queries; A = the key block from LDS, B = the wave's queries, split once, in registers): the accumulator layout then
 *                 gives every lane ONE query (column lane % 32) and 16 of the block's 32 keys — the row maximum / sum are 16 registers and
 *                 one exchange with lane ^ 32, and the rescale factor is a per-lane scalar;
 *   O^T += V^T P^T (dims x queries; B = the lane's 16 probabilities, split in registers — the MFMA's k slots are filled in the order the
 *                 accumulator holds the keys; A = V^T from LDS, staged transposed with its key columns in that same order).
 * Every operand is split into two f16 halves as in k_gemm_h3_*: three MFMAs per 16 k.
 * ------------------------------------------------------------------------------------------ */
#define FA_KLD 72 /* halves per K row in LDS: 64 dims + 8 */
#define FA_VLD 40 /* halves per V^T row: 32 keys + 8 */
__global__ __launch_bounds__(256, 2) void k_attention_h3(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                                                      int Tq, int Tk, int64_t q_bstride, int64_t kv_bstride, int d, float scale) {
    __shared__ __attribute__((aligned(16))) _Float16 Kh[32 * FA_KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Kl[32 * FA_KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[64 * FA_VLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[64 * FA_VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, n = lane & 31;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * 128 + wave * 32 + n;
    const float* qp = q + (int64_t)b * q_bstride + (int64_t)head * 64;
    const float* kp = k + (int64_t)b * kv_bstride + (int64_t)head * 64;
    const float* vp = v + (int64_t)b * kv_bstride + (int64_t)head * 64;
    /* the lane's query row as the B operand of S^T: dims 16 ks + 8 g + [0, 8), scaled, split once */
    v8h qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float x[8];
        ld4_if(qrow < Tq, qp + (int64_t)qrow * d + ks * 16 + g * 8, x);
        ld4_if(qrow < Tq, qp + (int64_t)qrow * d + ks * 16 + g * 8 + 4, x + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = x[i] * scale;
            const _Float16 hi = (_Float16)t;
            qh[ks][i] = hi;
            ql[ks][i] = (_Float16)((t - (float)hi) * H3_SCALE);
        }
    }
    v16f o_hh[2], o_hx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o_hh[t][r] = 0.0f; o_hx[t][r] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;
    /* staging.  K: 32 keys x 64 dims = 512 groups of four floats, thread -> (key = e / 16, dims 4 (e % 16) ..) for e = tid, tid + 256.
     * V goes into LDS TRANSPOSED ([dim][key], what the A operand of O^T = V^T P^T reads) with its key columns in the order the S^T accumulator
     * holds the keys: column c = 16 s + 8 g + j  <->  key 16 s + 4 g + (j % 4) + 8 (j / 4).  Thread -> (dim = tid % 64, column group tid / 64):
     * its eight keys are eight loads that a wave coalesces over the dims, and ONE 16-byte LDS store per half-plane (2-byte transposing
     * stores from a key-major thread mapping measured the same: the staging is not what a block waits for) */
    float rk[2][4], rv[8];
    const int vd = tid & 63, vc = tid >> 6; /* column group vc = 2 s + g */
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, key = k0 + (e >> 4), dg = (e & 15) * 4;
            ld4_if(key < Tk, kp + (int64_t)key * d + dg, rk[i]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = k0 + 16 * (vc >> 1) + 4 * (vc & 1) + (j & 3) + 8 * (j >> 2);
            rv[j] = key < Tk ? vp[(int64_t)key * d + vd] : 0.0f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, key = e >> 4, dg = (e & 15) * 4;
            h3_split4(rk[i], Kh + key * FA_KLD + dg, Kl + key * FA_KLD + dg);
        }
        v8h hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 t = (_Float16)rv[j];
            hi[j] = t;
            lo[j] = (_Float16)((rv[j] - (float)t) * H3_SCALE);
        }
        *(v8h*)(Vh + vd * FA_VLD + vc * 8) = hi;
        *(v8h*)(Vl + vd * FA_VLD + vc * 8) = lo;
    };
    const float L2E = 1.4426950408889634f;
    gload(0);
    for (int k0 = 0; k0 < Tk; k0 += 32) {
        __syncthreads(); /* everybody is done with the previous block's K / V */
        lstore();
        __syncthreads();
        if (k0 + 32 < Tk) gload(k0 + 32);
        v16f s_hh, s_hx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_hh[r] = 0.0f; s_hx[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const v8h ah = *(const v8h*)(Kh + n * FA_KLD + ks * 16 + g * 8), al = *(const v8h*)(Kl + n * FA_KLD + ks * 16 + g * 8);
            s_hh = H3_MFMA(ah, qh[ks], s_hh);
            s_hx = H3_MFMA(ah, ql[ks], s_hx);
            s_hx = H3_MFMA(al, qh[ks], s_hx);
        }
        float st[16], mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + 4 * g + (r & 3) + 8 * (r >> 2);
            st[r] = key < Tk ? s_hh[r] + s_hx[r] * (1.0f / H3_SCALE) : -INFINITY;
            mloc = fmaxf(mloc, st[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc); /* finite: every block holds at least one key */
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L2E);
        m_run = m_new;
        float psum = 0.0f;
        v8h ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f((st[r] - m_new) * L2E);
            psum += pr;
            const _Float16 hi = (_Float16)pr;
            ph[r >> 3][r & 7] = hi;
            pl[r >> 3][r & 7] = (_Float16)((pr - (float)hi) * H3_SCALE);
        }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) { /* wave-uniform: after the first blocks a query's maximum seldom moves, and the 64 accumulator values stay put */
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { o_hh[t][r] *= alpha; o_hx[t][r] *= alpha; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const v8h vh = *(const v8h*)(Vh + (t * 32 + n) * FA_VLD + sp * 16 + g * 8), vl = *(const v8h*)(Vl + (t * 32 + n) * FA_VLD + sp * 16 + g * 8);
                o_hh[t] = H3_MFMA(vh, ph[sp], o_hh[t]);
                o_hx[t] = H3_MFMA(vh, pl[sp], o_hx[t]);
                o_hx[t] = H3_MFMA(vl, ph[sp], o_hx[t]);
            }
    }
    const float l = l_run + __shfl_xor(l_run, 32);
    if (qrow >= Tq) return;
    const float inv = __builtin_amdgcn_rcpf(l);
    float* op = out + (int64_t)b * q_bstride + (int64_t)qrow * d + (int64_t)head * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { /* rows (dims) t 32 + 4 g + 8 rr + [0, 4): one 16-byte store */
            float4 w;
            w.x = (o_hh[t][4 * rr + 0] + o_hx[t][4 * rr + 0] * (1.0f / H3_SCALE)) * inv;
            w.y = (o_hh[t][4 * rr + 1] + o_hx[t][4 * rr + 1] * (1.0f / H3_SCALE)) * inv;
            w.z = (o_hh[t][4 * rr + 2] + o_hx[t][4 * rr + 2] * (1.0f / H3_SCALE)) * inv;
            w.w = (o_hh[t][4 * rr + 3] + o_hx[t][4 * rr + 3] * (1.0f / H3_SCALE)) * inv;
            *(float4*)(op + t * 32 + 4 * g + 8 * rr) = w;
        }
}

bool tk_launch_attention_h3(const float* q, const float* k, const float* v, float* out, int B, int nh, int Tq, int Tk, int64_t q_bstride, int64_t kv_bstride, int d,
                            float scale, hipStream_t s) {
    if (d != nh * 64 || (d & 3) || Tq < 1 || Tk < 1 || ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) || (q_bstride & 3) || (kv_bstride & 3)) return false;
    hipLaunchKernelGGL(k_attention_h3, dim3((Tq + 127) / 128, nh, B), dim3(256), 0, s, q, k, v, out, Tq, Tk, q_bstride, kv_bstride, d, scale);
    return true;
}

/* > 64 KiB of dynamic LDS is an opt-in HIP keeps per (function, device): one flag per device, set once under a lock (several host
 * threads drive the detector / ASR / VAD streams, possibly on different GPUs).  Callers that capture launches into a hipGraph (the LLM's
 * f16-weight passes) call tk_nn_prepare_device() first so that no attribute is ever set inside a capture. */
static std::atomic<bool> g_big_opted[64];
static std::mutex g_big_mu;
bool tk_nn_prepare_device() {
    const size_t lds = (size_t)2 * (LBM + LBN) * LDS_LD * sizeof(float);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (g_big_opted[dev].load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lk(g_big_mu);
    const void* fns[4] = {(const void*)k_gemm_f32_big<false, 0>, (const void*)k_gemm_f32_big<false, 1>, (const void*)k_gemm_f32_big<false, 2>,
                          (const void*)k_gemm_f32_big<true, 1>};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    const void* h3[2] = {(const void*)k_gemm_h3_big<false>, (const void*)k_gemm_h3_big<true>};
    for (const void* f : h3)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H3_BIG_LDS) != hipSuccess) return false;
    g_big_opted[dev].store(true, std::memory_order_release);
    return true;
}

/* the conditions of the one-load-per-group paths (PATH 1 / 2): f32, every base 16-byte aligned, every pitch and batch offset a multiple
 * of 4 floats, and the extent along the loaded vector (K; N too for [K][N] weights) a multiple of 4 */
static bool tk_gemm_fast(const TkGemm& g) {
    auto m4 = [](int64_t v) { return (v & 3) == 0; };
    if (g.b_f16 || !m4(g.K) || !m4(g.ldb) || (((uintptr_t)g.B) & 15)) return false;
    if (g.im_C == 0 && (!m4(g.lda) || (((uintptr_t)g.A) & 15))) return false;
    if (g.b_kn && !m4(g.N)) return false;
    if (g.batch > 1 && (!m4(g.sA) || !m4(g.sB))) return false;
    if (g.batch_inner > 0 && (!m4(g.sA2) || !m4(g.sB2))) return false;
    return true;
}

void tk_launch_gemm(const TkGemm& g, hipStream_t s) {
    const int nz = g.batch > 0 ? g.batch : 1;
    const bool im = g.im_C > 0; /* validated by tk_gemm_im2col_ok(): aligned [N][K] f32 weights, C % 4 == 0, no batch */
    const int path = tk_gemm_fast(g) ? (g.b_kn ? 2 : 1) : 0;
#define TK_GEMM_PATHS(KERNEL, GRID, LDS)                                                                      \
    do {                                                                                                      \
        if (im) hipLaunchKernelGGL((KERNEL<true, 1>), GRID, dim3(256), LDS, s, g);                            \
        else if (path == 1) hipLaunchKernelGGL((KERNEL<false, 1>), GRID, dim3(256), LDS, s, g);               \
        else if (path == 2) hipLaunchKernelGGL((KERNEL<false, 2>), GRID, dim3(256), LDS, s, g);               \
        else hipLaunchKernelGGL((KERNEL<false, 0>), GRID, dim3(256), LDS, s, g);                              \
    } while (0)
    /* TkGemm::fast: the split-f16 kernels where the operands meet the one-load-per-group conditions ([N][K] f32 weights); every other
     * shape keeps the exact chain.  Which kernel runs depends on N and the batch layout only, NEVER on M: a row's result must not depend
     * on how many other rows (frames, utterances) share its launch — the two split kernels evaluate an element identically (the same 16-k
     * MFMA steps from k = 0), the exact ones differ from them, so a choice by M would make a fast handle's result depend on its batch. */
    const bool fast = g.fast != 0 && !g.b_kn && (im || path == 1);
    if (fast && g.N >= 96) {
        (void)tk_nn_prepare_device();
        const dim3 grid((g.N + LBN - 1) / LBN, (g.M + LBM - 1) / LBM, nz);
        if (im) hipLaunchKernelGGL((k_gemm_h3_big<true>), grid, dim3(256), H3_BIG_LDS, s, g);
        else hipLaunchKernelGGL((k_gemm_h3_big<false>), grid, dim3(256), H3_BIG_LDS, s, g);
        return;
    }
    if (fast && nz == 1 && g.batch_inner == 0) { /* N < 96 */
        const dim3 grid(1, (g.M + NBM - 1) / NBM, 1);
        const int nt = (g.N + 31) / 32;
#define TK_TALL_H3(NTV)                                                                                       \
    do {                                                                                                      \
        if (im) hipLaunchKernelGGL((k_gemm_h3_tall<true, NTV>), grid, dim3(256), 0, s, g);                    \
        else hipLaunchKernelGGL((k_gemm_h3_tall<false, NTV>), grid, dim3(256), 0, s, g);                      \
    } while (0)
        if (nt == 1) TK_TALL_H3(1); else if (nt == 2) TK_TALL_H3(2); else TK_TALL_H3(3);
#undef TK_TALL_H3
        return;
    }
    if (g.M >= 256 && g.N >= 96) {
        const size_t lds = (size_t)2 * (LBM + LBN) * LDS_LD * sizeof(float);
        (void)tk_nn_prepare_device();
        const dim3 grid((g.N + LBN - 1) / LBN, (g.M + LBM - 1) / LBM, nz);
        TK_GEMM_PATHS(k_gemm_f32_big, grid, lds);
        return;
    }
    if (g.N < 96 && g.M >= 4 * NBM && !g.b_kn && !g.b_f16 && nz == 1 && g.batch_inner == 0) { /* narrow and tall: every column in one workgroup */
        const dim3 grid(1, (g.M + NBM - 1) / NBM, 1);
        const int nt = (g.N + 31) / 32;
#define TK_TALL(NTV)                                                                                          \
    do {                                                                                                      \
        if (im) hipLaunchKernelGGL((k_gemm_f32_tall<true, 1, NTV>), grid, dim3(256), 0, s, g);                \
        else if (path == 1) hipLaunchKernelGGL((k_gemm_f32_tall<false, 1, NTV>), grid, dim3(256), 0, s, g);   \
        else hipLaunchKernelGGL((k_gemm_f32_tall<false, 0, NTV>), grid, dim3(256), 0, s, g);                  \
    } while (0)
        if (nt == 1) TK_TALL(1); else if (nt == 2) TK_TALL(2); else TK_TALL(3);
#undef TK_TALL
        return;
    }
    const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nz);
    TK_GEMM_PATHS(k_gemm_f32, grid, 0);
#undef TK_GEMM_PATHS
}

bool tk_gemm_im2col_ok(const float* x, int C, int ldx, const float* w) {
    return C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
}

/*
 * The detector's stem (3 input channels: no 16-byte groups for the implicit path, and its column matrix would be 9x the image): one thread
 * per output pixel evaluates all N output channels as the SAME chain the GEMM evaluates — acc = fma(a[k], w[n][k], acc) over
 * k = (ky kw + kx) C + c ascending from +0, padding taps taking part as a = 0 — then the GEMM's epilogue.  Weights sit in LDS.
 */
template <int N, int C, int KW>
__global__ __launch_bounds__(256) void k_conv_stem(const float* x, int H, int W, int ldx, const float* w, const float* bias, int act, int stride, int pad, int Ho,
                                                   int Wo, uint32_t total, float* y, int ldy) {
    constexpr int K = KW * KW * C;
    __shared__ __attribute__((aligned(16))) float ws[K * N]; /* [k][n]: a wave reads one k of all n as 16-byte broadcasts */
    for (int i = threadIdx.x; i < K * N; i += 256) ws[i] = w[(i % N) * K + i / N];
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    const uint32_t ox = t % (uint32_t)Wo, r = t / (uint32_t)Wo, oy = r % (uint32_t)Ho, b = r / (uint32_t)Ho;
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < KW; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int iy = (int)(oy * stride) + ky - pad, ix = (int)(ox * stride) + kx - pad;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* p = x + (((int64_t)b * H + (in ? iy : 0)) * W + (in ? ix : 0)) * ldx;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float a = in ? p[c] : 0.0f;
                const float4* wk = (const float4*)(ws + ((ky * KW + kx) * C + c) * N);
#pragma unroll
                for (int n = 0; n < N; n += 4) {
                    const float4 q = wk[n >> 2];
                    acc[n] = __builtin_fmaf(a, q.x, acc[n]); acc[n + 1] = __builtin_fmaf(a, q.y, acc[n + 1]);
                    acc[n + 2] = __builtin_fmaf(a, q.z, acc[n + 2]); acc[n + 3] = __builtin_fmaf(a, q.w, acc[n + 3]);
                }
            }
        }
    float4* o = (float4*)(y + (int64_t)t * ldy); /* ldy % 4 == 0 and y 16-byte aligned: checked by the launcher */
#pragma unroll
    for (int n = 0; n < N; n += 4) {
        float4 v;
        v.x = apply_act(acc[n] + (bias ? bias[n] : 0.0f), act); v.y = apply_act(acc[n + 1] + (bias ? bias[n + 1] : 0.0f), act);
        v.z = apply_act(acc[n + 2] + (bias ? bias[n + 2] : 0.0f), act); v.w = apply_act(acc[n + 3] + (bias ? bias[n + 3] : 0.0f), act);
        o[n >> 2] = v;
    }
}

bool tk_launch_conv_stem(const float* x, int B, int H, int W, int C, int ldx, const float* w, const float* bias, int act, int N, int k, int stride, int pad, float* y,
                         int ldy, hipStream_t s) {
    if (!(N == 16 && C == 3 && k == 3) || (ldy & 3) || (((uintptr_t)y) & 15)) return false;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)B * Ho * Wo;
    if (total <= 0 || total >= ((int64_t)1 << 31)) return false;
    hipLaunchKernelGGL((k_conv_stem<16, 3, 3>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, H, W, ldx, w, bias, act, stride, pad, Ho, Wo, (uint32_t)total, y, ldy);
    return true;
}

__global__ void k_im2col(const float* x, int B, int H, int W, int C, int ldx, int kh, int kw, int stride, int pad, int Ho, int Wo, float* col) {
    const int64_t K = (int64_t)kh * kw * C;
    const int64_t total = (int64_t)B * Ho * Wo * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / K;
        const int k = (int)(i % K);
        const int c = k % C, kx = (k / C) % kw, ky = k / (C * kw);
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        float v = 0.0f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * H + iy) * W + ix) * ldx + c];
        col[i] = v;
    }
}

/* the same matrix, four channels per thread as one 16-byte load and store, 32-bit index arithmetic (the element-wise kernel above spends
 * ~250 instructions per element on 64-bit divisions: 0.4 TB/s; this one streams).  Needs C % 4 == 0, ldx % 4 == 0 and < 2^31 vectors. */
__global__ void k_im2col_v4(const float* x, int H, int W, int C4, int ldx, int kh, int kw, int stride, int pad, int Ho, int Wo, uint32_t total4, float* col) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total4) return;
    const uint32_t taps = (uint32_t)(kh * kw);
    const uint32_t c4 = t % (uint32_t)C4, q = t / (uint32_t)C4;
    const uint32_t tap = q % taps, pix = q / taps;
    const uint32_t kx = tap % (uint32_t)kw, ky = tap / (uint32_t)kw;
    const uint32_t ox = pix % (uint32_t)Wo, r = pix / (uint32_t)Wo;
    const uint32_t oy = r % (uint32_t)Ho, b = r / (uint32_t)Ho;
    const int iy = (int)(oy * stride + ky) - pad, ix = (int)(ox * stride + kx) - pad;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const float4*)(x + (((int64_t)b * H + iy) * W + ix) * ldx + 4 * c4);
    *(float4*)(col + (int64_t)t * 4) = v;
}

void tk_launch_im2col(const float* x, int B, int H, int W, int C, int ldx, int kh, int kw, int stride, int pad, float* col, hipStream_t s) {
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const int64_t total = (int64_t)B * Ho * Wo * kh * kw * C;
    if ((C & 3) == 0 && (ldx & 3) == 0 && total / 4 < ((int64_t)1 << 31) && (((uintptr_t)x | (uintptr_t)col) & 15) == 0) {
        const uint32_t total4 = (uint32_t)(total / 4);
        hipLaunchKernelGGL(k_im2col_v4, dim3((total4 + 255u) / 256u), dim3(256), 0, s, x, H, W, C / 4, ldx, kh, kw, stride, pad, Ho, Wo, total4, col);
        return;
    }
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_im2col, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, C, ldx, kh, kw, stride, pad, Ho, Wo, col);
}

/* 5x5 max pool, stride 1, pad 2 (SPPF); padding never wins (-inf) */
__global__ void k_maxpool5(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float m = -INFINITY;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx) {
                const int iy = oy + dy, ix = ox + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) m = tk_fmaxf(m, x[(((int64_t)b * H + iy) * W + ix) * ldx + c]);
            }
        y[pix * ldy + c] = m;
    }
}
void tk_launch_maxpool5(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(k_maxpool5, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, B, H, W, C, ldx, y, ldy);
}

__global__ void k_upsample2x(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy) {
    const int64_t total = (int64_t)B * 2 * H * 2 * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int ox = (int)(pix % (2 * W)), oy = (int)((pix / (2 * W)) % (2 * H)), b = (int)(pix / ((int64_t)4 * W * H));
        y[pix * ldy + c] = x[(((int64_t)b * H + oy / 2) * W + ox / 2) * ldx + c];
    }
}
void tk_launch_upsample2x(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy, hipStream_t s) {
    const int64_t total = (int64_t)B * 4 * H * W * C;
    hipLaunchKernelGGL(k_upsample2x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, B, H, W, C, ldx, y, ldy);
}

__global__ void k_copy_cols(const float* x, int64_t rows, int C, int ldx, float* y, int ldy) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        y[(i / C) * ldy + (i % C)] = x[(i / C) * ldx + (i % C)];
}
void tk_launch_copy_cols(const float* x, int rows, int C, int ldx, float* y, int ldy, hipStream_t s) {
    const int64_t total = (int64_t)rows * C;
    hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (int64_t)rows, C, ldx, y, ldy);
}

__device__ __forceinline__ float nn_block_sum256(float v, float* red) {
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return r;
}

/* y = ((x - mean) * rstd) * w + b ; sums in the canonical 256-partial order */
/* img (optional): the rows are ALSO written as the tiled GEMM's operand image (csrc/nn/tk_gemm_tiled.h: element (row, k = 16 j + 4 t + g) at
 * [row / 16][j][g][row % 16][t]) — the linear layer behind a norm then needs no pack launch */
__global__ __launch_bounds__(256) void k_layernorm(const float* x, int D, const float* w, const float* b, float eps, float* y, float* img) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * D;
    float s = 0.0f;
    for (int i = threadIdx.x; i < D; i += 256) s = s + xr[i];
    const float mean = tk_divf(nn_block_sum256(s, red), (float)D);
    float q = 0.0f;
    for (int i = threadIdx.x; i < D; i += 256) { const float d = xr[i] - mean; q = tk_fmaf(d, d, q); }
    const float var = tk_divf(nn_block_sum256(q, red), (float)D);
    const float rstd = tk_divf(1.0f, tk_sqrtf(var + eps));
    float* ir = img ? img + (r >> 4) * 16 * (int64_t)D + (r & 15) * 4 : nullptr;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float v = ((xr[i] - mean) * rstd) * w[i] + b[i];
        y[r * D + i] = v;
        if (ir) ir[(int64_t)(i >> 4) * 256 + (i & 3) * 64 + ((i & 15) >> 2)] = v;
    }
}
void tk_launch_layernorm(const float* x, int rows, int D, const float* w, const float* b, float eps, float* y, hipStream_t s, float* img) {
    hipLaunchKernelGGL(k_layernorm, dim3(rows), dim3(256), 0, s, x, D, w, b, eps, y, img);
}

/* in-place row softmax over the first `cols` entries of every row (row pitch ld) */
__global__ __launch_bounds__(256) void k_softmax_rows(float* x, int cols, int ld) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    float* xr = x + r * ld;
    const int n = cols;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = tk_fmaxf(m, xr[i]);
    for (int s = 32; s >= 1; s >>= 1) m = tk_fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = tk_fmaxf(tk_fmaxf(red[0], red[1]), tk_fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) { const float e = tk_expf(xr[i] - m); xr[i] = e; s = s + e; }
    const float tot = nn_block_sum256(s, red);
    for (int i = threadIdx.x; i < cols; i += 256) xr[i] = i < n ? tk_divf(xr[i], tot) : 0.0f;
}
/* rows of up to 2048 columns: the row is read once into registers (thread j owns elements j, j + 256, ...), and every later step of
 * k_softmax_rows runs on those registers in the same order — the same values bit for bit with one read and one write per element
 * instead of three reads and two writes (the Whisper encoder's score rows: 8.6 GB -> 3.5 GB per layer and call) */
__global__ __launch_bounds__(256) void k_softmax_rows_reg(float* x, int cols, int ld) {
    __shared__ float red[4];
    float* xr = x + (int64_t)blockIdx.x * ld;
    float v[8];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = threadIdx.x + 256 * u;
        v[u] = i < cols ? xr[i] : -INFINITY;
        if (i < cols) m = tk_fmaxf(m, v[u]);
    }
    for (int s = 32; s >= 1; s >>= 1) m = tk_fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = tk_fmaxf(tk_fmaxf(red[0], red[1]), tk_fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (threadIdx.x + 256 * u < cols) { v[u] = tk_expf(v[u] - m); s = s + v[u]; }
    const float tot = nn_block_sum256(s, red);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (threadIdx.x + 256 * u < cols) xr[threadIdx.x + 256 * u] = tk_divf(v[u], tot);
}

/* rows of up to 2048 columns with 16-byte aligned rows (cols and ld multiples of 4): ONE WAVE per row, no barrier and no LDS.  Lane l reads
 * the 16-byte groups l, l + 64, ...: element e = 4 l + c + 256 u is exactly an element of the canonical partial t = e % 256 = 4 l + c, so
 * the lane holds partials 4 l .. 4 l + 3 and adds to each in ascending u — the order of k_softmax_rows.  The canonical tree (a 64-wide
 * xor butterfly inside each quarter t / 64, then ((q0 + q1) + q2) + q3) becomes: xor 8, 4, 2, 1 across the 16 lanes of a quarter (t ^ 32,
 * 16, 8, 4), then components c ^ 2 and c ^ 1 inside the lane (t ^ 2, 1).  Same pairs at every level: bit-identical. */
__global__ __launch_bounds__(256) void k_softmax_rows_wave(float* x, int64_t rows, int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float4* xr = (float4*)(x + r * ld);
    const int n4 = cols >> 2;
    float4 v[8];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = lane + 64 * u;
        if (i < n4) {
            v[u] = xr[i];
            m = tk_fmaxf(tk_fmaxf(tk_fmaxf(tk_fmaxf(m, v[u].x), v[u].y), v[u].z), v[u].w);
        }
    }
    for (int s = 32; s >= 1; s >>= 1) m = tk_fmaxf(m, __shfl_xor(m, s, 64));
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (lane + 64 * u < n4) {
            v[u].x = tk_expf(v[u].x - m); v[u].y = tk_expf(v[u].y - m); v[u].z = tk_expf(v[u].z - m); v[u].w = tk_expf(v[u].w - m);
            p0 = p0 + v[u].x; p1 = p1 + v[u].y; p2 = p2 + v[u].z; p3 = p3 + v[u].w;
        }
    for (int s = 8; s >= 1; s >>= 1) { /* t ^ 32, 16, 8, 4 */
        p0 = p0 + __shfl_xor(p0, s, 64); p1 = p1 + __shfl_xor(p1, s, 64); p2 = p2 + __shfl_xor(p2, s, 64); p3 = p3 + __shfl_xor(p3, s, 64);
    }
    const float q = (p0 + p2) + (p1 + p3); /* t ^ 2: (c0 + c2), (c1 + c3); t ^ 1: their sum */
    const float tot = ((__shfl(q, 0, 64) + __shfl(q, 16, 64)) + __shfl(q, 32, 64)) + __shfl(q, 48, 64);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = lane + 64 * u;
        if (i < n4) {
            float4 o;
            o.x = tk_divf(v[u].x, tot); o.y = tk_divf(v[u].y, tot); o.z = tk_divf(v[u].z, tot); o.w = tk_divf(v[u].w, tot);
            xr[i] = o;
        }
    }
}

void tk_launch_softmax_rows(float* x, int rows, int cols, int ld, hipStream_t s) {
    if (cols <= 2048 && (cols & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0 && rows >= 1024)
        hipLaunchKernelGGL(k_softmax_rows_wave, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, (int64_t)rows, cols, ld);
    else if (cols <= 2048) hipLaunchKernelGGL(k_softmax_rows_reg, dim3(rows), dim3(256), 0, s, x, cols, ld);
    else hipLaunchKernelGGL(k_softmax_rows, dim3(rows), dim3(256), 0, s, x, cols, ld);
}

__global__ void k_add_rows(float* x, const float* add, int64_t rows, int D, int add_rows) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        x[i] = x[i] + add[(r % add_rows) * D + (i % D)];
    }
}
void tk_launch_add_rows(float* x, const float* add, int rows, int D, int add_rows, hipStream_t s) {
    const int64_t total = (int64_t)rows * D;
    hipLaunchKernelGGL(k_add_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, add, (int64_t)rows, D, add_rows);
}

__global__ void k_im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, int To, float* col) {
    const int64_t K = (int64_t)kw * C;
    const int64_t total = (int64_t)B * To * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / K;
        const int k = (int)(i % K);
        const int c = k % C, kx = k / C;
        const int t = (int)(row % To), b = (int)(row / To);
        const int it = t * stride + kx - pad;
        col[i] = (it >= 0 && it < T) ? x[((int64_t)b * T + it) * ldx + c] : 0.0f;
    }
}
void tk_launch_im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, float* col, hipStream_t s) {
    const int To = (T + 2 * pad - kw) / stride + 1;
    const int64_t total = (int64_t)B * To * kw * C;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_im2col1d, dim3((unsigned)blocks), dim3(256), 0, s, x, B, T, C, ldx, kw, stride, pad, To, col);
}

__global__ void k_embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int D, float* out) {
    const int r = blockIdx.x;
    const float* t = table + (int64_t)idx[r] * D;
    const float* p = pos + (int64_t)pos_idx[r] * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) out[(int64_t)r * D + i] = t[i] + p[i];
}
void tk_launch_embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int rows, int D, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_embed_rows, dim3(rows), dim3(128), 0, s, table, pos, idx, pos_idx, D, out);
}

/* first index of the row maximum */
__global__ __launch_bounds__(1024) void k_argmax_rows(const float* x, int cols, int ld, int32_t* out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int r = blockIdx.x, t = threadIdx.x;
    const float* xr = x + (int64_t)r * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = t; i < cols; i += 1024) {
        const float v = xr[i];
        if (v > best) { best = v; idx = i; }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s, 64);
        const int oi = __shfl_xor(idx, s, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[r] = idx;
    }
}
/* The ASR decoder's token pick when whisper.cpp's decoding policy is on (the reference's wrapper sets temperature_inc 0.2, entropy_thold 2.4,
 * logprob_thold -1.0: src/audio/tk_asr_whisper.c:126-138): per row, the token — first index of the maximum at temperature 0, else one draw from
 * softmax(l / temperature) by the canonical sampler (common/tk_sample_device.h; candidates = the 64 largest logits) — and its log-probability
 * under softmax(l / temperature) over the WHOLE vocabulary, which the policy's thresholds are taken on:
 *     z_i = (l_i - max) [/ temperature],  S = sum_i exp(z_i) as 1024 chains (chain t: i = t, t + 1024, ... ascending) joined per wave by the
 *     xor butterfly (32 .. 1) and then wave 0 .. 15 in order,  logprob = z_tok - log S.
 * Restated in oracle/tk_oracle_audio.cpp (pick_rows). */
__global__ __launch_bounds__(1024) void k_pick_rows(const float* x, int cols, int ld, int32_t* out, TkPick pk) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ uint32_t sm[256 + 8 + 4 * TK_SAMPLE_MAX_K];
    __shared__ int32_t picked;
    const int r = blockIdx.x, t = threadIdx.x;
    const float* xr = x + (int64_t)r * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = t; i < cols; i += 1024) {
        const float v = xr[i];
        if (v > best) { best = v; idx = i; }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s, 64);
        const int oi = __shfl_xor(idx, s, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        bv[0] = best; bi[0] = idx;
    }
    __syncthreads();
    const float mx = bv[0];
    int tok = bi[0];
    __syncthreads();
    if (pk.temp > 0.0f) {
        TkSampleRow sp{};
        sp.temp = pk.temp; sp.top_p = 1.0f; sp.min_p = 0.0f; sp.top_k = 0; sp.seed = pk.seed; sp.counter = pk.counter0 + (uint32_t)r;
        sample_row(xr, cols, nullptr, sp, &picked, sm);
        __syncthreads();
        tok = picked;
    }
    if (t == 0) out[r] = tok;
    if (pk.logprob) {
        float ssum = 0.0f;
        for (int i = t; i < cols; i += 1024) ssum = ssum + tk_expf(pk.temp > 0.0f ? tk_divf(xr[i] - mx, pk.temp) : xr[i] - mx);
        for (int s = 32; s >= 1; s >>= 1) ssum = ssum + __shfl_xor(ssum, s, 64);
        __syncthreads();
        if ((t & 63) == 0) bv[t >> 6] = ssum;
        __syncthreads();
        if (t == 0) {
            float S = bv[0];
            for (int w = 1; w < 16; ++w) S = S + bv[w];
            const float z = pk.temp > 0.0f ? tk_divf(xr[tok] - mx, pk.temp) : xr[tok] - mx;
            pk.logprob[r] = z - tk_logf(S);
        }
    }
}
void tk_launch_pick_rows(const float* x, int rows, int cols, int ld, int32_t* out, const TkPick& pk, hipStream_t s) {
    hipLaunchKernelGGL(k_pick_rows, dim3(rows), dim3(1024), 0, s, x, cols, ld, out, pk);
}

void tk_launch_argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_argmax_rows, dim3(rows), dim3(1024), 0, s, x, cols, ld, out);
}

/* ---- whisper.cpp's logit filters and decode bookkeeping on the device (one workgroup per row; restated sequentially in
 * oracle/tk_oracle_audio.cpp: orc_whisper_filter_pick).  Canonical arithmetic, with z_i = l_i (temperature 0) or l_i / temperature:
 *   allowed set A0 = not in the static table, and by the row's state: first token -> no timestamp above beg + tid0; last token a timestamp ->
 *     (the one before too: no timestamp) / (otherwise: nothing below eot); has_ts -> no timestamp below beg + seek_delta / 2
 *   mx = max z over A0,  S = sum over A0 of exp(z_i - mx) as 1024 chains (chain t: i = t, t + 1024, ...) joined per wave by the xor butterfly
 *     (32 .. 1), then waves 0 .. 15 in order,  lp_i = (z_i - mx) - log S
 *   timestamps: m_ts = max lp over A0 above beg, S_ts = sum exp(lp_i - m_ts) the same way, lp_ts = log S_ts + m_ts; text: m_tx = max lp over A0
 *     below beg;  lp_ts > m_tx -> A = A0 without the text tokens, else A = A0
 *   token = first index of the maximum of z over A (temperature 0) or one draw of the canonical sampler over A; its log-probability = lp_token */
__global__ __launch_bounds__(1024) void k_pick_rows_filtered(const float* x, int cols, int ld, int32_t* out, TkPick pk, TkWhFilter f) {
    __shared__ uint32_t allow[2048]; /* vocabularies of at most 65536 tokens */
    __shared__ float bv[16], bw[16];
    __shared__ int bi[16];
    __shared__ uint32_t sm[256 + 8 + 4 * TK_SAMPLE_MAX_K];
    __shared__ int32_t picked;
    const int r = blockIdx.x, t = threadIdx.x;
    int32_t* st = f.state + (int64_t)r * TK_WH_STATE_INTS;
    if (st[6] != 0) { if (t == 0) out[r] = f.eot; return; } /* completed or failed earlier: the row stands still */
    const float* xr = x + (int64_t)r * ld;
    const int n_tok = st[0], has_ts = st[3], seek_delta = st[4];
    const int last_ts = n_tok > 0 ? st[1] : 0;  /* tokens_cur.size() > 0 && tokens_cur.back() is a timestamp */
    const int prev_ts = n_tok < 2 ? 1 : st[2];  /* tokens_cur.size() < 2 || the one before it is */
    const int nwords = (cols + 31) >> 5;
    const int ts_lo = has_ts ? f.beg + seek_delta / 2 : f.beg;               /* timestamps below it would go back in time */
    const int ts_hi = n_tok == 0 && f.tid0 >= 0 ? f.beg + f.tid0 : 0x7fffffff; /* the first timestamp cannot lie beyond max_initial_ts */
    for (int w = t; w < nwords; w += 1024) {
        uint32_t bits = 0;
        for (int b = 0; b < 32; ++b) {
            const int i = 32 * w + b;
            if (i >= cols) break;
            bool ok = f.suppress[i] == 0;
            if (i >= f.beg) ok = ok && i >= ts_lo && i <= ts_hi && !(last_ts && prev_ts);
            if (last_ts && !prev_ts && i < f.eot) ok = false;
            bits |= ok ? (1u << b) : 0u;
        }
        allow[w] = bits;
    }
    __syncthreads();
    auto allowed = [&](int i) { return (allow[i >> 5] >> (i & 31)) & 1u; };
    const bool hot = pk.temp > 0.0f;
    auto zof = [&](int i) { return hot ? tk_divf(xr[i], pk.temp) : xr[i]; };
    /* maxima of z over A0: all, text, timestamps (a workgroup-wide max is order-free) */
    float m_all = -INFINITY, m_tx = -INFINITY, m_ts = -INFINITY;
    for (int i = t; i < cols; i += 1024) {
        if (!allowed(i)) continue;
        const float z = zof(i);
        m_all = tk_fmaxf(m_all, z);
        if (i < f.beg) m_tx = tk_fmaxf(m_tx, z); else m_ts = tk_fmaxf(m_ts, z);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        m_all = tk_fmaxf(m_all, __shfl_xor(m_all, s, 64));
        m_tx = tk_fmaxf(m_tx, __shfl_xor(m_tx, s, 64));
        m_ts = tk_fmaxf(m_ts, __shfl_xor(m_ts, s, 64));
    }
    if ((t & 63) == 0) { bv[t >> 6] = m_all; bw[t >> 6] = m_tx; bi[t >> 6] = __float_as_int(m_ts); }
    __syncthreads();
    for (int w = 0; w < 16; ++w) { m_all = tk_fmaxf(m_all, bv[w]); m_tx = tk_fmaxf(m_tx, bw[w]); m_ts = tk_fmaxf(m_ts, __int_as_float(bi[w])); }
    __syncthreads();
    /* S over A0 */
    float ssum = 0.0f;
    for (int i = t; i < cols; i += 1024)
        if (allowed(i)) ssum = ssum + tk_expf(zof(i) - m_all);
    for (int s = 32; s >= 1; s >>= 1) ssum = ssum + __shfl_xor(ssum, s, 64);
    if ((t & 63) == 0) bv[t >> 6] = ssum;
    __syncthreads();
    float S = bv[0];
    for (int w = 1; w < 16; ++w) S = S + bv[w];
    const float logS = tk_logf(S);
    __syncthreads();
    /* the timestamps' summed probability against the best text token */
    const float lp_mts = (m_ts - m_all) - logS, lp_mtx = (m_tx - m_all) - logS;
    float tsum = 0.0f;
    for (int i = t; i < cols; i += 1024)
        if (i >= f.beg && allowed(i)) tsum = tsum + tk_expf(((zof(i) - m_all) - logS) - lp_mts);
    for (int s = 32; s >= 1; s >>= 1) tsum = tsum + __shfl_xor(tsum, s, 64);
    if ((t & 63) == 0) bv[t >> 6] = tsum;
    __syncthreads();
    float S_ts = bv[0];
    for (int w = 1; w < 16; ++w) S_ts = S_ts + bv[w];
    const float lp_ts = S_ts > 0.0f ? tk_logf(S_ts) + lp_mts : -INFINITY;
    __syncthreads();
    if (lp_ts > lp_mtx) { /* sample a timestamp: the text tokens leave the allowed set */
        for (int w = t; w < nwords; w += 1024) {
            const int lo = 32 * w;
            if (lo + 32 <= f.beg) allow[w] = 0;
            else if (lo < f.beg) allow[w] &= ~((1u << (f.beg - lo)) - 1u);
        }
        __syncthreads();
    }
    /* the pick */
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = t; i < cols; i += 1024) {
        if (!allowed(i)) continue;
        const float v = xr[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s, 64);
        const int oi = __shfl_xor(idx, s, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        bi[0] = idx;
    }
    __syncthreads();
    int tok = bi[0];
    __syncthreads();
    if (hot) {
        TkSampleRow sp{};
        sp.temp = pk.temp; sp.top_p = 1.0f; sp.min_p = 0.0f; sp.top_k = 0; sp.seed = pk.seed; sp.counter = pk.counter0 + (uint32_t)r;
        sample_row(xr, cols, allow, sp, &picked, sm);
        __syncthreads();
        tok = picked;
    }
    if (t == 0) {
        out[r] = tok;
        if (pk.logprob) pk.logprob[r] = (zof(tok) - m_all) - logS;
        /* whisper_full's bookkeeping of the sampled token (i = its index in the sequence) */
        const int i = n_tok;
        int hts = has_ts, sd = seek_delta, rl = st[5], status = 0;
        const int seek_end = st[7];
        if (tok > f.beg) {
            const int sd_new = 2 * (tok - f.beg);
            if (hts && sd > sd_new && rl < i) status = 2; /* "do not allow to go back in time" */
            else { sd = sd_new; rl = i + 1; hts = 1; }
        }
        if (status == 0 && (tok == f.eot || (hts && sd + 100 >= seek_end))) {
            if (rl == 0) {
                if (sd + 100 >= seek_end) rl = i + 1;
                else status = 2;
            }
            if (status == 0) status = 1;
        }
        st[0] = i + 1; st[1] = tok >= f.beg ? 1 : 0; st[2] = last_ts; st[3] = hts; st[4] = sd; st[5] = rl; st[6] = status;
    }
}
void tk_launch_pick_rows_filtered(const float* x, int rows, int cols, int ld, int32_t* out, const TkPick& pk, const TkWhFilter& f, hipStream_t s) {
    hipLaunchKernelGGL(k_pick_rows_filtered, dim3(rows), dim3(1024), 0, s, x, cols, ld, out, pk, f);
}
