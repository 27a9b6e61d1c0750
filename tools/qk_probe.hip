// Developer probe (needs an MI355X): the Whisper encoder's attention GEMMs as the graph issues them (Q.K^T: M = N = 1500, K = 64,
// 32 clips x 6 heads; P.V: M = 1500, N = 64, K = 1500), timed with variations of the output pitch and shape to find what bounds them.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I trackiellm_amd/csrc tools/qk_probe.hip trackiellm_amd/csrc/nn/tk_nn_kernels.hip -o build/qk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "nn/tk_nn_kernels.h"

void run_variants(const TkGemm& g);
static float run(const TkGemm& g, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    tk_launch_gemm(g, 0); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) tk_launch_gemm(g, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const int B = 32, nh = 6, T = 1500, d = 384, hd = 64;
    float *q, *k, *v, *sc, *out;
    const size_t nqk = (size_t)B * T * d, nsc = (size_t)B * nh * 1536 * 1536;
    hipMalloc(&q, nqk * 4); hipMalloc(&k, nqk * 4); hipMalloc(&v, nqk * 4); hipMalloc(&out, nqk * 4); hipMalloc(&sc, nsc * 4);
    std::vector<float> h(nqk);
    for (size_t i = 0; i < nqk; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    hipMemcpy(q, h.data(), nqk * 4, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), nqk * 4, hipMemcpyHostToDevice); hipMemcpy(v, h.data(), nqk * 4, hipMemcpyHostToDevice);
    hipMemset(sc, 0, nsc * 4);
    auto qk = [&](int Tq, int Tk, int ldc, float alpha) {
        TkGemm s{};
        s.A = q; s.B = k; s.C = sc; s.M = Tq; s.N = Tk; s.K = hd; s.lda = d; s.ldb = d; s.ldc = ldc; s.alpha = alpha;
        s.batch = B * nh; s.batch_inner = nh; s.sA = hd; s.sB = hd; s.sC = (int64_t)Tq * ldc;
        s.sA2 = (int64_t)T * d; s.sB2 = (int64_t)T * d; s.sC2 = (int64_t)nh * Tq * ldc;
        return s;
    };
    const double fl = 2.0 * 1500 * 1500 * 64 * B * nh;
    struct { const char* name; int Tq, Tk, ldc; float alpha; } cases[] = {
        {"QK as issued (ldc 1500, alpha 1/8)", 1500, 1500, 1500, 0.125f},
        {"QK ldc 1536", 1500, 1500, 1536, 0.125f},
        {"QK 1408 x 1408 (whole tiles only), ldc 1536", 1408, 1408, 1536, 0.125f},
        {"QK alpha 1", 1500, 1500, 1500, 1.0f},
    };
    for (auto& c : cases) {
        const TkGemm g = qk(c.Tq, c.Tk, c.ldc, c.alpha);
        const float ms = run(g, 5);
        printf("%-58s %8.3f ms  %6.1f TFLOP/s  stores %5.2f TB/s\n", c.name, ms, 2.0 * c.Tq * c.Tk * 64 * B * nh / ms * 1e-9, (double)c.Tq * c.Tk * 4 * B * nh / ms * 1e-9);
    }
    (void)fl;
    { /* the whole attention (Q.K^T, softmax, P.V) over all 32 clips at once vs in chunks of few clips whose scores stay in the 256 MB
         Infinity Cache between the three launches */
        for (int chunk : {32, 8, 4, 2}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                if (rep == 1) hipEventRecord(a, 0);
                for (int c0 = 0; c0 < B; c0 += chunk) {
                    TkGemm s1 = qk(1500, 1500, 1500, 0.125f);
                    s1.A = q + (size_t)c0 * T * d; s1.B = k + (size_t)c0 * T * d; s1.batch = chunk * nh;
                    tk_launch_gemm(s1, 0);
                    tk_launch_softmax_rows(sc, (int64_t)chunk * nh * T, T, T, 0);
                    TkGemm p{};
                    p.A = sc; p.B = v + (size_t)c0 * T * d; p.C = out + (size_t)c0 * T * d; p.M = T; p.N = hd; p.K = T; p.lda = T; p.ldb = d; p.ldc = d; p.b_kn = 1; p.alpha = 1.0f;
                    p.batch = chunk * nh; p.batch_inner = nh; p.sA = (int64_t)T * T; p.sB = hd; p.sC = hd;
                    p.sA2 = (int64_t)nh * T * T; p.sB2 = (int64_t)T * d; p.sC2 = (int64_t)T * d;
                    tk_launch_gemm(p, 0);
                }
            }
            hipEventRecord(b, 0); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            printf("attention of 32 clips in chunks of %2d clips (scores %4.0f MB)  %8.3f ms\n", chunk, (double)chunk * nh * T * T * 4 * 1e-6, ms / 3);
        }
    }
    run_variants(qk(1500, 1500, 1500, 0.125f));
    { /* every (clip, head) writes the SAME 9 MB: the stores stay on chip — is HBM's write path the bound? */
        TkGemm g = qk(1500, 1500, 1500, 0.125f);
        g.sC = 0; g.sC2 = 0;
        const float ms = run(g, 5);
        printf("%-58s %8.3f ms  %6.1f TFLOP/s\n", "QK, all batches onto one 9 MB output", ms, 2.0 * 1500 * 1500 * 64 * B * nh / ms * 1e-9);
    }
    { /* an eighth of the batches: 3456 workgroups, 6.75 per workgroup slot */
        TkGemm g = qk(1500, 1500, 1500, 0.125f);
        g.batch = 4 * nh;
        const float ms = run(g, 5);
        printf("%-58s %8.3f ms  %6.1f TFLOP/s  stores %5.2f TB/s\n", "QK, 4 clips only", ms, 2.0 * 1500 * 1500 * 64 * 4 * nh / ms * 1e-9, 1500.0 * 1500 * 4 * 4 * nh / ms * 1e-9);
    }
    {
        TkGemm p{};
        p.A = sc; p.B = v; p.C = out; p.M = T; p.N = hd; p.K = T; p.lda = T; p.ldb = d; p.ldc = d; p.b_kn = 1; p.alpha = 1.0f;
        p.batch = B * nh; p.batch_inner = nh; p.sA = (int64_t)T * T; p.sB = hd; p.sC = hd;
        p.sA2 = (int64_t)nh * T * T; p.sB2 = (int64_t)T * d; p.sC2 = (int64_t)T * d;
        const float ms = run(p, 5);
        printf("%-58s %8.3f ms  %6.1f TFLOP/s  reads  %5.2f TB/s\n", "PV as issued", ms, 2.0 * T * T * 64 * B * nh / ms * 1e-9, (double)T * T * 4 * B * nh / ms * 1e-9);
    }
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        tk_launch_softmax_rows(sc, (int64_t)B * nh * T, T, T, 0); hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < 5; ++i) tk_launch_softmax_rows(sc, (int64_t)B * nh * T, T, T, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-58s %8.3f ms  read + write %5.2f TB/s\n", "softmax rows", ms, 2.0 * T * T * 4 * B * nh / ms * 1e-9);
    }
    return 0;
}

// ---- variants of the 128 x 128 kernel's structure (copies of its loop with one thing removed each), run by `qk_probe variants`
typedef float v16f_t __attribute__((ext_vector_type(16)));
#define PLD 33
template <int MODE> /* 0: loop as shipped, epilogue alpha only; 1: no stores; 2: no global loads; 3: no MFMAs; 4: float4 row stores through LDS */
__global__ __launch_bounds__(256) void k_var(TkGemm g) {
    extern __shared__ float lsm[];
    float* As = lsm;
    float* Bs = lsm + 2 * 128 * PLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int zo = blockIdx.z / g.batch_inner, zi = blockIdx.z % g.batch_inner;
    const float* A = g.A + zo * g.sA2 + zi * g.sA;
    const float* B = g.B + zo * g.sB2 + zi * g.sB;
    float* C = g.C + zo * g.sC2 + zi * g.sC;
    v16f_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h, row = e >> 3, k = k0 + (e & 7) * 4;
            ra[h] = make_float4(0, 0, 0, 0); rb[h] = make_float4(0, 0, 0, 0);
            if (MODE == 2) { ra[h].x = (float)row; rb[h].y = (float)k; continue; }
            if (m0 + row < g.M) ra[h] = *(const float4*)(A + (int64_t)(m0 + row) * g.lda + k);
            if (n0 + row < g.N) rb[h] = *(const float4*)(B + (int64_t)(n0 + row) * g.ldb + k);
        }
    };
    auto lstore = [&](int buf) {
        float* as = As + buf * 128 * PLD; float* bs = Bs + buf * 128 * PLD;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h, row = e >> 3, kc = (e & 7) * 4;
            as[row * PLD + kc] = ra[h].x; as[row * PLD + kc + 1] = ra[h].y; as[row * PLD + kc + 2] = ra[h].z; as[row * PLD + kc + 3] = ra[h].w;
            bs[row * PLD + kc] = rb[h].x; bs[row * PLD + kc + 1] = rb[h].y; bs[row * PLD + kc + 2] = rb[h].z; bs[row * PLD + kc + 3] = rb[h].w;
        }
    };
    gload(0); lstore(0); __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += 32) {
        const bool more = k0 + 32 < g.K;
        if (more) gload(k0 + 32);
        const float* ap = As + buf * 128 * PLD + (wm * 64 + (lane & 31)) * PLD + (lane >> 5);
        const float* bp = Bs + buf * 128 * PLD + (wn * 64 + (lane & 31)) * PLD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a0 = ap[kk], a1 = ap[32 * PLD + kk], b0 = bp[kk], b1 = bp[32 * PLD + kk];
            if (MODE == 3) { acc[0][0][kk & 15] += a0 * b0; acc[0][1][kk & 15] += a0 * b1; acc[1][0][kk & 15] += a1 * b0; acc[1][1][kk & 15] += a1 * b1; continue; }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (MODE == 4) { /* each wave's 64 x 64 block through its own 16 KiB of LDS, then 16-byte row stores */
        float* sp = lsm + wave * 64 * 65;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sp[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][j][r] * g.alpha;
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4), c = (lane & 15) * 4;
            const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + c;
            if (m < g.M && n + 3 < g.N) {
                float4 v; v.x = sp[row * 65 + c]; v.y = sp[row * 65 + c + 1]; v.z = sp[row * 65 + c + 2]; v.w = sp[row * 65 + c + 3];
                *(float4*)(C + (int64_t)m * g.ldc + n) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[i][j][r] * g.alpha;
                if (MODE == 1) { if (v == 12345.678f) C[(int64_t)m * g.ldc + n] = v; continue; }
                if (m < g.M) C[(int64_t)m * g.ldc + n] = v;
            }
    }
}

template <int MODE>
static float run_var(const TkGemm& g, int iters) {
    const size_t lds = (size_t)4 * 128 * PLD * 4;
    hipFuncSetAttribute((const void*)k_var<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, g.batch);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_var<MODE>, grid, dim3(256), lds, 0, g); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_var<MODE>, grid, dim3(256), lds, 0, g);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}
void run_variants(const TkGemm& g) {
    printf("variant 0 (shipped loop, plain epilogue)   %8.3f ms\n", run_var<0>(g, 5));
    printf("variant 1 (no stores)                      %8.3f ms\n", run_var<1>(g, 5));
    printf("variant 2 (no global loads)                %8.3f ms\n", run_var<2>(g, 5));
    printf("variant 3 (no MFMAs)                       %8.3f ms\n", run_var<3>(g, 5));
    printf("variant 4 (16-byte row stores through LDS) %8.3f ms\n", run_var<4>(g, 5));
}
