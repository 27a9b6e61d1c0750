// Developer probe: do the VALU finishing of one wave and the 32x32x32 i8 MFMAs of its SIMD partner overlap on gfx950?
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD.  Each wave runs `tiles` iterations of
//   [MFMA phase: 16 x v_mfma_i32_32x32x32_i8 (two 8-chains) + 1 x v_mfma_f32_32x32x16_f16] [VALU phase: the 96-instruction finishing of k_gemm32_w4a8]
// in one of these arrangements:
//   mode 0  both waves the same program, started together (lockstep)
//   mode 1  waves 4..7 start with the VALU phase of a dummy tile (half a period out of phase)
//   mode 2  waves 0..3 only MFMA phases, waves 4..7 only VALU phases (perfect role split: what the SIMD can overlap at best)
//   mode 3  only waves 0..3 run (one wave per SIMD): the serial time of one wave
//   mode 4  MFMA phases only, both waves;  mode 5  VALU phases only, both waves
//   mode 6  one wave per SIMD, finishing of tile t - 1 interleaved by hand between the MFMAs of tile t (needs a second result set)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_phase(const v4i (&A)[8], const v4i (&Bl)[8], const v4i (&Bh)[8], v8h mn, v8h bm, v16i& pl, v16i& ph, v16f& cm) {
    const v16i z = {};
    pl = z; ph = z;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pl = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[u], Bl[u], pl, 0, 0, 0);
        ph = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[u], Bh[u], ph, 0, 0, 0);
    }
    const v16f fz = {};
    cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(mn, bm, fz, 0, 0, 0);
}
__device__ __forceinline__ void valu_phase(v16i& pl, v16i& ph, v16f& cm, float& dw, float& dmin, float* da, float (&acc)[16]) {
    /* every input is opaque to the optimiser each tile (no instruction is emitted): nothing of the 96-instruction finishing can be hoisted */
    asm volatile("" : "+v"(dw), "+v"(dmin));
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(pl[r]), "+v"(ph[r]), "+v"(cm[r]), "+v"(da[r]));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[r] = __builtin_fmaf(dw * da[r], (float)((ph[r] << 3) + pl[r]), acc[r]);
        acc[r] = __builtin_fmaf(-(dmin * da[r]), cm[r], acc[r]);
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int tiles, float* out, const float* dain, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (MODE == 3 || MODE == 6) { if (wave >= 4) return; }
    v4i A[8], Bl[8], Bh[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { A[u] = (v4i){lane + u, u, 3, 7}; Bl[u] = (v4i){lane ^ u, 5, u, 1}; Bh[u] = (v4i){u, lane, 9, 2}; }
    v8h mn, bm;
#pragma unroll
    for (int e = 0; e < 8; ++e) { mn[e] = (_Float16)(float)(lane + e); bm[e] = (_Float16)(float)e; }
    float da[16], acc[16], acc2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { da[r] = dain[(lane + r) & 63]; acc[r] = 0.0f; acc2[r] = 0.0f; }
    float dw = dain[lane] * 0.5f, dmin = dain[63 - lane];
    v16i pl = {}, ph = {}; v16f cm = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0 || MODE == 3) {
        for (int t = 0; t < tiles; ++t) { mfma_phase(A, Bl, Bh, mn, bm, pl, ph, cm); __builtin_amdgcn_sched_barrier(0); valu_phase(pl, ph, cm, dw, dmin, da, acc); __builtin_amdgcn_sched_barrier(0); A[t & 7][0] += 1; }
    } else if (MODE == 1) {
        if (wave >= 4) { valu_phase(pl, ph, cm, dw, dmin, da, acc2); __builtin_amdgcn_sched_barrier(0); }
        for (int t = 0; t < tiles; ++t) { mfma_phase(A, Bl, Bh, mn, bm, pl, ph, cm); __builtin_amdgcn_sched_barrier(0); valu_phase(pl, ph, cm, dw, dmin, da, acc); __builtin_amdgcn_sched_barrier(0); A[t & 7][0] += 1; }
    } else if (MODE == 2) {
        if (wave < 4) { for (int t = 0; t < tiles; ++t) { mfma_phase(A, Bl, Bh, mn, bm, pl, ph, cm); __builtin_amdgcn_sched_barrier(0); A[t & 7][0] += pl[0] & 1; } }
        else { for (int t = 0; t < tiles; ++t) { valu_phase(pl, ph, cm, dw, dmin, da, acc); __builtin_amdgcn_sched_barrier(0); pl[0] += 1; } }
    } else if (MODE == 4) {
        for (int t = 0; t < tiles; ++t) { mfma_phase(A, Bl, Bh, mn, bm, pl, ph, cm); __builtin_amdgcn_sched_barrier(0); A[t & 7][0] += pl[0] & 1; }
    } else if (MODE == 5) {
        for (int t = 0; t < tiles; ++t) { valu_phase(pl, ph, cm, dw, dmin, da, acc); __builtin_amdgcn_sched_barrier(0); pl[0] += 1; }
    } else if (MODE == 6) {
        v16i ql = {}, qh = {}; v16f qm = {};
        for (int t = 0; t < tiles; ++t) {
            // MFMAs of tile t into (pl, ph, cm), finishing of tile t - 1 from (ql, qh, qm): 6 VALU after every MFMA
            const v16i z = {};
            pl = z; ph = z;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pl = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[u], Bl[u], pl, 0, 0, 0);
                { const int r = 2 * u; acc[r] = __builtin_fmaf(dw * da[r], (float)((qh[r] << 3) + ql[r]), acc[r]); acc[r] = __builtin_fmaf(-(dmin * da[r]), qm[r], acc[r]); }
                ph = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[u], Bh[u], ph, 0, 0, 0);
                { const int r = 2 * u + 1; acc[r] = __builtin_fmaf(dw * da[r], (float)((qh[r] << 3) + ql[r]), acc[r]); acc[r] = __builtin_fmaf(-(dmin * da[r]), qm[r], acc[r]); }
            }
            const v16f fz = {};
            cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(mn, bm, fz, 0, 0, 0);
            ql = pl; qh = ph; qm = cm;
            A[t & 7][0] += 1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r] + (float)pl[r] + (float)ph[r] + cm[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name) {
    float *out, *da; unsigned long long* cyc;
    hipMalloc((void**)&out, 256 * 512 * 4); hipMalloc((void**)&da, 64 * 4); hipMalloc((void**)&cyc, 8 * 8);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.001f * (i + 1);
    hipMemcpy(da, h, sizeof h, hipMemcpyHostToDevice); hipMemset(cyc, 0, 64);
    const int tiles = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, 50, out, da, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, tiles, out, da, cyc);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f us  cycles per tile: wave0 %6.0f  wave4 %6.0f\n", name, ms * 1e3, (double)c[0] / tiles, (double)c[4] / tiles);
}
int main() {
    run<4>("(warm-up)");
    for (int rep = 0; rep < 2; ++rep) {
        run<3>("3: one wave per SIMD, [17 MFMA][96 VALU] serial");
        run<0>("0: two waves per SIMD, same program, lockstep");
        run<1>("1: two waves per SIMD, partner half a period out of phase");
        run<2>("2: role split: wave0 MFMA phases only, wave4 VALU only");
        run<4>("4: MFMA phases only, both waves");
        run<5>("5: VALU phases only, both waves");
        run<6>("6: one wave per SIMD, finishing interleaved by hand");
    }
    return 0;
}
