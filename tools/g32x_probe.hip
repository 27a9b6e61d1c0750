// Developer probe for a ONE-wave-per-SIMD form of the 256-row W4A8 GEMM tile loop (512 registers per lane): what does a tile cost when the
// finishing of tile t - 1, an eighth of the weight unpack and the LDS operand reads of tile t + 1 are all issued between the MFMAs of tile t
// by the SAME wave?  Per 32x32x256 tile: 16 x v_mfma_i32_32x32x32_i8 (two independent digit chains) + 1 x v_mfma_f32_32x32x16_f16,
// 96 finishing VALU, UNP filler VALU (the unpack's share), 13 x ds_read_b128, DMA x global_load_lds_dwordx4.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/g32x_probe.hip -o /tmp/g32x_probe && /tmp/g32x_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

struct Res { v16i pl, ph; v16f cm; v4f da[4]; };
struct Opa { v4i a[8]; v8h mn; };

template <int SCHED, int UNP, bool LDSR, int DMA, int SID>
__device__ __forceinline__ void tile_step(const uint8_t* lp, int t, const Opa& cur, Opa& nxt, const v4i (&Bl)[8], const v4i (&Bh)[8], v8h bm, Res& now, const Res& prev,
                                          float dw, float dmin, float (&acc)[16], unsigned (&junk)[8], const uint8_t* gsrc, uint8_t* ldst) {
    // LDS operand reads of the next tile
    if (LDSR) {
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt.a[u] = *(const v4i*)(lp + ((t + 1) & 7) * 8192 + (u >> 1) * 1024 + (u & 1) * 256);
        nxt.mn = *(const v8h*)(lp + 65536 + ((t + 1) & 7) * 1024);
#pragma unroll
        for (int b = 0; b < 4; ++b) now.da[b] = *(const v4f*)(lp + 73728 + (t & 7) * 128 + b * 32);
    }
    if (DMA) {
        const auto gs = (const __attribute__((address_space(1))) void*)(gsrc + (t & 63) * 1024);
        const auto ls = (__attribute__((address_space(3))) void*)(ldst + (t & 7) * 1024);
        __builtin_amdgcn_global_load_lds(gs, ls, 16, 0, 0);
        if (DMA > 1 && (t & 1)) __builtin_amdgcn_global_load_lds(gs, ls, 16, 8192, 0);
    }
    const v16i z = {};
    now.pl = z; now.ph = z;
    if (SCHED != 9) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            now.pl = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a[u], Bl[u], now.pl, 0, 0, 0);
            now.ph = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a[u], Bh[u], now.ph, 0, 0, 0);
        }
        const v16f fz = {};
        now.cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.mn, bm, fz, 0, 0, 0);
    } else { /* no MFMAs: the operands stay live */
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(cur.a[u]));
        asm volatile("" :: "v"(cur.mn));
        asm volatile("" : "+v"(now.pl), "+v"(now.ph));
    }
    // finishing of the previous tile (UNP < 0: none)
    if (UNP >= 0)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * b + i;
            acc[r] = __builtin_fmaf(dw * prev.da[b][i], (float)((prev.ph[r] << 3) + prev.pl[r]), acc[r]);
            acc[r] = __builtin_fmaf(-(dmin * prev.da[b][i]), prev.cm[r], acc[r]);
        }
    // the unpack's share: UNP integer VALU instructions on values the optimiser cannot fold
#pragma unroll
    for (int i = 0; i < (UNP > 0 ? UNP : 0); ++i) junk[i & 7] = __builtin_amdgcn_perm(junk[i & 7], junk[(i + 5) & 7], 0x07020500u + i); /* one full-rate VALU instruction each */
    if (SCHED == 1) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SID);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, SID);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, SID);
        }
    } else if (SCHED == 2) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SID);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, SID);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, SID);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, SID);
        }
    }
}

template <int SCHED, int UNP, bool LDSR, int DMA>
__global__ __launch_bounds__(256, 1) void k1(int tiles, float* out, const float* dain, unsigned long long* cyc, const uint8_t* gsrc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 81920 / 4; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const uint8_t* lp = lds + ((lane >> 4) & 1) * 4096 + ((lane >> 5) * 32 + (lane & 15)) * 16;
    v4i Bl[8], Bh[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { Bl[u] = (v4i){lane ^ u, 5, u, 1}; Bh[u] = (v4i){u, lane, 9, 2}; }
    v8h bm;
#pragma unroll
    for (int e = 0; e < 8; ++e) bm[e] = (_Float16)(float)e;
    float acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    float dw = dain[lane] * 0.5f, dmin = dain[63 - lane];
    unsigned junk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) junk[i] = lane * 77u + i;
    Opa o0, o1;
    Res r0, r1;
#pragma unroll
    /* the two ping-pong operand sets DIFFER: with identical sets (round 3) the compiler computed the two tile steps' MFMAs once — the modes
     * without LDS reads issued 17 MFMAs per TWO tiles, which is what "17 MFMA only: 273 cycles per tile" and "17 MFMA + 96 VALU: 621" timed */
    for (int u = 0; u < 8; ++u) { o0.a[u] = (v4i){lane + u, u, 3, 7}; o1.a[u] = (v4i){lane - u, u + 1, 5, 2}; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0.mn[e] = (_Float16)(float)(lane + e); o1.mn[e] = (_Float16)(float)(lane - e); }
    r0.pl = r0.ph = r1.pl = r1.ph = (v16i){};
    r0.cm = r1.cm = (v16f){};
#pragma unroll
    for (int b = 0; b < 4; ++b) { r0.da[b] = (v4f){1.f, 2.f, 3.f, 4.f}; r1.da[b] = r0.da[b]; }
    uint8_t* ldst = lds + 81920 + wave * 16384; /* DMA destination: wave-uniform base */
    const uint8_t* gs = gsrc + ((size_t)blockIdx.x * 4 + wave) * 65536 + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; t += 2) {
        /* operand / result sets ping-pong; two of the real kernel's eight accumulator tiles (the count does not change the instruction stream) */
        tile_step<SCHED, UNP, LDSR, DMA, 0>(lp, t, o0, o1, Bl, Bh, bm, r0, r1, dw, dmin, acc[1], junk, gs, ldst);
        tile_step<SCHED, UNP, LDSR, DMA, 1>(lp, t + 1, o1, o0, Bl, Bh, bm, r1, r0, dw, dmin, acc[0], junk, gs, ldst);
        Bl[0][0] += 1; /* every operand set changes per iteration: a loop-invariant chain (round 3: the Bh chains and the min-term MFMA in the modes */
        Bh[0][0] += 1; /* without finishing) is hoisted out of the loop by the compiler and the mode then times 8 MFMAs per tile, not 17 */
        bm[0] = bm[0] + (_Float16)1.0f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[m][r];
#pragma unroll
    /* BOTH digit chains of BOTH result sets are consumed (round 3 read only r0.pl and r1.ph: in the modes without finishing the other two
     * chains were dead code) — see also the operand updates in the loop: "17 MFMA only: 273 cycles per tile" of profiles/r03_g32x_probe2.txt
     * timed 8 MFMAs per tile (8 x 32 + 17 cycles), the int8 32x32x32 MFMA takes 32 cycles (profiles/r04_g32x_probe_fixed.txt) */
    for (int r = 0; r < 16; ++r) s += (float)r0.pl[r] + (float)r1.ph[r] + (float)r0.ph[r] + (float)r1.pl[r] + r0.cm[r] + r1.cm[r];
    for (int i = 0; i < 8; ++i) s += (float)junk[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int SCHED, int UNP, bool LDSR, int DMA>
static void run(const char* name) {
    float *out, *da; unsigned long long* cyc; uint8_t* g;
    hipMalloc((void**)&out, 256 * 256 * 4); hipMalloc((void**)&da, 64 * 4); hipMalloc((void**)&cyc, 8 * 8); hipMalloc((void**)&g, (size_t)256 * 4 * 65536 + 65536);
    hipMemset(g, 1, (size_t)256 * 4 * 65536 + 65536);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.001f * (i + 1);
    hipMemcpy(da, h, sizeof h, hipMemcpyHostToDevice); hipMemset(cyc, 0, 64);
    auto kern = k1<SCHED, UNP, LDSR, DMA>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int tiles = 2048;
    const size_t ldsb = 81920 + 4 * 16384;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), ldsb, 0, 64, out, da, cyc, g);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), ldsb, 0, tiles, out, da, cyc, g);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-72s %8.1f us  cycles per tile: %6.0f (matrix pipe 544)\n", name, ms * 1e3, (double)c[0] / tiles);
    hipFree(out); hipFree(da); hipFree(cyc); hipFree(g);
}
int main() {
    run<0, 0, false, 0>("(warm-up)");
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, false, 0>("17 MFMA + 96 finishing VALU");
        run<0, -1, false, 0>("17 MFMA only");
        run<0, -1, true, 0>("17 MFMA + 13 ds_read_b128 (no VALU)");
        run<0, 0, true, 0>("17 MFMA + 96 VALU + 13 ds_read_b128");
        run<9, 0, true, 0>("96 VALU + 13 ds_read_b128 (no MFMA)");
        run<9, 0, false, 0>("96 VALU only");
        run<9, -1, true, 0>("13 ds_read_b128 only");
        /* the whole one-wave tile: + the unpack's share at 256 rows per wave (24 VALU) + two LDS-DMA pieces per tile, in compiler order and with
         * the instruction stream grouped per MFMA by sched_group_barrier (1 MFMA : 1 LDS read : 7 VALU / 1 : 4 VALU : 1 read : 3 VALU) */
        run<0, 24, true, 2>("compiler order: 17 MFMA + 96 + 24 VALU + 13 ds_read + 1.5 DMA");
        run<1, 24, true, 2>("grouped (read first): the same");
        run<2, 24, true, 2>("grouped (read mid): the same");
        run<0, 24, false, 0>("compiler order: 17 MFMA + 96 + 24 VALU");
        run<1, 24, false, 0>("grouped: 17 MFMA + 96 + 24 VALU");
    }
    return 0;
}
