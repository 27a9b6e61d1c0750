// Developer probe: issue cost of the VALU instructions the W4A8 unpack / finishing use (cycles per instruction and wave, one and two waves
// per SIMD), measured with s_memtime around a long unrolled stream of independent instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(512) void k(int iters, unsigned* out, unsigned long long* cyc) {
    unsigned a[8], b = threadIdx.x * 2654435761u | 1u, c = 0x00070007u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 77u + i * 0x01010101u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (OP == 2) { REP16(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 6) { REP16(asm volatile("v_lshrrev_b32 %0, 4, %0\n v_lshrrev_b32 %1, 4, %1\n v_lshrrev_b32 %2, 4, %2\n v_lshrrev_b32 %3, 4, %3\n v_lshrrev_b32 %4, 4, %4\n v_lshrrev_b32 %5, 4, %5\n v_lshrrev_b32 %6, 4, %6\n v_lshrrev_b32 %7, 4, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (OP == 8) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 9) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %8, %8\n v_mad_i32_i24 %1, %1, %8, %8\n v_mad_i32_i24 %2, %2, %8, %8\n v_mad_i32_i24 %3, %3, %8, %8\n v_mad_i32_i24 %4, %4, %8, %8\n v_mad_i32_i24 %5, %5, %8, %8\n v_mad_i32_i24 %6, %6, %8, %8\n v_mad_i32_i24 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (OP == 10) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %3, %3, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_lo_u16 %7, %7, %8 op_sel_hi:[1,0]" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP>
static void run(const char* name) {
    unsigned* out; unsigned long long* cyc;
    hipMalloc((void**)&out, 256 * 512 * 4); hipMalloc((void**)&cyc, 8);
    for (int waves = 4; waves <= 8; waves += 4) { // per workgroup: 4 = one wave per SIMD, 8 = two
        const int iters = 200;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, 10, out, cyc);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, iters, out, cyc);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction and wave\n", name, waves / 4, (double)c / (iters * 128.0));
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<2>("(warm-up) v_and_b32");
    run<2>("v_and_b32"); run<6>("v_lshrrev_b32"); run<0>("v_pk_mul_lo_u16"); run<10>("v_pk_mul_lo_u16 op_sel_hi:[1,0]"); run<1>("v_mul_u32_u24"); run<9>("v_mad_i32_i24");
    run<7>("v_mul_lo_u32"); run<3>("v_permlane16_swap_b32"); run<8>("v_perm_b32"); run<4>("v_cvt_f32_i32"); run<5>("v_fmac_f32");
    return 0;
}
