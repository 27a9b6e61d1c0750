// Developer probe: issue rate of the int8 MFMA forms used by the W4A8 kernels (independent and 2-chain dependent issue).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int* out, long a8, v4i a16) {
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    v4i b16 = a16 + (int)threadIdx.x;
    long b8 = a8 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 4 independent x64 chains
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c3, 0, 0, 0);
        } else if (MODE == 1) { // 2 chains
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c1, 0, 0, 0);
        } else if (MODE == 2) { // 1 chain
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0);
        } else { // x32 form, 4 independent
            c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c3, 0, 0, 0);
        }
    }
    v4i s = c0 + c1 + c2 + c3;
    if (s.x == 0x7fffffff) out[0] = s.y + s.z + s.w;
}
template <int MODE>
static void run(const char* name, int waves_per_simd) {
    int* out; hipMalloc((void**)&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    v4i a16 = {1, 2, 3, 4};
    hipLaunchKernelGGL(k<MODE>, dim3(256 * waves_per_simd), dim3(256), 0, 0, 100, out, 5L, a16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * waves_per_simd), dim3(256), 0, 0, iters, out, 5L, a16);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e6 / ((double)iters * 4 * waves_per_simd); // ns per MFMA per SIMD
    printf("%-28s waves/SIMD=%d : %.2f ns per MFMA per SIMD  (= %.1f cycles @2.4 GHz)\n", name, waves_per_simd, per, per * 2.4);
}
int main() {
    run<2>("(clock warm-up, ignore)", 2);
    run<2>("(clock warm-up, ignore)", 2);
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) {
            run<2>("x64 i8, 1 chain", w);
            run<1>("x64 i8, 2 chains", w);
            run<0>("x64 i8, 4 indep chains", w);
            run<3>("x32 i8, 4 indep chains", w);
        }
    return 0;
}
