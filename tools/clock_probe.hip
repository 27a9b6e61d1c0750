// What clock does a nearly idle chip run a lone dependent-fma chain at?  (the long-context PV chain is such a load: 128 single-wave workgroups)
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o build/probes/clock_probe && build/probes/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_chain(float* out, unsigned long long* stamps, int n) {
    float a = out[threadIdx.x], b = 1.000001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
#pragma unroll 100
    for (int i = 0; i < n; ++i) { a = __builtin_fmaf(a, b, c); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float* d; unsigned long long* st; unsigned long long h[2 * 2048];
    hipMalloc(&d, 2048 * 64 * 4); hipMemset(d, 0, 2048 * 64 * 4); hipMalloc(&st, sizeof h);
    const int n = 200000;
    for (int wgs : {1, 16, 128, 256, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, 0, d, st, n);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, st, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
        const double cyc = (double)h[0], ns = (double)h[1] * 10.0;
        printf("%4d workgroups of one wave: %.2f shader cycles per dependent fma, %.2f ns per fma -> %.0f MHz shader clock\n", wgs, cyc / n, ns / n, cyc / ns * 1000.0);
    }
    return 0;
}
