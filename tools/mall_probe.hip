// Developer probe: does the 256 MB Infinity Cache (memory-side) keep what a plain load stream brought in, and do non-temporal loads hit it?
// A buffer of S MB is read twice per mode; GB/s of each pass (HIP events).  Between modes 1 GB of other memory is read to flush.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const v4u* p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    size_t i = t0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        v4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
    }
    for (; i < n; i += stride) acc ^= p[i].x;
    if (acc == 0x9E3779B9u && n == 1) *sink = acc;
}
static float run(bool nt, const void* p, size_t bytes, unsigned* sink, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, (const v4u*)p, bytes / 16, sink);
    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, (const v4u*)p, bytes / 16, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (float)(bytes / 1e6 / ms); // GB/s
}
int main() {
    const size_t MB = 1 << 20;
    uint8_t *buf, *flush; unsigned* sink;
    hipMalloc((void**)&buf, 512 * MB); hipMalloc((void**)&flush, 1024 * MB); hipMalloc((void**)&sink, 64);
    hipMemset(buf, 1, 512 * MB); hipMemset(flush, 2, 1024 * MB);
    hipDeviceSynchronize();
    run(false, flush, 1024 * MB, sink, 2048);
    const int sizes[] = {32, 64, 96, 128, 192, 256, 384};
    printf("%-8s %-34s %10s %10s %10s\n", "MB", "mode (first pass, then twice more)", "pass1", "pass2", "pass3");
    for (int grid : {256, 2048}) {
        printf("grid %d workgroups of 256 threads\n", grid);
        for (int s : sizes) {
            struct { const char* name; bool a, b; } modes[] = {{"plain, plain, plain", false, false}, {"plain, nt, nt", false, true}, {"nt, plain, plain", true, false}, {"nt, nt, nt", true, true}};
            for (auto& m : modes) {
                run(false, flush, 1024 * MB, sink, 2048);
                const float g1 = run(m.a, buf, s * MB, sink, grid), g2 = run(m.b, buf, s * MB, sink, grid), g3 = run(m.b, buf, s * MB, sink, grid);
                printf("%-8d %-34s %10.0f %10.0f %10.0f  GB/s\n", s, m.name, g1, g2, g3);
            }
        }
    }
    return 0;
}
