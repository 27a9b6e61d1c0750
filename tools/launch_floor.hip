// Developer probe: what does one dependent kernel launch cost on this box?  (empty kernels, back-to-back in one stream,
// eager and as a hipGraph chain, with and without a large dynamic-LDS request, with a small global write epilogue)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { char pad[160]; float* out; };
__global__ __launch_bounds__(512) void k_empty(Args a, int write) {
    extern __shared__ char lds[];
    if (write) a.out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = 1.0f;
}
static float run(hipStream_t s, int grid, int block, size_t lds, int write, int n, bool graph, Args a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    float ms = 0;
    if (!graph) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, s, a, write);
        hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, s, a, write);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, e0, e1);
    } else {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, s, a, write);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, e0, e1);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return ms * 1000.0f / n;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    Args a{}; hipMalloc((void**)&a.out, 256 * 512 * 4 * 8);
    const int n = 500;
    for (int graph = 0; graph < 2; ++graph)
        for (size_t lds : {(size_t)0, (size_t)136 * 1024})
            for (int grid : {256, 2048})
                for (int write = 0; write < 2; ++write)
                    printf("%s lds=%6zu grid=%4d block=512 write=%d : %.2f us/launch\n", graph ? "graph" : "eager", lds, grid, write,
                           run(s, grid, 512, lds, write, n, graph != 0, a));
    return 0;
}
