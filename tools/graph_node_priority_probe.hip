// Does this HIP runtime honour per-node priorities of a captured graph (hipGraphInstantiateFlagUseNodePriority)?  Developer probe, needs a GPU:
//   hipcc --offload-arch=gfx950 -O2 tools/graph_node_priority_probe.hip -o build/probes/graph_node_priority_probe && build/probes/graph_node_priority_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_nop(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1); }
#define SHOW(call) do { hipError_t e = (call); printf("%-70s -> %s\n", #call, hipGetErrorString(e)); } while (0)
int main() {
    int* d = nullptr;
    hipStream_t s;
    SHOW(hipMalloc(&d, 4));
    SHOW(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    SHOW(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    hipGraph_t g = nullptr;
    SHOW(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_nop, dim3(64), dim3(64), 0, s, d);
    SHOW(hipStreamEndCapture(s, &g));
    size_t n = 0;
    SHOW(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    SHOW(hipGraphGetNodes(g, nodes.data(), &n));
    for (size_t i = 0; i < n; ++i) {
        hipKernelNodeAttrValue v{};
        v.priority = (i & 1) ? hi : lo;
        hipError_t e = hipGraphKernelNodeSetAttribute(nodes[i], hipKernelNodeAttributePriority, &v);
        hipKernelNodeAttrValue r{};
        hipError_t e2 = hipGraphKernelNodeGetAttribute(nodes[i], hipKernelNodeAttributePriority, &r);
        printf("node %zu: set priority %d -> %s; get -> %s (%d)\n", i, v.priority, hipGetErrorString(e), hipGetErrorString(e2), r.priority);
    }
    hipGraphExec_t ge = nullptr;
    SHOW(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagUseNodePriority));
    if (ge) { SHOW(hipGraphLaunch(ge, s)); SHOW(hipStreamSynchronize(s)); }
    hipGraphExec_t g0 = nullptr;
    SHOW(hipGraphInstantiate(&g0, g, nullptr, nullptr, 0));
    if (g0) { SHOW(hipGraphLaunch(g0, s)); SHOW(hipStreamSynchronize(s)); }
    return 0;
}
