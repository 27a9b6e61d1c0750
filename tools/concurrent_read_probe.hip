// Developer probe ("next round" 1a): is the ~3.3 TB/s that three concurrent 16-row decode streams reach together the
// part's cap for many small concurrent launches, or k_gemv_w4a8's?  Plain read kernels of exactly the decode step's mat-vec launch sizes
// (per layer q|k|v 14.2 MB, o 9.4 MB, gate|up 66.1 MB, down 33.0 / 48.2 MB; logits matrix 107.5 MB: 129 launches, 4.29 GB) are issued
//   (1) on one stream, (2) on three streams at once (all reading the SAME 4.29 GB, as three decode groups of one model do),
//   (3) as (2) with a 1-workgroup "producer" kernel between the reads (the norm / quantise launches of a real step),
//   (4) three streams reading three DIFFERENT copies.
// Each stream's launches are captured into one hipGraph (as the product's decode step is) and replayed.
//   hipcc --offload-arch=gfx950 -O3 tools/concurrent_read_probe.hip -o /tmp/crp && /tmp/crp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const v4u* p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    size_t i = t0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        v4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
    }
    for (; i < n; i += stride) acc ^= p[i].x;
    if (acc == 0x9E3779B9u && n == 1) *sink = acc;
}
__global__ void k_small(unsigned* sink) { if (threadIdx.x == 1023 && sink[1] == 77) sink[2] = 1; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static std::vector<size_t> step_sizes() { // bytes per launch of one decode step, Mistral-7B Q4_K_M
    std::vector<size_t> v;
    for (int l = 0; l < 32; ++l) {
        const bool more = l < 4 || l >= 28 || (l - 4) % 3 == 2;
        const double bv = more ? 210.0 / 256 : 144.0 / 256;
        v.push_back((size_t)(4096.0 * 5120 * 0.5625 + 4096.0 * 1024 * (bv - 0.5625)));  // q|k|v (v may be Q6_K)
        v.push_back((size_t)(4096.0 * 4096 * 0.5625));
        v.push_back((size_t)(2 * 4096.0 * 14336 * 0.5625));
        v.push_back((size_t)(4096.0 * 14336 * bv));
    }
    v.push_back((size_t)(4096.0 * 32000 * 210 / 256));
    for (auto& s : v) s &= ~(size_t)4095;
    return v;
}

int main() {
    const auto sizes = step_sizes();
    size_t total = 0; for (auto s : sizes) total += s;
    uint8_t* buf[3]; unsigned* sink;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc((void**)&buf[i], total)); CK(hipMemset(buf[i], i + 1, total)); }
    CK(hipMalloc((void**)&sink, 64)); CK(hipMemset(sink, 0, 64));
    hipStream_t st[3]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("decode step as plain reads: %zu launches, %.3f GB\n", sizes.size(), total / 1e9);
    struct Mode { const char* name; int nstreams; bool small; bool distinct; int grid; };
    const Mode modes[] = {
        {"1 stream, grid 256", 1, false, false, 256}, {"1 stream, grid 256, + small kernel after each read", 1, true, false, 256},
        {"1 stream, grid 512", 1, false, false, 512},
        {"3 streams, same weights, grid 256", 3, false, false, 256}, {"3 streams, same weights, grid 256, + small kernels", 3, true, false, 256},
        {"3 streams, same weights, grid 512", 3, false, false, 512}, {"3 streams, three copies, grid 256", 3, false, true, 256},
        {"2 streams, same weights, grid 256", 2, false, false, 256},
    };
    for (const auto& m : modes) {
        hipGraphExec_t ex[3];
        for (int s = 0; s < m.nstreams; ++s) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
            size_t off = 0;
            for (size_t b : sizes) {
                hipLaunchKernelGGL(k_read, dim3(m.grid), dim3(256), 0, st[s], (const v4u*)(buf[m.distinct ? s : 0] + off), b / 16, sink);
                if (m.small) { hipLaunchKernelGGL(k_small, dim3(16), dim3(1024), 0, st[s], sink); hipLaunchKernelGGL(k_small, dim3(16), dim3(1024), 0, st[s], sink); }
                off += b;
            }
            CK(hipStreamEndCapture(st[s], &g));
            CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        const int reps = 6;
        for (int s = 0; s < m.nstreams; ++s) CK(hipGraphLaunch(ex[s], st[s]));  // warm-up
        CK(hipDeviceSynchronize());
        hipEvent_t e0[3], e1[3];
        for (int s = 0; s < m.nstreams; ++s) { CK(hipEventCreate(&e0[s])); CK(hipEventCreate(&e1[s])); }
        for (int s = 0; s < m.nstreams; ++s) CK(hipEventRecord(e0[s], st[s]));
        for (int r = 0; r < reps; ++r) for (int s = 0; s < m.nstreams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
        for (int s = 0; s < m.nstreams; ++s) CK(hipEventRecord(e1[s], st[s]));
        CK(hipDeviceSynchronize());
        float worst = 0;
        for (int s = 0; s < m.nstreams; ++s) { float ms; CK(hipEventElapsedTime(&ms, e0[s], e1[s])); worst = ms > worst ? ms : worst; }
        printf("%-58s %7.3f ms per step and stream, aggregate %6.2f TB/s (%.3f of 8)\n", m.name, worst / reps, m.nstreams * reps * (total / 1e9) / worst,
               m.nstreams * reps * (total / 1e9) / worst / 8.0);
        for (int s = 0; s < m.nstreams; ++s) { hipGraphExecDestroy(ex[s]); hipEventDestroy(e0[s]); hipEventDestroy(e1[s]); }
    }
    return 0;
}
