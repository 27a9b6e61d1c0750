// Developer probe: LDS read bandwidth per CU for the operand read patterns of the W4A8 kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((int*)lds)[i] = i;
    __syncthreads();
    v4i s = {0, 0, 0, 0};
    const unsigned char* base = lds + (wave & 7) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) { // two 8-byte reads 512 B apart (ds_read2st64_b64), lane * 8
                const v2i a0 = *(const v2i*)(base + lane * 8 + (2 * j) * 512);
                const v2i a1 = *(const v2i*)(base + lane * 8 + (2 * j + 1) * 512);
                s += (v4i){a0.x, a0.y, a1.x, a1.y};
            } else if (MODE == 1) { // one 16-byte read (ds_read_b128), lane * 16
                s += *(const v4i*)(base + lane * 16 + j * 1024);
            } else { // one 8-byte read
                const v2i a0 = *(const v2i*)(base + lane * 8 + j * 512);
                s += (v4i){a0.x, a0.y, 0, 0};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (s.x + s.y + s.z + s.w == 0x12345678) out[0] = 1;
}
template <int MODE>
static void run(const char* name, int waves, int bytes_per_iter_per_wave) {
    int* out; hipMalloc((void**)&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 32768, 0, 2000, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 32768, 0, iters, out);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double gbs = (double)iters * bytes_per_iter_per_wave * waves / (ms * 1e-3) / 1e9;
    printf("%-34s waves/CU=%d : %.1f GB/s per CU (= %.1f B/clk @2.4 GHz)\n", name, waves, gbs, gbs / 2.4);
}
int main() {
    for (int w : {4, 7, 8}) {
        run<0>("ds_read2st64_b64 (2 x 8 B/lane)", w, 4096);
        run<1>("ds_read_b128 (16 B/lane)", w, 4096);
        run<2>("ds_read_b64 (8 B/lane)", w, 2048);
    }
    return 0;
}
