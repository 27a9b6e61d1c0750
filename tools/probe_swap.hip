// probe: v_permlane16_swap_b32 lane semantics and the k pairing of v_mfma_i32_32x32x32_i8 on gfx950 (exact integer data)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o, int* c) {
    const int l = threadIdx.x;
    v2u r = __builtin_amdgcn_permlane16_swap(1000u + l, 2000u + l, false, false);
    o[l] = r.x; o[64 + l] = r.y;
    // A[m][k] = (m == 3) ? 1 at k = lane-half h, element j : encoded so that D[3][n] tells which (h, j) of B pairs with which of A
    v4i a = {0, 0, 0, 0}, b;
    const int m = l & 31, h = l >> 5;
    if (m == 3) { a[0] = 1 | (2 << 8) | (3 << 16) | (4 << 24); a[1] = 5 | (6 << 8) | (7 << 16) | (8 << 24); a[2] = 9 | (10 << 8) | (11 << 16) | (12 << 24); a[3] = 13 | (14 << 8) | (15 << 16) | (16 << 24);
                  if (h) { a[0] += 0x10101010; a[1] += 0x10101010; a[2] += 0x10101010; a[3] += 0x10101010; } }
    // B[k][n]: column n = m has a single 1 at element (h == (n >> 4 & 1), j == n & 15)
    b = (v4i){0, 0, 0, 0};
    if (h == ((m >> 4) & 1)) b[(m & 15) >> 2] = 1 << (8 * (m & 3));
    v16i acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
int main() {
    unsigned* o; int* c;
    hipMalloc(&o, 128 * 4); hipMalloc(&c, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c);
    unsigned ho[128]; int hc[1024];
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
    printf("swap.x:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, ho[i]); printf("\n");
    printf("swap.y:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, ho[64 + i]); printf("\n");
    // D[3][n]: row 3 = reg 3 of lanes with h == 0 (row = (reg&3) + 8*(reg>>2) + 4*h)
    printf("D[3][n] (expect 1..16 for n<16 (h=0,j=n), 17..32 for n>=16):"); for (int n = 0; n < 32; ++n) printf(" %d", hc[n * 16 + 3]); printf("\n");
    return 0;
}
