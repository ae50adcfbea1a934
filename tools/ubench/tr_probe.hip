// What ds_read_b64_tr_b16 returns: LDS holds the element index at every 16-bit slot; lane l passes the 8-byte address of
// elements 4 l .. 4 l + 3 (lane-linear) and prints what it received.  Expected (cdna_hip_programming.md, T10): within a
// 16-lane group the 16 x 4 elements are a row-major [4][16] block and lane i receives column i of it, i.e.
// out[l][j] = 64 (l >> 4) + 16 j + (l & 15).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" %3d", h[l * 4 + j]);
            bad += h[l * 4 + j] != 64 * (l >> 4) + 16 * j + (l & 15);
        }
        printf("\n");
    }
    printf("mismatches against the expected gather: %d\n", bad);
    return 0;
}
