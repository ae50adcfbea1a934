// What v_permlane32_swap_b32 / v_permlane16_swap_b32 do on gfx950, lane by lane (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -o permlane_probe permlane_probe.hip && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    // the same through two float variables that are written back (the pattern csrc/tchain.hip uses)
    float fa = (float)threadIdx.x, fb = 100.f + threadIdx.x;
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, fa), __builtin_bit_cast(unsigned, fb), false, false);
    fa = __builtin_bit_cast(float, q[0]);
    fb = __builtin_bit_cast(float, q[1]);
    out[128 + threadIdx.x] = (unsigned)fa;
    out[192 + threadIdx.x] = (unsigned)fb;
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 4; ++s) {
        printf("%s:", s == 0 ? "r[0] (a=lane, b=100+lane)" : s == 1 ? "r[1]" : s == 2 ? "fa" : "fb");
        for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 * s + i]);
        printf("\n");
    }
    return 0;
}
