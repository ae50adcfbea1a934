// Micro-benchmark: issue rate of the gfx950 MFMA shapes used / considered by the kernels (one wave per SIMD, 4
// independent accumulators).  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    h8 a8; h4 a4;
    for (int i = 0; i < 8; ++i) a8[i] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int i = 0; i < 4; ++i) a4[i] = a8[i];
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v d[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c[u], 0, 0, 0);
            if (MODE == 1) c[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, c[u], 0, 0, 0);
            if (MODE == 2) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, d[u & 1], 0, 0, 0);
            if (MODE == 3) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, a4, d[u & 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][3];
    s += d[0][0] + d[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double flops_per) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    k<MODE><<<1024, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<1024, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 1024.0 * 4 * iters * 4;  // wave-instructions
    printf("%s: %.3f ms, %.1f TFLOP/s, %.2f ns per wave-instruction per SIMD\n", name, ms, n * flops_per / ms / 1e9,
           ms * 1e6 / (n / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("16x16x32_f16", 2.0 * 16 * 16 * 32);
    run<1>("16x16x16_f16", 2.0 * 16 * 16 * 16);
    run<2>("32x32x16_f16", 2.0 * 32 * 32 * 16);
    run<3>("32x32x8_f16", 2.0 * 32 * 32 * 8);
    return 0;
}
