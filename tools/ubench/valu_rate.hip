// Micro-benchmark: per-SIMD issue cost of the VALU instructions of the softmax (4 waves per SIMD, 8 independent
// registers per wave).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[u]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[u]));
            if (MODE == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[u]) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[u]) : "v"(a));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[u & 6]) : "v"(*(double*)&r[(u & 6) ^ 2]));
            if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(r[u]) : "v"(a));
        }
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += r[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    k<MODE><<<1024, 256>>>(out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<1024, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 1024.0 * 4 * iters * 8;  // wave-instructions
    printf("%-18s %.3f ms, %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (n / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_exp_f32");
    run<2>("v_max3_f32");
    run<3>("v_cvt_pk_f16_f32");
    run<4>("v_pk_mul_f32");
    run<5>("v_mov_b32");
    return 0;
}
