// Micro-benchmark: issue rate of the gfx950 MFMA shapes used / considered by the kernels.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_rate mfma_rate.hip && ./mfma_rate
// (without -amdgpu-mfma-vgpr-form hipcc shuffles the 16x16x32 accumulators through v_accvgpr copies inside the loop and the
// one-wave figures measure those stalls: 46 cycles per MFMA instead of the hardware rate)
// Round-3 rewrite: the first version fed the SAME register as A and B, ran 4 waves per SIMD while
// its comment said one, and gave the 32x32x16 shape only 2 accumulators -- its 16x16x32 figure (27 cycles / 1.16 PF)
// contradicted MI355X_MICROARCH.md (17 cycles back to back on one wave, 2495 TF for 32x32x16).  This version:
//   * distinct A and B operand registers, both loaded from memory (the compiler cannot fold them);
//   * NACC independent accumulators (4 and 8) for BOTH shapes;
//   * waves per SIMD = 1 (256-thread blocks, 1 block per CU forced by a 100 KB LDS request), 2 and 4 (2 / 4 blocks per CU);
//   * cycles per MFMA per SIMD measured INSIDE the kernel with s_memtime (shader-clock ticks) as well as wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(256) k(const h8* __restrict__ ab, float* out, long long* ticks, int iters) {
    extern __shared__ char lds_pad[];  // only to bound the blocks per CU
    const h8 a = ab[threadIdx.x], b = ab[256 + threadIdx.x];
    f4 c[NACC];
    f16v d[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) {
        c[u] = f4{0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 16; ++v) d[u][v] = 0.f;
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) {
            if (SHAPE == 16) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u], 0, 0, 0);
            else d[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d[u], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += c[u][0] + c[u][3] + d[u][0] + d[u][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = one wave per SIMD each
    const size_t lds = waves_per_simd == 1 ? 100 * 1024 : (waves_per_simd == 2 ? 70 * 1024 : 36 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; long long* ticks; h8* ab;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&ticks, blocks * 4 * 8);
    hipMalloc(&ab, 512 * sizeof(h8));
    _Float16 hab[512 * 8];
    for (int i = 0; i < 512 * 8; ++i) hab[i] = (_Float16)((rand() % 2001 - 1000) * 1e-3f);  // random operands (DVFS: not zeros)
    hipMemcpy(ab, hab, sizeof(hab), hipMemcpyHostToDevice);
    const int iters = 20000;
    k<SHAPE, NACC><<<blocks, 256, lds>>>(ab, out, ticks, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<SHAPE, NACC><<<blocks, 256, lds>>>(ab, out, ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long* ht = (long long*)malloc(blocks * 4 * 8);
    hipMemcpy(ht, ticks, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double tsum = 0;
    for (int i = 0; i < blocks * 4; ++i) tsum += (double)ht[i];
    const double per_wave_ticks = tsum / (blocks * 4);
    const double n_wave = (double)iters * NACC;                     // MFMAs issued by one wave
    const double n_simd = n_wave * waves_per_simd;                  // MFMAs through one SIMD's matrix pipe
    const double flop = 2.0 * (SHAPE == 16 ? 16 * 16 * 32 : 32 * 32 * 16);
    // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md, constants table): ticks / MFMA = issue cycles
    printf("%dx%dx%d f16, %d acc, %d wave/SIMD: %.3f ms, %.0f TFLOP/s, %.2f ns per MFMA per SIMD (wall), "
           "%.2f s_memtime ticks per MFMA per SIMD\n",
           SHAPE, SHAPE, SHAPE == 16 ? 32 : 16, NACC, waves_per_simd, ms, 1024.0 * n_simd * flop / ms / 1e9,
           ms * 1e6 / n_simd, per_wave_ticks / n_simd);
    free(ht);
    hipFree(out); hipFree(ticks); hipFree(ab);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<16, 4>(w);
        run<16, 8>(w);
        run<32, 4>(w);
        run<32, 8>(w);
    }
    return 0;
}
