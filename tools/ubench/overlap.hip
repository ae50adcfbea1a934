// Micro-benchmark: do the matrix pipe and the VALU of ONE SIMD overlap?  512-thread workgroups, one per CU: waves
// 0-3 (one per SIMD) run back-to-back v_mfma_f32_32x32x16_f16, waves 4-7 (their SIMD partners) run v_exp_f32 + v_fma.
// Times: MFMA waves alone, VALU waves alone, both, and the same work interleaved inside every wave.
//   hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip && ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode bit 0: MFMA waves active, bit 1: VALU waves active, mode 4: every wave interleaves both (half the work each)
template <int MODE, int nm, int nv, int WPS = 2>
__global__ void __launch_bounds__(WPS * 256) k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    f16v d[2] = {};
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = -0.001f * (threadIdx.x + i);
    if (MODE == 4) {  // every wave: one MFMA, then its share of VALU, repeated (1 / WPS of the work per wave)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < nm / WPS; ++u) {
                d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, d[u & 1], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < (nv / WPS) / (nm / WPS); ++v) e[v & 7] = __builtin_amdgcn_exp2f(e[v & 7]) - 1.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MODE == 5) {  // every wave: all its MFMAs, then all its VALU (the structure of a softmax between two GEMMs)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < nm / WPS; ++u) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, d[u & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < nv / WPS; ++v) e[v & 7] = __builtin_amdgcn_exp2f(e[v & 7]) - 1.0f;
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 6) {  // like 5, but the VALU block consumes the MFMA results (softmax after a GEMM)
        for (int it = 0; it < iters; ++it) {
            f16v z = {};
#pragma unroll
            for (int u = 0; u < nm / WPS; ++u) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, (u < 2) ? z : d[u & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < nv / WPS; ++v) e[v & 7] += __builtin_amdgcn_exp2f(d[v & 1][(v >> 1) & 15] * 1e-6f);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (wave < 4) {
        if (MODE & 1)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < nm; ++u) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, d[u & 1], 0, 0, 0);
    } else {
        if (MODE & 2)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int v = 0; v < nv; ++v) e[v & 7] = __builtin_amdgcn_exp2f(e[v & 7]) - 1.0f;
    }
    float s = d[0][0] + d[1][5];
    for (int i = 0; i < 8; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int nm, int nv, int WPS = 2>
float run(float* out, int iters) {
    k<MODE, nm, nv, WPS><<<256, WPS * 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, nm, nv, WPS><<<256, WPS * 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
#define ROW(NM, NV)                                                                                                   \
    {                                                                                                                 \
        const float m = run<1, NM, NV>(out, iters), v = run<2, NM, NV>(out, iters), b = run<3, NM, NV>(out, iters),    \
                    x = run<4, NM, NV>(out, iters), y = run<5, NM, NV>(out, iters), x4 = run<4, NM, NV, 4>(out, iters),  \
                    y4 = run<5, NM, NV, 4>(out, iters), x1 = run<4, NM, NV, 1>(out, iters), y1 = run<5, NM, NV, 1>(out, iters), w4 = run<6, NM, NV, 4>(out, iters), w2 = run<6, NM, NV, 2>(out, iters); \
        printf("per iteration: %d MFMA 32x32x16 | %d (v_exp + v_sub): mfma waves alone %.3f ms, valu waves alone %.3f ms, " \
               "both (partner waves of a SIMD) %.3f ms [sum %.3f, max %.3f]; same total work split over W waves per SIMD, "  \
               "MFMA/VALU interleaved | blocked inside each wave: W=1 %.3f | %.3f, W=2 %.3f | %.3f, W=4 %.3f | %.3f ms; blocked with the VALU consuming the MFMA results: W=2 %.3f, W=4 %.3f ms\n", \
               NM, NV, m, v, b, m + v, m > v ? m : v, x1, y1, x, y, x4, y4, w2, w4);                                                               \
    }
    ROW(8, 16) ROW(8, 32) ROW(8, 64) ROW(8, 128)
    return 0;
}
