// Micro-benchmark: achievable global(L2-resident) -> LDS fill rate per CU with global_load_lds (LDS-DMA), in the
// access pattern of the igemm loader (8 rows x 128 B per wave-instruction, row stride `ld` bytes), as a function
// of waves per workgroup, workgroups per CU, ring depth and pieces per wave per stage.
//   hipcc --offload-arch=gfx950 -O3 -o lds_fill lds_fill.hip && ./lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int NW, int STAGES, int PIECES, bool BARRIER>
__global__ void __launch_bounds__(NW * 64) fill(const char* src, int ld, int iters, size_t span, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STAGE_BYTES = NW * PIECES * 1024;
    // this lane's row / chunk inside a piece, like the igemm loader
    const char* p[PIECES];
    for (int i = 0; i < PIECES; ++i) {
        const size_t row = (size_t)blockIdx.x * NW * PIECES * 8 + (wave * PIECES + i) * 8 + (lane >> 3);
        p[i] = src + (row * ld) % span + (lane & 7) * 16;
    }
    float acc = 0.f;
    // prologue: STAGES-1 stages in flight
    for (int s = 0; s < STAGES - 1; ++s)
        for (int i = 0; i < PIECES; ++i) { glds16(p[i], smem + s * STAGE_BYTES + (wave * PIECES + i) * 1024); p[i] += 128; }
    int buf = 0, nbuf = STAGES - 1;
    for (int t = 0; t < iters; ++t) {
        if (STAGES >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int i = 0; i < PIECES; ++i) { glds16(p[i], smem + nbuf * STAGE_BYTES + (wave * PIECES + i) * 1024); p[i] += 128; }
        acc += *reinterpret_cast<const float*>(smem + buf * STAGE_BYTES + wave * PIECES * 1024 + lane * 16);
        buf = (buf + 1 == STAGES) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
        if ((t & 63) == 63)  // stay inside the (L2-resident) span
            for (int i = 0; i < PIECES; ++i) p[i] -= 64 * 128;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NW, int STAGES, int PIECES, bool BARRIER>
void run(const char* src, float* out, int wg_per_cu, size_t span) {
    const int lds = STAGES * NW * PIECES * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<NW, STAGES, PIECES, BARRIER>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * wg_per_cu, iters = 4000;
    const int ld = 8192 + 128;  // row stride like a K = 4096 fp16 operand (+ a line, avoids channel aliasing)
    fill<NW, STAGES, PIECES, BARRIER><<<grid, NW * 64, lds>>>(src, ld, 64, span, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    fill<NW, STAGES, PIECES, BARRIER><<<grid, NW * 64, lds>>>(src, ld, iters, span, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * NW * PIECES * 1024.0;
    printf("waves/WG %2d  WG/CU %d  stages %d  pieces/wave %d  barrier %d  LDS/WG %3d KB : %7.2f TB/s aggregate, %6.1f GB/s per CU, "
           "%.2f us per stage\n", NW, wg_per_cu, STAGES, PIECES, (int)BARRIER, lds / 1024, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
           ms * 1e3 / iters);
}

int main(int argc, char** argv) {
    // source region: 24 MB (default) lives in the 256 MB Infinity Cache after the warm-up, but not in one XCD's 4 MB
    // L2; 1 MB (argv[1] = bytes) is L2 resident in every XCD
    const size_t span = argc > 1 ? (size_t)atoll(argv[1]) : (24u << 20);
    printf("source span %zu bytes\n", span);
    char* src;
    float* out;
    hipMalloc(&src, span + (1 << 20));
    hipMemset(src, 1, span + (1 << 20));
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    run<4, 2, 4, true>(src, out, 1, span);
    run<4, 2, 4, true>(src, out, 2, span);
    run<4, 2, 4, true>(src, out, 4, span);
    run<4, 3, 4, true>(src, out, 2, span);
    run<4, 4, 4, true>(src, out, 2, span);
    run<4, 2, 8, true>(src, out, 2, span);
    run<4, 3, 8, true>(src, out, 1, span);
    run<8, 2, 4, true>(src, out, 1, span);
    run<8, 2, 4, true>(src, out, 2, span);
    run<8, 3, 4, true>(src, out, 1, span);
    run<8, 2, 6, true>(src, out, 1, span);
    run<4, 2, 4, false>(src, out, 2, span);
    run<4, 4, 4, false>(src, out, 2, span);
    run<8, 4, 2, false>(src, out, 2, span);
    run<16, 2, 4, true>(src, out, 1, span);
    run<4, 2, 2, true>(src, out, 4, span);
    return 0;
}
