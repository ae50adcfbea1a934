// Micro-benchmark: what one CU can pull global -> LDS with LDS-DMA (`buffer_load_dwordx4 ... lds`) when EVERY CU streams at
// once -- the delivery side of the conv / chain kernels' per-CU budget ("Conv, second half of round 3").
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip && ./ldsdma_rate
// Patterns:
//   shared : all workgroups stream the SAME region (a weight stream: L2 hits after the first reader of an XCD)
//   private: workgroup w streams its own region (activations: every byte is fetched once from HBM / Infinity Cache)
// One workgroup of W waves per CU (100 KB of LDS requested), each wave copies 1 KiB pieces into a 2-slot ring and waits
// with counted vmcnt, no compute.  Reports GB/s aggregate and bytes per clock per CU (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k(const char* base, size_t region, int shared, int stages, long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 40960;               // bytes per stage (the kernels' weight stage image)
    constexpr int PIECES = STAGE / 1024 / WAVES;  // pieces per wave per stage
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* src = base + (shared ? 0 : (size_t)blockIdx.x * region);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (unsigned)region, 0x00020000);
    const int voff = lane * 16;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const unsigned so = (unsigned)(((size_t)s * STAGE) % (region - STAGE)) & ~1023u;
        char* dst = smem + (s & 1) * STAGE + wave * PIECES * 1024;
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, voff,
                                                     so + (wave * PIECES + i) * 1024, 0, 0);
        if (s > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // the previous stage has landed
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int WAVES>
void run(const char* buf, size_t region, int shared, long long* ticks, int blocks) {
    const int stages = 400;
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<WAVES><<<blocks, WAVES * 64, lds>>>(buf, region, shared, 20, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<WAVES><<<blocks, WAVES * 64, lds>>>(buf, region, shared, stages, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    double t = 0;
    for (auto v : h) t += (double)v;
    t /= blocks;
    const double bytes = (double)stages * 40960.0;
    printf("%-7s %d waves/CU: %7.3f ms, %7.1f GB/s aggregate, %6.1f GB/s per CU, %5.1f B per s_memtime tick per CU\n",
           shared ? "shared" : "private", WAVES, ms, bytes * blocks / ms / 1e6, bytes / ms / 1e6, bytes / t);
}

int main() {
    const int blocks = 256;
    const size_t region = 4u << 20;  // 4 MiB per workgroup (private) / in total (shared): L2-sized working set per XCD
    char* buf;
    long long* ticks;
    hipMalloc(&buf, region * blocks);
    hipMemset(buf, 1, region * blocks);
    hipMalloc(&ticks, blocks * 8);
    for (int shared : {1, 0}) {
        run<4>(buf, region, shared, ticks, blocks);
        run<8>(buf, region, shared, ticks, blocks);
        run<10>(buf, region, shared, ticks, blocks);
    }
    return 0;
}
