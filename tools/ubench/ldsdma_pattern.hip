// Micro-benchmark (round 4): global -> LDS delivery rate of ONE CU, every CU streaming, as a function of the ADDRESS
// PATTERN of the LDS-DMA pieces -- the conv / GEMM loaders copy row SEGMENTS (128 B or 64 B of a row whose neighbours
// are a leading dimension apart), not contiguous KiB like tools/ubench/ldsdma_rate.hip.  The ablation builds of the
// ping-pong kernel (tools/experiments/r04_run3.sh) deliver only ~45-55 GB/s per CU with no compute at all; which part of the pattern
// costs that?
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_pattern ldsdma_pattern.hip && ./ldsdma_pattern
// Patterns (8 waves per CU, 28 pieces of 1 KiB per stage, ring of NSLOT stages, counted vmcnt, one barrier per stage):
//   seg = bytes of one row segment (1024 = contiguous piece, 128, 64); ld = row pitch in bytes; the region is shared by
//   all workgroups (weights) or private per workgroup (activations) or 20 shared + 8 private pieces (the 128x320 tile's mix);
//   addr = "buf" (buffer_load ... lds, SGPR base + 32-bit lane offset) or "flat" (global_load_lds, 64-bit lane address).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct Cfg { int seg, ld, mode /*0 shared, 1 private, 2 mix*/, flat, nslot; };

template <int NSLOT>
__global__ void __launch_bounds__(512, 1) k(const char* base, size_t region, Cfg c, int stages, long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECES = 28, STAGE = PIECES * 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* shr = base;
    const char* prv = base + (size_t)(1 + blockIdx.x) * region;
    const rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(shr), 0, (unsigned)region, 0x00020000);
    const rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(prv), 0, (unsigned)region, 0x00020000);
    // lane -> (row of the piece, byte inside the segment)
    const int lanes_per_row = c.seg / 16;
    const int rows_per_piece = 1024 / c.seg;
    const int row = lane / lanes_per_row, col = (lane % lanes_per_row) * 16;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned span = (unsigned)region - (unsigned)(PIECES * rows_per_piece + 8) * (unsigned)c.ld - 4096u;
    for (int s = 0; s < stages; ++s) {
        // stage s reads segment number s of every row (the K walk), wrapping inside the region
        const unsigned kofs = ((unsigned)s * (unsigned)c.seg) % (unsigned)c.ld;
        const unsigned wrap = (((unsigned)s * (unsigned)c.seg) / (unsigned)c.ld) * (unsigned)(PIECES * rows_per_piece) * (unsigned)c.ld;
        char* dst = smem + (s % NSLOT) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = i * 8 + wave;
            if (pc < PIECES) {
                const bool priv = c.mode == 1 || (c.mode == 2 && pc >= 20);
                const unsigned off = ((wrap % span) + (unsigned)(pc * rows_per_piece + row) * (unsigned)c.ld + kofs + col);
                if (c.flat) {
                    const char* g = (priv ? prv : shr) + off;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
                } else if (priv) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, off, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, off, 0, 0, 0);
                }
            }
        }
        // leave NSLOT - 2 newer stages in flight (waves 0-3 issue 4 pieces per stage, 4-7 issue 3)
        if (s >= NSLOT - 2) {
            if (NSLOT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NSLOT == 3) { if (wave < 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
            else if (NSLOT == 5) { if (wave < 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NSLOT>
void run(const char* buf, size_t region, Cfg c, long long* ticks, int blocks) {
    const int stages = 400;
    const size_t lds = (size_t)NSLOT * 28 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<NSLOT><<<blocks, 512, lds>>>(buf, region, c, 20, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NSLOT><<<blocks, 512, lds>>>(buf, region, c, stages, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    double t = 0;
    for (auto v : h) t += (double)v;
    t /= blocks;
    const double bytes = (double)stages * 28.0 * 1024.0;
    const char* modes[] = {"shared", "private", "mix20+8"};
    printf("seg %4d ld %5d %-8s %-4s ring %d: %7.3f ms, %6.1f GB/s per CU, %5.1f B per tick per CU, %5.2f us per 28-KiB stage\n", c.seg, c.ld,
           modes[c.mode], c.flat ? "flat" : "buf", NSLOT, ms, bytes / ms / 1e6, bytes / t, ms * 1e3 / stages);
}

int main() {
    const int blocks = 256;
    const size_t region = 4u << 20;
    char* buf;
    long long* ticks;
    hipMalloc(&buf, region * (blocks + 1));
    hipMemset(buf, 1, region * (blocks + 1));
    hipMalloc(&ticks, blocks * 8);
    for (int mode : {0, 2, 1})
        for (int flat : {0, 1})
            for (Cfg c : {Cfg{1024, 1024, mode, flat, 0}, Cfg{128, 640, mode, flat, 0}, Cfg{128, 5760, mode, flat, 0},
                          Cfg{64, 640, mode, flat, 0}, Cfg{64, 5760, mode, flat, 0}}) {
                run<3>(buf, region, c, ticks, blocks);
                if (c.seg != 1024 && c.ld == 5760) { run<2>(buf, region, c, ticks, blocks); run<5>(buf, region, c, ticks, blocks); }
            }
    return 0;
}
