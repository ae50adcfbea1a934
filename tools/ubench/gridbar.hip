// Micro-benchmark behind the "persistent level kernel" question: what does a DEVICE-WIDE barrier inside one
// persistent kernel cost on MI355X (8 XCDs, private L2s), against the kernel boundary of a captured HIP graph (~1.7 us for a
// dependent chain of trivial kernels, profiles/r05_launch_floor.txt (a))?
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
// Every phase each workgroup reads the 256-byte slot ANOTHER workgroup (next id = another XCD) wrote in the previous phase,
// adds one and writes its own slot: the result is only right if the barrier also makes the data visible across XCDs.
//   mode 0  one kernel launch per phase, captured in a graph (the baseline)
//   mode 1  persistent kernel; barrier = agent-scope release fence (L2 write-back) + counter + acquire fence (L2 invalidate),
//           plain loads / stores for the data
//   mode 2  persistent kernel; data through agent-scope atomic loads / stores (sc1: the L2 is by-passed / written through per
//           access), barrier = counter only (no cache maintenance)
//   mode 3  as mode 1, but only ONE workgroup per XCD runs the fences (the others wait on it through the counter): the L2 is
//           a per-XCD resource, 32 write-back + invalidate requests per phase are redundant
//   mode 4  the cheapest barrier this chip allows, still without cache maintenance (data as in mode 2): HIERARCHICAL -- arrivals
//           counted per XCD by atomics that execute in that XCD's L2 (workgroup scope: no trip to the fabric), the last arriver
//           of each XCD bumps ONE device-scope counter (8 serialised fabric atomics instead of 256), polls it, and releases its
//           XCD through a second L2-local word.  MEASURED: never completes (spin limit) -- arrivals of one XCD do not meet in
//           one L2 word under workgroup scope / the blockIdx % 8 placement is not something to build correctness on.  Kept for
//           the record, run only with a third command-line argument.
// Reports microseconds per phase and whether every slot holds the phase count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int SLOT = 64;  // floats per workgroup slot (256 bytes = two cache lines)
constexpr unsigned SPIN_LIMIT = 1u << 22;

__global__ void __launch_bounds__(256) phase_kernel(const float* __restrict__ in, float* __restrict__ out, int nwg) {
    const int src = (blockIdx.x + 1) % nwg;
    if (threadIdx.x < SLOT) out[blockIdx.x * SLOT + threadIdx.x] = in[src * SLOT + threadIdx.x] + 1.0f;
}

template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const bool fence = MODE == 1 || (MODE == 3 && blockIdx.x < 8);
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's write-through stores are out
        if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2 sc1 + waits
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
        }
        if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1
    }
    __syncthreads();
    return ok;
}

// mode 4: ctr[0] = device counter; ctr[64 + 64 * xcd] = arrivals of the XCD; ctr[64 + 64 * xcd + 32] = its release word
__device__ __forceinline__ bool hier_barrier(unsigned* ctr, unsigned phase1, unsigned nwg, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned xcd = blockIdx.x & 7u, nloc = (nwg - xcd + 7u) >> 3;  // workgroups the dispatcher put on this XCD
        unsigned* arrive = ctr + 64 + 64 * xcd;
        unsigned* release = arrive + 32;
        const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        if (old + 1 == phase1 * nloc) {  // last of this XCD
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1 * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
            }
            __hip_atomic_fetch_add(release, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while (__hip_atomic_fetch_add(release, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < phase1) {  // RMW: read at the L2
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
            }
        }
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ void __launch_bounds__(256) persistent_kernel(float* buf0, float* buf1, int nwg, int phases, unsigned* ctr, unsigned* err) {
    const int src = (blockIdx.x + 1) % nwg;
    // mode 3: a second counter orders "every workgroup arrived" before the XCD leaders' write-back and "leaders done" after
    for (int p = 0; p < phases; ++p) {
        const float* in = (p & 1) ? buf1 : buf0;
        float* out = (p & 1) ? buf0 : buf1;
        if (threadIdx.x < SLOT) {
            if (MODE == 2 || MODE == 4) {
                const float v = __hip_atomic_load(in + src * SLOT + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out + blockIdx.x * SLOT + threadIdx.x, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                out[blockIdx.x * SLOT + threadIdx.x] = in[src * SLOT + threadIdx.x] + 1.0f;
            }
        }
        if (MODE == 3) {
            // (a) everybody's stores are in their L2 (vmcnt(0)) -> counter A; (b) the 8 leaders (one per XCD: blockIdx % 8 is the XCD)
            // wait for A, write back + invalidate, bump counter B; (c) everybody waits for B
            __syncthreads();
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                if (blockIdx.x < 8) {
                    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p + 1) * nwg) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT) { *err = 1; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this XCD's dirty lines (every CU's) -> memory
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // and its stale lines dropped
                    __hip_atomic_fetch_add(ctr + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                spins = 0;
                while (__hip_atomic_load(ctr + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p + 1) * 8u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                }
                asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1 only (the leaders took care of the L2)
            }
            __syncthreads();
            if (*err) return;
        } else if (MODE == 4) {
            if (!hier_barrier(ctr, (unsigned)(p + 1), (unsigned)nwg, err)) return;
        } else {
            if (!grid_barrier<MODE>(ctr, (unsigned)(p + 1) * nwg, err)) return;
            if (*err) return;
        }
    }
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256, phases = argc > 2 ? atoi(argv[2]) : 200, reps = 20;
    float *b0, *b1;
    unsigned *ctr, *err;
    CK(hipMalloc(&b0, nwg * SLOT * 4));
    CK(hipMalloc(&b1, nwg * SLOT * 4));
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&err, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < (argc > 3 ? 5 : 4); ++mode) {  // mode 4 only on request (third argument): it spins into its limit
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(b0, 0, nwg * SLOT * 4, s));
        CK(hipMemsetAsync(b1, 0, nwg * SLOT * 4, s));
        CK(hipMemsetAsync(ctr, 0, 4096, s));
        CK(hipMemsetAsync(err, 0, 4, s));
        if (mode == 0) {
            for (int p = 0; p < phases; ++p)
                hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, s, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, nwg);
        } else if (mode == 1) {
            hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(256), 0, s, b0, b1, nwg, phases, ctr, err);
        } else if (mode == 2) {
            hipLaunchKernelGGL(persistent_kernel<2>, dim3(nwg), dim3(256), 0, s, b0, b1, nwg, phases, ctr, err);
        } else if (mode == 4) {
            hipLaunchKernelGGL(persistent_kernel<4>, dim3(nwg), dim3(256), 0, s, b0, b1, nwg, phases, ctr, err);
        } else {
            hipLaunchKernelGGL(persistent_kernel<3>, dim3(nwg), dim3(256), 0, s, b0, b1, nwg, phases, ctr, err);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> h(nwg * SLOT);
        CK(hipMemcpy(h.data(), (phases & 1) ? b1 : b0, nwg * SLOT * 4, hipMemcpyDeviceToHost));
        unsigned herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (float v : h) bad += (v != (float)phases);
        printf("{\"mode\": %d, \"workgroups\": %d, \"phases\": %d, \"us_per_phase\": %.3f, \"wrong_slots\": %d, \"spin_timeout\": %u}\n", mode, nwg,
               phases, ms * 1000.f / reps / phases, bad, herr);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
