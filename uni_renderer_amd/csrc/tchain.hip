// Row-local transformer chains at C = 320 (the 64x64 level of the SD-1.x UNets): several GEMMs, the LayerNorm between
// them, the GEGLU feed-forward and the residual adds of one BasicTransformerBlock / Transformer2DModel tail as ONE
// launch.  Reference semantics: diffusers 0.24 BasicTransformerBlock as instantiated by models/unet_2d_blocks.py:1115-1126
// (SURVEY.md rows a15 / a16): x += attn(LN(x)); x += FF(LN3(x)); out = proj_out(x) + block input.
//
// Why: at this level every projection / feed-forward GEMM has M = 2 x 16384 rows and K = 320 ..
// 1280: 6.7 - 54 GFLOP over 60 - 150 MB of activations, i.e. below the ridge -- each launch streams its activation
// matrix through HBM / the Infinity Cache twice and pays a fill + drain.  Here the rows never leave the CU:
//
//   * a workgroup owns 128 token rows, a wave 32 of them, and a lane ONE row (pixel) for half of the channels;
//   * activations live in REGISTERS for the whole chain: with v_mfma_f32_32x32x16 computing Y^T = W . X^T (weights =
//     operand A, rows = output channels; activations = operand B, columns = pixels) the accumulator block of a lane
//     (rows 8q + 4h + r of column `pixel`) is, after fp16/bf16 packing, exactly a B operand of the next GEMM under a
//     fixed permutation of its k index -- which the host folds into the weight columns (tchain.py: KPERM).  LayerNorm,
//     bias, GEGLU and the residual adds are lane-local (+ one exchange with lane ^ 32);
//   * only WEIGHTS move: the host lays every matrix of a chain out as a sequence of 40-KiB LDS stage images (rows x
//     64 k, 16-byte chunks XOR-swizzled by (row >> 1) & 7 for conflict-free ds_read_b128 fragment reads), the four waves
//     copy them global -> LDS with `buffer_load ... lds` (SGPR offsets only, no address VALU) into a 3-slot ring,
//     prefetch distance 2, one s_barrier per stage, counted vmcnt;
//   * 4 waves per workgroup = one per SIMD with up to 512 VGPRs each (accumulator 160 + operand 80 + FF tiles).
//
// Global I/O happens only at the ends of a chain, 16 bytes per lane; the accumulator layout (4 consecutive channels per
// register quad) is converted to / from 8 consecutive channels per lane with v_permlane32_swap.
#include "ur_common.h"
#include "../../include/ur_kernels.h"
#include "tchain_asm.inc"

namespace ur {

constexpr int TC_C = 320;                  // channels of the level this kernel is built for
constexpr int TC_NT = TC_C / 32;           // 10 output-channel tiles of 32
constexpr int TC_KS = TC_C / 16;           // 20 k16 steps over the channels
constexpr int TC_STAGE = 40960;            // bytes of one weight-stream stage image (320 rows x 128 B)
constexpr int TC_NSLOT = 3;                // LDS ring slots
constexpr int TC_PIECES = TC_STAGE / 1024 / 4;  // LDS-DMA instructions (1 KiB each) per wave per stage
constexpr int TC_FF = 4 * TC_C;            // GEGLU hidden width
constexpr int TC_RING = TC_NSLOT * TC_STAGE;

// const-vector offsets (floats) inside the per-z block the host builds (tchain.py)
constexpr int TCC_BIAS0 = 0, TCC_GAMMA = 320, TCC_BETA = 640, TCC_Q_END = 960;
constexpr int TCC_B1V = 960, TCC_B1G = 960 + TC_FF, TCC_B2 = 960 + 2 * TC_FF, TCC_BPO = TCC_B2 + 320, TCC_FF_END = TCC_BPO + 320;

typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <typename T> struct Acc { f32x16 t[TC_NT]; };

__device__ __forceinline__ void swap32(float& upper_of_a, float& lower_of_b) {
    // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second, both registers updated in
    // place.  Inline asm on purpose: through __builtin_amdgcn_permlane32_swap with float operands hipcc (ROCm 7.2)
    // returned the FIRST result in both elements (tools/ubench/permlane_probe.hip: fb == fa), while the instruction
    // itself does what the ISA says.  The s_nops cover the VALU-write -> permlane-swap wait states the compiler would
    // have inserted for its own instruction (I/O phases only: 240 swaps per lane per launch).
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(upper_of_a), "+v"(lower_of_b));
}

// D arrangement of a 16-channel group (f[0..3] = rows 4h + r of the even 8-row block, f[4..7] = of the odd one) <->
// 8 consecutive channels 8h .. 8h + 7 per lane.  An involution: the same four swaps both ways.
__device__ __forceinline__ void d_mem_swap(float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) swap32(f[i], f[4 + i]);
}

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type pack8(const float (&f)[8]) {
    typename Vec8<T>::type v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
    return v;
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256, 1) tchain_kernel(const ur_tchain_desc p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int NSTAGES = MODE == UR_TCHAIN_Q ? 10 : 5 + 3 * (TC_FF / 64) + 5;
    constexpr int NCONST = MODE == UR_TCHAIN_Q ? TCC_Q_END : TCC_FF_END;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cst = reinterpret_cast<float*>(smem + TC_RING);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int tiles = (p.M + 127) >> 7;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD works on one z (one weight stream in its L2)
    const int z = lid / tiles, tile = lid - z * tiles;
    const int m = tile * 128 + wave * 32 + l31;
    const bool row_ok = m < p.M;
    const int64_t mrow = (int64_t)z * p.M + (row_ok ? m : p.M - 1);  // clamped: every lane loads, only valid rows store

    // optional diagnostics: s_memtime stamps of wave 0 of every workgroup ([blocks][16] int64), tools/tchain_bench.py --profile
    long long* prof = p.profile ? reinterpret_cast<long long*>(p.profile) + (int64_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if (prof && tid == 0) prof[i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- constants -> LDS (before any LDS-DMA is in flight: plain loads + ds_write + one ordinary barrier) ----
    {
        const float* src = p.consts + (int64_t)z * p.z_consts;
        for (int i = tid; i < NCONST; i += 256) cst[i] = src[i];
    }
    __syncthreads();

    // ---- weight stream: 3-slot ring, prefetch distance 2 ----
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.wstream)) + (int64_t)z * p.z_wstream, 0, NSTAGES * TC_STAGE, 0x00020000);
    const int voff = lane * 16;
    int issued = 0;   // stages whose copies this wave has issued
    int cons = 0;     // stages consumed
    // Every workgroup streams the SAME images in the same order, and the workgroups of an XCD run in lock step: issued in
    // image order, all 32 CUs of an XCD would ask the same L2 channel for the same kilobyte at the same moment (measured:
    // 4700 cycles per stage = 18 GB/s per CU).  So the 40 pieces of a stage are walked from a workgroup-specific starting
    // point: wave w takes pieces (4 i + w + rot) mod 40.
    const int rot = __builtin_amdgcn_readfirstlane((tile * 7 + z * 3) % 40);
    auto issue = [&]() __attribute__((always_inline)) {
        if (issued < NSTAGES) {
            const int slot = issued % TC_NSLOT;
            const int sbase = issued * TC_STAGE;
            char* dst = smem + slot * TC_STAGE;
            int pc = rot + wave;
#pragma unroll
            for (int i = 0; i < TC_PIECES; ++i) {
                if (pc >= 40) pc -= 40;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, voff,
                                                         sbase + pc * 1024, 0, 0);
                pc += 4;
            }
        }
        issued += 1;
    };
    // wait until stage `cons` has landed for everybody, free the slot of stage cons - 1, refill it with stage cons + 2
    auto next_stage = [&]() __attribute__((always_inline)) -> int {
        if (cons + 1 < NSTAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TC_PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const int s = cons % TC_NSLOT;
        cons += 1;
        return s;
    };
    issue();
    issue();
    stamp(1);

    // fragment address of (row l31 of a 32-row block, k16 step s of the stage's 64-k chunk)
    const int key = (l31 >> 1) & 7;
    auto afrag = [&](const char* base, int s) __attribute__((always_inline)) {
        return *reinterpret_cast<const vec8*>(base + l31 * 128 + (((2 * s + hh) ^ key) << 4));
    };
    // per-channel fp32 vector in LDS, accumulator arrangement: 4 consecutive channels 32 t + 8 q + 4 h + r
    auto cvec = [&](int off, int t, int q) __attribute__((always_inline)) {
        return *reinterpret_cast<const float4*>(cst + off + 32 * t + 8 * q + 4 * hh);
    };

    // ---- global I/O helpers (16 bytes per lane; rows are 320 channels wide) ----
    const T* a0 = reinterpret_cast<const T*>(p.a0) + mrow * TC_C;
    // residual-stream tensor (hi [+ lo]) -> fp32 in the accumulator arrangement, ADDED to acc
    auto add_stream = [&](Acc<T>& acc, const void* hi_, const void* lo_) __attribute__((always_inline)) {
        const T* hi = reinterpret_cast<const T*>(hi_) + mrow * TC_C;
        const lo_t<T>* lo = lo_ ? reinterpret_cast<const lo_t<T>*>(lo_) + mrow * TC_C : nullptr;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 32 * t + 16 * g + 8 * hh;
                float f[8];
                load8(hi + c, f);
                if (lo) {
                    float l[8];
                    load_lo<8>(lo + c, l);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] += l[i];
                }
                d_mem_swap(f);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc.t[t][8 * g + i] += f[i];
                    acc.t[t][8 * g + 4 + i] += f[4 + i];
                }
            }
    };
    // accumulator -> global: hi (+ lo when asked)
    auto store_stream = [&](const Acc<T>& acc, void* hi_, void* lo_) __attribute__((always_inline)) {
        T* hi = reinterpret_cast<T*>(hi_) + mrow * TC_C;
        lo_t<T>* lo = lo_ ? reinterpret_cast<lo_t<T>*>(lo_) + mrow * TC_C : nullptr;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 32 * t + 16 * g + 8 * hh;
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[i] = acc.t[t][8 * g + i];
                    f[4 + i] = acc.t[t][8 * g + 4 + i];
                }
                d_mem_swap(f);
                if (row_ok) {
                    store8(hi + c, f);
                    if (lo) store_lo8<T>(lo + c, f);
                }
            }
    };
    auto add_cvec = [&](Acc<T>& acc, int off) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = cvec(off, t, q);
                acc.t[t][4 * q + 0] += b.x; acc.t[t][4 * q + 1] += b.y; acc.t[t][4 * q + 2] += b.z; acc.t[t][4 * q + 3] += b.w;
            }
    };
    auto zero = [&](Acc<T>& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc.t[t][v] = 0.f;
    };
    // accumulator block -> B operands of the next GEMM: tile t gives k16 steps 2t (row blocks q = 0, 1) and 2t + 1
    // (q = 2, 3); logical k = 8 h + i of a step is channel [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15][8 h + i] of its
    // 16-group -- the KPERM column order of the consuming weight images
    auto to_operand = [&](const Acc<T>& acc, vec8 (&bop)[TC_KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i) bop[2 * t + g][i] = (T)acc.t[t][8 * g + i];
    };
    // LDS byte address of this lane's A fragment for k16 step s inside ring slot `slot` (the row-tile / sub-image offset
    // is an instruction immediate): loop invariant, 12 VGPRs
    unsigned fa[TC_NSLOT][4];
#pragma unroll
    for (int sl = 0; sl < TC_NSLOT; ++sl)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            fa[sl][s4] = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + sl * TC_STAGE + l31 * 128 +
                         (((2 * s4 + hh) ^ key) << 4);
    // one N = 320 GEMM stage (rows x 64 k): acc[t] += W[32 t .. 32 t + 31][k] * operand[k], 40 MFMAs, as ONE hand-scheduled
    // instruction stream (tchain_asm.inc): six fragment reads in flight, counted lgkmcnt, a read behind every MFMA
    auto gemm_stage = [&](Acc<T>& acc, int slot, const vec8& b0, const vec8& b1, const vec8& b2, const vec8& b3) __attribute__((always_inline)) {
        vec8 f0, f1, f2, f3, f4, f5, f6, f7;
#define TC_GEMM_OPERANDS                                                                                                       \
        : [c0] "+a"(acc.t[0]), [c1] "+a"(acc.t[1]), [c2] "+a"(acc.t[2]), [c3] "+a"(acc.t[3]), [c4] "+a"(acc.t[4]),             \
          [c5] "+a"(acc.t[5]), [c6] "+a"(acc.t[6]), [c7] "+a"(acc.t[7]), [c8] "+a"(acc.t[8]), [c9] "+a"(acc.t[9]),             \
          [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6),      \
          [f7] "=&v"(f7)                                                                                                       \
        : [a0] "v"(fa[slot][0]), [a1] "v"(fa[slot][1]), [a2] "v"(fa[slot][2]), [a3] "v"(fa[slot][3]), [b0] "v"(b0),            \
          [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3)
        if constexpr (sizeof(T) == 2 && __is_same(T, f16)) asm volatile(TC_ASM_GEMM_STAGE("f16") TC_GEMM_OPERANDS);
        else asm volatile(TC_ASM_GEMM_STAGE("bf16") TC_GEMM_OPERANDS);
#undef TC_GEMM_OPERANDS
    };
    // compiler code that touches the accumulators after an asm stage: an MFMA result may be read 12+ states after issue
    auto acc_fence = [&](Acc<T>& acc) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc.t[0]), "+a"(acc.t[1]), "+a"(acc.t[2]), "+a"(acc.t[3]), "+a"(acc.t[4]), "+a"(acc.t[5]),
                       "+a"(acc.t[6]), "+a"(acc.t[7]), "+a"(acc.t[8]), "+a"(acc.t[9]));
    };
    auto gemm320 = [&](Acc<T>& acc, const vec8 (&bop)[TC_KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < TC_C / 64; ++kc) {
            const int slot = next_stage();
            // the slot index is a compile-time constant only when the stage counter is; select the address set without
            // dynamic register indexing
            if (slot == 0) gemm_stage(acc, 0, bop[4 * kc], bop[4 * kc + 1], bop[4 * kc + 2], bop[4 * kc + 3]);
            else if (slot == 1) gemm_stage(acc, 1, bop[4 * kc], bop[4 * kc + 1], bop[4 * kc + 2], bop[4 * kc + 3]);
            else gemm_stage(acc, 2, bop[4 * kc], bop[4 * kc + 1], bop[4 * kc + 2], bop[4 * kc + 3]);
        }
        acc_fence(acc);
    };

    // =============================== leading GEMM: y = a0 W0^T + bias0 + residual ===============================
    Acc<T> acc;
    vec8 bop[TC_KS];
#pragma unroll
    for (int s = 0; s < TC_KS; ++s) bop[s] = *reinterpret_cast<const vec8*>(a0 + 16 * s + 8 * hh);
    zero(acc);
    add_stream(acc, p.res, p.res_lo);
    add_cvec(acc, TCC_BIAS0);
    stamp(2);
    gemm320(acc, bop);
    stamp(3);
    if constexpr (MODE == UR_TCHAIN_Q) {
        // the updated residual stream leaves here; the LDS-DMA queue is drained first so that the stores are the
        // only vector-memory operations counted between the two GEMM passes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_stream(acc, p.y_out, p.y_out_lo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    stamp(4);

    // =============================== LayerNorm (exact two-pass, fp32) -> operand ===============================
    {
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) s1 += acc.t[t][v];
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * (1.0f / TC_C);
        float s2 = 0.f;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float d = acc.t[t][v] - mean;
                s2 = fmaf(d, d, s2);
            }
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 * (1.0f / TC_C) + p.eps);
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 g = cvec(TCC_GAMMA, t, q), b = cvec(TCC_BETA, t, q);
                const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bop[2 * t + (q >> 1)][4 * (q & 1) + r] = (T)fmaf((acc.t[t][4 * q + r] - mean) * rstd, gg[r], bb[r]);
            }
    }

    stamp(5);
    if constexpr (MODE == UR_TCHAIN_Q) {
        // =============================== q = LN(y) Wq^T (scale folded into Wq by the host) ===============================
        zero(acc);
        gemm320(acc, bop);
        stamp(6);
        store_stream(acc, p.out, nullptr);
        stamp(7);
    } else {
        // =============================== GEGLU feed-forward: acc = y + b2 + sum_j h_j W2_j^T ===============================
        add_cvec(acc, TCC_B2);
        for (int j = 0; j < TC_FF / 64; ++j) {
            vec8 hb[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // stage image: five [64 rows][64 k] sub-images; rows 0..31 = value rows, 32..63 = gate rows of 32 hidden units
                const int slot = next_stage();
                f32x16 hv0, hg0, hv1, hg1;
#pragma unroll
                for (int v = 0; v < 16; ++v) { hv0[v] = 0.f; hg0[v] = 0.f; hv1[v] = 0.f; hg1[v] = 0.f; }
                {
                    vec8 f0, f1, f2, f3, f4, f5, f6, f7;
                    const unsigned a0_ = slot == 0 ? fa[0][0] : (slot == 1 ? fa[1][0] : fa[2][0]);
                    const unsigned a1_ = slot == 0 ? fa[0][1] : (slot == 1 ? fa[1][1] : fa[2][1]);
                    const unsigned a2_ = slot == 0 ? fa[0][2] : (slot == 1 ? fa[1][2] : fa[2][2]);
                    const unsigned a3_ = slot == 0 ? fa[0][3] : (slot == 1 ? fa[1][3] : fa[2][3]);
#define TC_FFA_OPERANDS                                                                                                        \
        : [hv0] "+a"(hv0), [hg0] "+a"(hg0), [hv1] "+a"(hv1), [hg1] "+a"(hg1), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), \
          [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6), [f7] "=&v"(f7)                                      \
        : [a0] "v"(a0_), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_), [b0] "v"(bop[0]), [b1] "v"(bop[1]), [b2] "v"(bop[2]),    \
          [b3] "v"(bop[3]), [b4] "v"(bop[4]), [b5] "v"(bop[5]), [b6] "v"(bop[6]), [b7] "v"(bop[7]), [b8] "v"(bop[8]),          \
          [b9] "v"(bop[9]), [b10] "v"(bop[10]), [b11] "v"(bop[11]), [b12] "v"(bop[12]), [b13] "v"(bop[13]),                   \
          [b14] "v"(bop[14]), [b15] "v"(bop[15]), [b16] "v"(bop[16]), [b17] "v"(bop[17]), [b18] "v"(bop[18]), [b19] "v"(bop[19])
                    if constexpr (__is_same(T, f16)) asm volatile(TC_ASM_FFA_STAGE("f16") "s_nop 15\n\ts_nop 15" TC_FFA_OPERANDS);
                    else asm volatile(TC_ASM_FFA_STAGE("bf16") "s_nop 15\n\ts_nop 15" TC_FFA_OPERANDS);
#undef TC_FFA_OPERANDS
                }
                const int hid = 64 * j + 32 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(cst + TCC_B1V + hid + 8 * q + 4 * hh);
                    const float4 bg = *reinterpret_cast<const float4*>(cst + TCC_B1G + hid + 8 * q + 4 * hh);
                    const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgg[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float val = hv0[4 * q + r] + hv1[4 * q + r] + bvv[r];
                        const float gate = hg0[4 * q + r] + hg1[4 * q + r] + bgg[r];
                        hb[2 * half + (q >> 1)][4 * (q & 1) + r] = (T)(val * gelu_erf_f(gate));
                    }
                }
            }
            const int slot = next_stage();  // W2[:, 64 j .. 64 j + 63]
            if (slot == 0) gemm_stage(acc, 0, hb[0], hb[1], hb[2], hb[3]);
            else if (slot == 1) gemm_stage(acc, 1, hb[0], hb[1], hb[2], hb[3]);
            else gemm_stage(acc, 2, hb[0], hb[1], hb[2], hb[3]);
        }
        acc_fence(acc);
        stamp(6);
        // =============================== out = y3 Wpo^T + bpo + block input ===============================
        to_operand(acc, bop);
        zero(acc);
        gemm320(acc, bop);
        stamp(7);
        add_cvec(acc, TCC_BPO);
        add_stream(acc, p.blk, p.blk_lo);
        stamp(8);
        store_stream(acc, p.out, p.out_lo);
        stamp(9);
    }
}

template <typename T, int MODE>
static int launch_tchain(const ur_tchain_desc& d, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    const int nconst = MODE == UR_TCHAIN_Q ? TCC_Q_END : TCC_FF_END;
    const int lds = TC_RING + nconst * 4;
    set_lds_limit_once(done, reinterpret_cast<const void*>(&tchain_kernel<T, MODE>), lds);
    const int tiles = (d.M + 127) / 128;
    hipLaunchKernelGGL((tchain_kernel<T, MODE>), dim3(tiles * d.zbatch), dim3(256), lds, s, d);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace ur

extern "C" int ur_tchain(const ur_tchain_desc* din, void* stream) {
    using namespace ur;
    if (!din) return UR_E_BADARG;
    ur_tchain_desc d = *din;
    if (!d.a0 || !d.res || !d.out || !d.wstream || !d.consts || d.M <= 0) return UR_E_BADARG;
    if (d.zbatch < 1) d.zbatch = 1;
    if (d.mode == UR_TCHAIN_Q) {
        if (!d.y_out) return UR_E_BADARG;
        if (d.z_consts < TCC_Q_END || d.z_wstream < 10 * (int64_t)TC_STAGE) return UR_E_BADARG;
    } else if (d.mode == UR_TCHAIN_FF) {
        if (!d.blk) return UR_E_BADARG;
        if (d.z_consts < TCC_FF_END || d.z_wstream < (10 + 3 * (TC_FF / 64)) * (int64_t)TC_STAGE) return UR_E_BADARG;
    } else {
        return UR_E_BADARG;
    }
    if (d.channels != TC_C) return UR_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d.wstream) | (uintptr_t)d.z_wstream) & 15) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.dtype == UR_DT_F16) return d.mode == UR_TCHAIN_Q ? launch_tchain<f16, UR_TCHAIN_Q>(d, s) : launch_tchain<f16, UR_TCHAIN_FF>(d, s);
    if (d.dtype == UR_DT_BF16) return d.mode == UR_TCHAIN_Q ? launch_tchain<bf16, UR_TCHAIN_Q>(d, s) : launch_tchain<bf16, UR_TCHAIN_FF>(d, s);
    return UR_E_BADARG;
}

extern "C" int ur_sizeof_tchain_desc(void) { return (int)sizeof(ur_tchain_desc); }
extern "C" int64_t ur_tchain_stream_bytes(int mode) {
    return (int64_t)ur::TC_STAGE * (mode == UR_TCHAIN_Q ? 10 : 10 + 3 * (ur::TC_FF / 64));
}
extern "C" int ur_tchain_const_floats(int mode) { return mode == UR_TCHAIN_Q ? ur::TCC_Q_END : ur::TCC_FF_END; }
