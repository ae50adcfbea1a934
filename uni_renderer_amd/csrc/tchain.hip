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
#include <type_traits>
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
constexpr int TC_LDS = 160 * 1024;         // the whole CU's LDS: one workgroup per CU
// I/O staging: a wave's 32-row tile is ONE contiguous 20-KiB block of global memory; it is moved with fully coalesced
// 16-byte-per-lane accesses and transposed to / from the one-row-per-lane accumulator arrangement through a private LDS
// region (rows padded by 16 bytes: conflict-free 16-byte reads down a column of rows).  The regions lie behind ring
// slot 0 (which keeps prefetching), so staging happens only while slots 1 and 2 are not in use.
constexpr int TC_IO_HI_ROW = TC_C * 2 + 16;          // padded row of the hi tile (bytes)
constexpr int TC_IO_BASE = TC_STAGE;                 // first byte of the staging area

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
    constexpr int NSTAGES = MODE == UR_TCHAIN_Q ? 10 : (MODE == UR_TCHAIN_PRE ? 20 : 5 + 3 * (TC_FF / 64) + 5);
    constexpr int NCONST = MODE == UR_TCHAIN_FF ? TCC_FF_END : TCC_Q_END;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cst = reinterpret_cast<float*>(smem + TC_LDS - NCONST * 4);  // the constant vectors sit at the very end
    constexpr int IO_WAVE = ((TC_LDS - NCONST * 4 - TC_IO_BASE) / 4) & ~15;  // staging bytes per wave
    static_assert(IO_WAVE >= 32 * TC_IO_HI_ROW, "staging region too small");


    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    char* const io = smem + TC_IO_BASE + wave * IO_WAVE;  // this wave's staging slice
    const int tiles = (p.M + 127) >> 7;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD works on one z (one weight stream in its L2)
    const int z = lid / tiles, tile = lid - z * tiles;
    const int m = tile * 128 + wave * 32 + l31;
    const bool row_ok = m < p.M;
    const int64_t mrow = (int64_t)z * p.M + (row_ok ? m : p.M - 1);  // clamped: every lane loads, only valid rows store

    // optional diagnostics: s_memtime stamps of wave 0 of every workgroup ([blocks][64] int64), tools/tchain_bench.py --profile
    long long* prof = p.profile ? reinterpret_cast<long long*>(p.profile) + (int64_t)blockIdx.x * 64 : nullptr;
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
    // Every workgroup streams the SAME images in the same order; the 40 pieces of a stage are walked from a
    // workgroup-specific starting point (wave w takes pieces (4 i + w + rot) mod 40) so that the CUs of an XCD do not all
    // ask for the same kilobyte at the same moment.  (Compiler-issued copies: pipeline fill only; in steady state the
    // copies of stage t + 2 ride inside the MFMA stream of stage t, tchain_asm.inc.)
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
            issued += 1;
        }
    };
    // wait until stage `cons` has landed for everybody (that also frees the slot of stage cons - 1); returns its slot.
    // The caller's stream then copies stage cons + 2 into the freed slot (dma_args) or nobody does (pipeline drain).
    auto next_stage = [&]() __attribute__((always_inline)) -> int {
        if (prof && tid == 0 && cons >= 16 && cons < 64) prof[cons] = __builtin_amdgcn_s_memtime();  // stage periods 16..63
        if (issued > cons + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TC_PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int s = cons % TC_NSLOT;
        cons += 1;
        return s;
    };
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    // stream offset / LDS destination of this wave's ten pieces of the stage the running stream copies (stage `issued`)
    auto dma_args = [&](unsigned& so, unsigned& ld) __attribute__((always_inline)) {
        so = issued * TC_STAGE + wave * (TC_PIECES * 1024);
        ld = lds0 + (issued % TC_NSLOT) * TC_STAGE + wave * (TC_PIECES * 1024);
        issued += 1;
    };
    issue();  // stage 0 -> slot 0; slots 1 / 2 are the staging area until the leading operands are in registers
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
    // ---- global I/O through the staging region ----
    // Every I/O phase recomputes its addresses from a LAUNDERED copy of the lane id: otherwise the compiler shares the
    // (loop-invariant) per-piece pointers of the first and the last phase and keeps ~60 registers alive across the whole
    // kernel, spilling inside the feed-forward loop.
    auto launder = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
    const int m0w = tile * 128 + wave * 32;  // first row of this wave
    constexpr int HI_ROW = TC_C * (int)sizeof(T), LO_ROW = TC_C * (int)sizeof(lo_t<T>);
    // 16-byte piece e = 64 k + ln of the wave's tile: row = e / CPR, chunk = e % CPR.  Global side: rows >= M clamped
    // (loads) / skipped (stores); LDS side: rows padded by 16 bytes.
    auto stage_in = [&](const void* base, int row_bytes) __attribute__((always_inline)) {
        const int ln = launder(lane);
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024;
#pragma unroll
        for (int k0 = 0; k0 < 20; k0 += 5) {
            u32x4 v[5];
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < np) {
                    const int e = 64 * k + ln, r = e / cpr, c = e - r * cpr;
                    const int64_t row = (int64_t)z * p.M + min(m0w + r, p.M - 1);
                    v[k - k0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + row * row_bytes + c * 16);
                }
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < np) {
                    const int e = 64 * k + ln, r = e / cpr, c = e - r * cpr;
                    *reinterpret_cast<u32x4*>(io + r * (row_bytes + 16) + c * 16) = v[k - k0];
                }
        }
    };
    auto stage_out = [&](void* base, int row_bytes) __attribute__((always_inline)) {
        const int ln = launder(lane);
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024;
#pragma unroll
        for (int k = 0; k < 20; ++k)
            if (k < np) {
                const int e = 64 * k + ln, r = e / cpr, c = e - r * cpr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(io + r * (row_bytes + 16) + c * 16);
                if (m0w + r < p.M)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + ((int64_t)z * p.M + m0w + r) * row_bytes + c * 16) = v;
            }
    };
    // q / k rows: token matrices like every other operand, or (qk_heads = 8, round 6) a HEAD-MAJOR image
    // [sample][8 heads][token][40], whose 64-key tiles the d = 40 attention then reads as contiguous runs (csrc/attention.hip).
    // ONE code path for both: piece c of row r goes to base + qk_base + r * qk_row + c * 16 + (c / 5) * qk_head with
    // wave-uniform scalars (token matrix: row stride 640, no head term).  A wave never straddles two samples.
    int64_t qk_base;
    int qk_row, qk_head;
    if (p.qk_heads > 0) {
        const int b = m0w / p.rows_per_b, tok0 = m0w - b * p.rows_per_b;
        qk_base = ((int64_t)z * p.M + (int64_t)b * p.rows_per_b) * HI_ROW + (int64_t)tok0 * (HI_ROW / 8);
        qk_row = HI_ROW / 8;
        qk_head = (p.rows_per_b - 1) * (HI_ROW / 8);
    } else {
        qk_base = ((int64_t)z * p.M + m0w) * HI_ROW;
        qk_row = HI_ROW;
        qk_head = 0;
    }
    auto stage_out_qk = [&](void* base) __attribute__((always_inline)) {
        const int ln = launder(lane);
        char* dst = reinterpret_cast<char*>(base) + qk_base;
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const int e = 64 * k + ln, r = e / 40, c = e - r * 40;
            const u32x4 v = *reinterpret_cast<const u32x4*>(io + r * (HI_ROW + 16) + c * 16);
            if (m0w + r < p.M) *reinterpret_cast<u32x4*>(dst + (int64_t)r * qk_row + c * 16 + (int64_t)(c / 5) * qk_head) = v;
        }
    };
    // residual-stream tensor (hi [+ lo]) -> fp32 in the accumulator arrangement, ADDED to acc.  The staging region must
    // be free (ring slots 1 / 2 idle); wave-private, so only the wave's own LDS ordering is needed.
    auto add_stream = [&](Acc<T>& acc, const void* hi_, const void* lo_) __attribute__((always_inline)) {
        stage_in(hi_, HI_ROW);
        const int ln = launder(lane);
        const int lrow = ln & 31, lh = ln >> 5;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T* src = reinterpret_cast<const T*>(io + lrow * (HI_ROW + 16)) + 32 * t + 8 * q + 4 * lh;
                typedef T t4 __attribute__((ext_vector_type(4)));
                const t4 v = *reinterpret_cast<const t4*>(src);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.t[t][4 * q + r] += (float)v[r];
            }
        if (lo_) {
            stage_in(lo_, LO_ROW);
#pragma unroll
            for (int t = 0; t < TC_NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const lo_t<T>* src = reinterpret_cast<const lo_t<T>*>(io + lrow * (LO_ROW + 16)) + 32 * t + 8 * q + 4 * lh;
                    float l[4];
                    load_lo<4>(src, l);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc.t[t][4 * q + r] += l[r];
                }
        }
    };
    // accumulator -> global: hi (+ lo when asked)
    auto store_stream = [&](const Acc<T>& acc, void* hi_, void* lo_, auto qk_tag) __attribute__((always_inline)) {
        constexpr bool qk = decltype(qk_tag)::value;
        const int ln = launder(lane);
        const int lrow = ln & 31, lh = ln >> 5;
#pragma unroll
        for (int t = 0; t < TC_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                t4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (T)acc.t[t][4 * q + r];
                *reinterpret_cast<t4*>(reinterpret_cast<T*>(io + lrow * (HI_ROW + 16)) + 32 * t + 8 * q + 4 * lh) = v;
            }
        if (qk) stage_out_qk(hi_);
        else stage_out(hi_, HI_ROW);
        if (lo_) {
#pragma unroll
            for (int t = 0; t < TC_NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo_t<T> b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = acc.t[t][4 * q + r];
                        b[r] = lo_from_f<lo_t<T>>(y - to_f(from_f<T>(y)));
                    }
                    __builtin_memcpy(reinterpret_cast<lo_t<T>*>(io + lrow * (LO_ROW + 16)) + 32 * t + 8 * q + 4 * lh, b, sizeof(b));
                }
            stage_out(lo_, LO_ROW);
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
    // LDS byte address of this lane's A fragment for k16 step s inside ring slot 0 (+ slot * TC_STAGE at use; the
    // row-tile / sub-image offset is an instruction immediate): loop invariant, 4 VGPRs
    unsigned fa0[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
        fa0[s4] = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + l31 * 128 + (((2 * s4 + hh) ^ key) << 4);
    // one N = 320 GEMM stage (rows x 64 k): acc[t] += W[32 t .. 32 t + 31][k] * operand[k], 40 MFMAs, as ONE hand-scheduled
    // instruction stream (tchain_asm.inc): six fragment reads in flight, counted lgkmcnt, a read behind every MFMA
    auto gemm_stage = [&](Acc<T>& acc, int slot, const vec8& b0, const vec8& b1, const vec8& b2, const vec8& b3, bool dma, auto b_in_agpr) __attribute__((always_inline)) {
        vec8 f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
        const unsigned so = slot * TC_STAGE;
        const unsigned A0 = fa0[0] + so, A1 = fa0[1] + so, A2 = fa0[2] + so, A3 = fa0[3] + so;
#define TC_GEMM_OUTS                                                                                                           \
          [c0] "+a"(acc.t[0]), [c1] "+a"(acc.t[1]), [c2] "+a"(acc.t[2]), [c3] "+a"(acc.t[3]), [c4] "+a"(acc.t[4]),             \
          [c5] "+a"(acc.t[5]), [c6] "+a"(acc.t[6]), [c7] "+a"(acc.t[7]), [c8] "+a"(acc.t[8]), [c9] "+a"(acc.t[9]),             \
          [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6),      \
          [f7] "=&v"(f7), [f8] "=&v"(f8), [f9] "=&v"(f9), [f10] "=&v"(f10), [f11] "=&v"(f11)
        // B operands: the chain operand `bop` lives in AGPRs (the feed-forward streams read it there), the packed GEGLU
        // words in VGPRs -- one constraint letter per call site, or the allocator copies 80 registers per stage
#define TC_GEMM_INS(BC)                                                                                                        \
          [a0] "v"(A0), [a1] "v"(A1), [a2] "v"(A2), [a3] "v"(A3), [b0] BC(b0), [b1] BC(b1), [b2] BC(b2), [b3] BC(b3)
#define TC_GEMM_RUN(MT, BC)                                                                                                    \
        if (dma) {                                                                                                             \
            unsigned dso, dld, t_dso;                                                                                          \
            dma_args(dso, dld);                                                                                                \
            asm volatile(TC_ASM_GEMM_STAGE_DMA(MT) : TC_GEMM_OUTS, [dso] "=&s"(t_dso)                                          \
                         : TC_GEMM_INS(BC), [vo] "v"(voff), [rs] "s"(rs), [so0] "s"(dso), [ld0] "s"(dld) : "memory");          \
        } else {                                                                                                               \
            asm volatile(TC_ASM_GEMM_STAGE(MT) : TC_GEMM_OUTS : TC_GEMM_INS(BC));                                              \
        }
        if constexpr (decltype(b_in_agpr)::value) {
            if constexpr (__is_same(T, f16)) { TC_GEMM_RUN("f16", "a") } else { TC_GEMM_RUN("bf16", "a") }
        } else {
            if constexpr (__is_same(T, f16)) { TC_GEMM_RUN("f16", "v") } else { TC_GEMM_RUN("bf16", "v") }
        }
#undef TC_GEMM_RUN
#undef TC_GEMM_OUTS
#undef TC_GEMM_INS
    };
    // compiler code that touches the accumulators after an asm stage: an MFMA result may be read 12+ states after issue
    auto acc_fence = [&](Acc<T>& acc) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc.t[0]), "+a"(acc.t[1]), "+a"(acc.t[2]), "+a"(acc.t[3]), "+a"(acc.t[4]), "+a"(acc.t[5]),
                       "+a"(acc.t[6]), "+a"(acc.t[7]), "+a"(acc.t[8]), "+a"(acc.t[9]));
    };
    // one N = 320 GEMM pass over 5 stages; the first `ndma` of them copy the stage two ahead
    auto gemm320 = [&](Acc<T>& acc, const vec8 (&bop)[TC_KS], int ndma) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < TC_C / 64; ++kc) {
            const int slot = next_stage();
            gemm_stage(acc, slot, bop[4 * kc], bop[4 * kc + 1], bop[4 * kc + 2], bop[4 * kc + 3], kc < ndma, std::true_type{});
        }
        acc_fence(acc);
    };

    // =============================== leading GEMM: y = a0 W0^T + bias0 + residual ===============================
    Acc<T> acc;
    vec8 bop[TC_KS];
    stage_in(p.a0, HI_ROW);
    {
        const int ln = launder(lane);
#pragma unroll
        for (int s = 0; s < TC_KS; ++s)
            bop[s] = *reinterpret_cast<const vec8*>(io + (ln & 31) * (HI_ROW + 16) + (2 * s + (ln >> 5)) * 16);
    }
    zero(acc);
    if (MODE != UR_TCHAIN_PRE) add_stream(acc, p.res, p.res_lo);  // PRE: proj_in has no residual
    add_cvec(acc, TCC_BIAS0);
    __syncthreads();  // every wave has read its staging slice: slots 1 / 2 belong to the weight stream from here on
    issue();
    stamp(2);
    gemm320(acc, bop, MODE == UR_TCHAIN_FF ? 5 : 3);
    stamp(3);
    // a tile leaves through the staging area in the middle of the chain: the last two stages before it copied nothing
    // (slots 1 / 2 stay free), the ring is refilled afterwards
    auto store_mid = [&](const Acc<T>& a, void* hi_, void* lo_, auto qk_tag) __attribute__((always_inline)) {
        __syncthreads();  // the other waves have left the last stage (slot 1 or 2)
        store_stream(a, hi_, lo_, qk_tag);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left before LDS-DMA pieces are counted again
        __syncthreads();  // staging slices read back: refill the ring
        issue();
        issue();
    };
    if constexpr (MODE == UR_TCHAIN_PRE) store_mid(acc, p.y_out, p.y_out_lo, std::false_type{});
    if constexpr (MODE == UR_TCHAIN_Q) {
        // the updated residual stream leaves here through the staging area (no weight stage is in flight: `limit`)
        __syncthreads();  // the other waves have left stage 4 (slot 1)
        store_stream(acc, p.y_out, p.y_out_lo, std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left before LDS-DMA pieces are counted again
        __syncthreads();  // staging slices read back: refill the ring
        issue();
        issue();
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
    if constexpr (MODE == UR_TCHAIN_PRE) {
        // =============================== q, k, V^T of the self-attention from one normalised operand ===============================
        zero(acc);
        gemm320(acc, bop, 3);
        store_mid(acc, p.out, nullptr, std::true_type{});   // q  (softmax scale * log2(e) split evenly over q and k by the host)
        zero(acc);
        gemm320(acc, bop, 3);
        store_mid(acc, p.out2, nullptr, std::true_type{});  // k
        zero(acc);
        gemm320(acc, bop, 3);
        // V^T[b][channel][token]: a wave's 32 tokens are 64 contiguous bytes of every channel row.  Stage [channel][32
        // tokens] in the wave's slice, then 16 bytes per lane = 8 tokens of one channel.
        // The last weight stage (19) sits in ring slot 1 = the staging slices of waves 0 / 1: nobody may write there until
        // every wave has left that stage (without this barrier 1 launch in ~300 stored a few wrong V^T channels of one
        // wave -- tools/tchain_determinism.py).
        __syncthreads();
        {
            const int ln = launder(lane);
            const int lrow = ln & 31, lh = ln >> 5;
#pragma unroll
            for (int t = 0; t < TC_NT; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int ch = 32 * t + 8 * (v >> 2) + 4 * lh + (v & 3);
                    reinterpret_cast<T*>(io)[ch * 32 + lrow] = (T)acc.t[t][v];
                }
            const int b = m0w / p.rows_per_b, tok0 = m0w - b * p.rows_per_b;  // a wave never straddles two samples
            char* vt = reinterpret_cast<char*>(p.out3) + (((int64_t)z * (p.M / p.rows_per_b) + b) * TC_C * p.ld_vt + tok0) * (int64_t)sizeof(T);
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const int e = 64 * k + ln, ch = e >> 2, c = e & 3;
                const u32x4 val = *reinterpret_cast<const u32x4*>(io + ch * 64 + c * 16);
                if (m0w + 8 * c < p.M) *reinterpret_cast<u32x4*>(vt + (int64_t)ch * p.ld_vt * sizeof(T) + c * 16) = val;
            }
        }
    } else if constexpr (MODE == UR_TCHAIN_Q) {
        // =============================== q = LN(y) Wq^T (scale folded into Wq by the host) ===============================
        zero(acc);
        gemm320(acc, bop, 3);
        stamp(6);
        store_stream(acc, p.out, nullptr, std::true_type{});
        stamp(7);
    } else {
        // =============================== GEGLU feed-forward: acc = y + b2 + sum_j h_j W2_j^T ===============================
        add_cvec(acc, TCC_B2);
        // Software-pipelined over 64 hidden units j (weight-stream order A0(0) A1(0) | A0(j) A1(j) B(j-1) ... | B(19),
        // tchain.py):   A0(j) || GEGLU of (j-1, upper 32)  ->  A1(j) || GEGLU of (j, lower 32)  ->  B(j-1).
        // A* = 40 MFMAs of the input projection (value + gate rows of 32 hidden units over all 320 k) into one of two
        // accumulator pairs, with the GEGLU VALU program of the OTHER pair interleaved instruction by instruction
        // (tchain_asm.inc); B = the 40 MFMAs of the output projection over the 64 packed h columns.
        // pin the chain operand into AGPRs for the whole loop (its producers wrote VGPRs; without this the allocator keeps it
        // there and copies all 80 registers in front of every stream)
#pragma unroll
        for (int k = 0; k < TC_KS; ++k) asm volatile("" : "+a"(bop[k]));
        const float KS[3] = TC_GELU_KS;
        const float KV[5] = TC_GELU_KV;
        f32x16 hvA, hgA, hvB, hgB;
        unsigned hbw[2][2][8];  // [j & 1][lower / upper 32 hidden][packed words]: B operands of B(j)
        auto bias4 = [&](int hid, f32x4 (&BV)[4], f32x4 (&BG)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                BV[q] = *reinterpret_cast<const f32x4*>(cst + TCC_B1V + hid + 8 * q + 4 * hh);
                BG[q] = *reinterpret_cast<const f32x4*>(cst + TCC_B1G + hid + 8 * q + 4 * hh);
            }
        };
#define TC_TEMPS float t_g0, t_g1, t_g2, t_g3, t_e0, t_e1, t_e2, t_e3, t_t0, t_t1, t_t2, t_t3, t_p0, t_p1, t_p2, t_p3
        auto ffa = [&](f32x16& HV, f32x16& HG) __attribute__((always_inline)) {
            const unsigned so = next_stage() * TC_STAGE;
            const unsigned A[4] = {fa0[0] + so, fa0[1] + so, fa0[2] + so, fa0[3] + so};
            vec8 f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
            unsigned dso, dld, t_dso;
            dma_args(dso, dld);
            if constexpr (__is_same(T, f16)) asm volatile(TC_ASM_FFA("f16") TC_OPS_FFA(HV, HG, A, bop, voff, rs, dso, dld));
            else asm volatile(TC_ASM_FFA("bf16") TC_OPS_FFA(HV, HG, A, bop, voff, rs, dso, dld));
        };
        auto ffag = [&](f32x16& HV, f32x16& HG, const f32x16& PV, const f32x16& PG, unsigned (&O)[8], int hid) __attribute__((always_inline)) {
            f32x4 BV[4], BG[4];
            bias4(hid, BV, BG);
            const unsigned so = next_stage() * TC_STAGE;
            const unsigned A[4] = {fa0[0] + so, fa0[1] + so, fa0[2] + so, fa0[3] + so};
            vec8 f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
            TC_TEMPS;
            unsigned dso, dld, t_dso;
            dma_args(dso, dld);
            if constexpr (__is_same(T, f16)) asm volatile(TC_ASM_FFAG("f16") TC_OPS_FFAG(HV, HG, PV, PG, O, A, bop, BV, BG, KS, KV, voff, rs, dso, dld));
            else asm volatile(TC_ASM_FFAG("bf16") TC_OPS_FFAG(HV, HG, PV, PG, O, A, bop, BV, BG, KS, KV, voff, rs, dso, dld));
        };
        auto g_only = [&](const f32x16& PV, const f32x16& PG, unsigned (&O)[8], int hid) __attribute__((always_inline)) {
            f32x4 BV[4], BG[4];
            bias4(hid, BV, BG);
            TC_TEMPS;
            if constexpr (__is_same(T, f16)) asm volatile(TC_ASM_G("f16") TC_OPS_G(PV, PG, O, BV, BG, KS, KV));
            else asm volatile(TC_ASM_G("bf16") TC_OPS_G(PV, PG, O, BV, BG, KS, KV));
        };
#undef TC_TEMPS
        auto bstage = [&](const unsigned (&W)[2][8]) __attribute__((always_inline)) {
            const int slot = next_stage();
            vec8 b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 u = {W[k >> 1][4 * (k & 1)], W[k >> 1][4 * (k & 1) + 1], W[k >> 1][4 * (k & 1) + 2], W[k >> 1][4 * (k & 1) + 3]};
                b[k] = __builtin_bit_cast(vec8, u);
            }
            gemm_stage(acc, slot, b[0], b[1], b[2], b[3], true, std::false_type{});
        };
        ffa(hvA, hgA);                                    // A0(0)
        ffag(hvB, hgB, hvA, hgA, hbw[0][0], 0);           // A1(0) || G(0, lower)
        for (int j = 1; j < TC_FF / 64; j += 2) {
            ffag(hvA, hgA, hvB, hgB, hbw[0][1], 64 * (j - 1) + 32);  // A0(j) || G(j-1, upper)      (j odd: j-1 even -> set 0)
            ffag(hvB, hgB, hvA, hgA, hbw[1][0], 64 * j);             // A1(j) || G(j, lower)
            bstage(hbw[0]);                                          // B(j-1)
            if (j + 1 < TC_FF / 64) {
                ffag(hvA, hgA, hvB, hgB, hbw[1][1], 64 * j + 32);    // A0(j+1) || G(j, upper)
                ffag(hvB, hgB, hvA, hgA, hbw[0][0], 64 * (j + 1));   // A1(j+1) || G(j+1, lower)
                bstage(hbw[1]);                                      // B(j)
            }
        }
        g_only(hvB, hgB, hbw[1][1], TC_FF - 32);          // G(19, upper)
        bstage(hbw[1]);                                   // B(19)
        acc_fence(acc);
        stamp(6);
        // =============================== out = y3 Wpo^T + bpo + block input ===============================
        to_operand(acc, bop);
        zero(acc);
        gemm320(acc, bop, 3);
        stamp(7);
        add_cvec(acc, TCC_BPO);
        add_stream(acc, p.blk, p.blk_lo);
        stamp(8);
        store_stream(acc, p.out, p.out_lo, std::false_type{});
        stamp(9);
    }
}

template <typename T, int MODE>
static int launch_tchain(const ur_tchain_desc& d, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    const int nconst = MODE == UR_TCHAIN_FF ? TCC_FF_END : TCC_Q_END;
    const int lds = TC_LDS;  // ring + staging area + constants: the whole CU
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
    if (!d.a0 || !d.out || !d.wstream || !d.consts || d.M <= 0) return UR_E_BADARG;
    if (d.mode != UR_TCHAIN_PRE && !d.res) return UR_E_BADARG;
    if (d.zbatch < 1) d.zbatch = 1;
    if (d.mode == UR_TCHAIN_Q) {
        if (!d.y_out) return UR_E_BADARG;
        if (d.z_consts < TCC_Q_END || d.z_wstream != 10 * (int64_t)TC_STAGE) return UR_E_BADARG;  // exactly this chain's stream
    } else if (d.mode == UR_TCHAIN_PRE) {
        if (!d.y_out || !d.out2 || !d.out3) return UR_E_BADARG;
        if (d.rows_per_b <= 0 || (d.rows_per_b % 32) || (d.M % d.rows_per_b) || d.ld_vt < d.rows_per_b || (d.ld_vt % 8)) return UR_E_BADARG;
        if (d.z_consts < TCC_Q_END || d.z_wstream != 20 * (int64_t)TC_STAGE) return UR_E_BADARG;
    } else if (d.mode == UR_TCHAIN_FF) {
        if (!d.blk) return UR_E_BADARG;
        // a feed-forward of another hidden size would pack a stream of another length: refuse it instead of streaming
        // the wrong stage images
        if (d.z_consts < TCC_FF_END || d.z_wstream != (10 + 3 * (TC_FF / 64)) * (int64_t)TC_STAGE) return UR_E_BADARG;
    } else {
        return UR_E_BADARG;
    }
    if (d.channels != TC_C) return UR_E_UNSUPPORTED;
    if (d.qk_heads) {  // head-major q / k: 8 heads of 40, whole samples of a multiple of 32 tokens (a wave never straddles two)
        if (d.mode == UR_TCHAIN_FF) return UR_E_BADARG;
        if (d.qk_heads != 8) return UR_E_UNSUPPORTED;
        if (d.rows_per_b <= 0 || (d.rows_per_b % 32) || (d.M % d.rows_per_b)) return UR_E_BADARG;
    }
    if ((reinterpret_cast<uintptr_t>(d.wstream) | (uintptr_t)d.z_wstream) & 15) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.dtype == UR_DT_F16)
        return d.mode == UR_TCHAIN_Q ? launch_tchain<f16, UR_TCHAIN_Q>(d, s)
               : (d.mode == UR_TCHAIN_PRE ? launch_tchain<f16, UR_TCHAIN_PRE>(d, s) : launch_tchain<f16, UR_TCHAIN_FF>(d, s));
    if (d.dtype == UR_DT_BF16)
        return d.mode == UR_TCHAIN_Q ? launch_tchain<bf16, UR_TCHAIN_Q>(d, s)
               : (d.mode == UR_TCHAIN_PRE ? launch_tchain<bf16, UR_TCHAIN_PRE>(d, s) : launch_tchain<bf16, UR_TCHAIN_FF>(d, s));
    return UR_E_BADARG;
}

extern "C" int ur_sizeof_tchain_desc(void) { return (int)sizeof(ur_tchain_desc); }
extern "C" int64_t ur_tchain_stream_bytes(int mode) {
    return (int64_t)ur::TC_STAGE * (mode == UR_TCHAIN_Q ? 10 : (mode == UR_TCHAIN_PRE ? 20 : 10 + 3 * (ur::TC_FF / 64)));
}
extern "C" int ur_tchain_const_floats(int mode) { return mode == UR_TCHAIN_FF ? ur::TCC_FF_END : ur::TCC_Q_END; }
