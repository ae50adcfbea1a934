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
    auto issue = [&]() __attribute__((always_inline)) {
        if (issued < NSTAGES) {
            const int slot = issued % TC_NSLOT;
            const int sbase = issued * TC_STAGE + wave * (TC_PIECES * 1024);
            char* dst = smem + slot * TC_STAGE + wave * (TC_PIECES * 1024);
#pragma unroll
            for (int i = 0; i < TC_PIECES; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, voff,
                                                         sbase + i * 1024, 0, 0);
        }
        issued += 1;
    };
    // wait until stage `cons` has landed for everybody, free the slot of stage cons - 1, refill it with stage cons + 2
    auto next_stage = [&]() __attribute__((always_inline)) -> const char* {
        if (cons + 1 < NSTAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TC_PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const char* s = smem + (cons % TC_NSLOT) * TC_STAGE;
        cons += 1;
        return s;
    };
    issue();
    issue();

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
    // one N = 320 GEMM pass over 5 stages: acc[t] += W[32 t .. 32 t + 31][k] * operand[k]
    auto gemm320 = [&](Acc<T>& acc, const vec8 (&bop)[TC_KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < TC_C / 64; ++kc) {
            const char* st = next_stage();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < TC_NT; ++t) acc.t[t] = mfma32(afrag(st + t * 4096, s), bop[4 * kc + s], acc.t[t]);
        }
    };

    // =============================== leading GEMM: y = a0 W0^T + bias0 + residual ===============================
    Acc<T> acc;
    vec8 bop[TC_KS];
#pragma unroll
    for (int s = 0; s < TC_KS; ++s) bop[s] = *reinterpret_cast<const vec8*>(a0 + 16 * s + 8 * hh);
    zero(acc);
    add_stream(acc, p.res, p.res_lo);
    add_cvec(acc, TCC_BIAS0);
    gemm320(acc, bop);
    if constexpr (MODE == UR_TCHAIN_Q) {
        // the updated residual stream leaves here; the LDS-DMA queue is drained first so that the stores are the
        // only vector-memory operations counted between the two GEMM passes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_stream(acc, p.y_out, p.y_out_lo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

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

    if constexpr (MODE == UR_TCHAIN_Q) {
        // =============================== q = LN(y) Wq^T (scale folded into Wq by the host) ===============================
        zero(acc);
        gemm320(acc, bop);
        store_stream(acc, p.out, nullptr);
    } else {
        // =============================== GEGLU feed-forward: acc = y + b2 + sum_j h_j W2_j^T ===============================
        add_cvec(acc, TCC_B2);
        for (int j = 0; j < TC_FF / 64; ++j) {
            vec8 hb[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // stage image: five [64 rows][64 k] sub-images; rows 0..31 = value rows, 32..63 = gate rows of 32 hidden units
                const char* st = next_stage();
                f32x16 hv[2], hg[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { hv[u][v] = 0.f; hg[u][v] = 0.f; }
#pragma unroll
                for (int c = 0; c < TC_C / 64; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        hv[s & 1] = mfma32(afrag(st + c * 8192, s), bop[4 * c + s], hv[s & 1]);
                        hg[s & 1] = mfma32(afrag(st + c * 8192 + 4096, s), bop[4 * c + s], hg[s & 1]);
                    }
                const int hid = 64 * j + 32 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(cst + TCC_B1V + hid + 8 * q + 4 * hh);
                    const float4 bg = *reinterpret_cast<const float4*>(cst + TCC_B1G + hid + 8 * q + 4 * hh);
                    const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgg[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float val = hv[0][4 * q + r] + hv[1][4 * q + r] + bvv[r];
                        const float gate = hg[0][4 * q + r] + hg[1][4 * q + r] + bgg[r];
                        hb[2 * half + (q >> 1)][4 * (q & 1) + r] = (T)(val * gelu_erf_f(gate));
                    }
                }
            }
            const char* st = next_stage();  // W2[:, 64 j .. 64 j + 63]
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < TC_NT; ++t) acc.t[t] = mfma32(afrag(st + t * 4096, s), hb[s], acc.t[t]);
        }
        // =============================== out = y3 Wpo^T + bpo + block input ===============================
        to_operand(acc, bop);
        zero(acc);
        gemm320(acc, bop);
        add_cvec(acc, TCC_BPO);
        add_stream(acc, p.blk, p.blk_lo);
        store_stream(acc, p.out, p.out_lo);
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
