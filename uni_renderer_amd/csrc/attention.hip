// Flash-style attention on MFMA for gfx950: O = softmax(Q K^T * scale) V, never materialising the
// score matrix (online softmax in fp32).  Used for the self-attention (Tq = Tk = H*W) and the 77-key
// cross-attention of every BasicTransformerBlock.
//
// Workgroup = 4 waves x 32 queries = 128 queries of one (batch, head); keys are visited in tiles of
// 64, K tile [64][DK] and V^T tile [DV][64] copied global -> LDS by LDS-DMA, double buffered.
//
// Everything is computed in the "transposed" MFMA orientation so that no cross-lane data movement is
// needed between the two GEMMs:
//   S^T[key][query] = K . Q^T      (A = K rows, B = Q rows)   -> lane (query j, q') holds keys
//   O^T[d][query]   = V^T . P^T    (A = V^T rows, B = P rows) -> lane (query j, q') holds d = 4q'+r
// The key rows fed to MFMA row i = 4q'+r are permuted (key = 32*kb + 8q' + 4*sub + r) so that after
// S^T a lane holds 8 CONSECUTIVE keys 8q'..8q'+7 of each 32-key block -- exactly the B-operand layout
// of the P.V MFMA, and V^T fragments become single 16-byte LDS reads.  V arrives pre-transposed
// ([B][H*d][Tk_pad]) from the projection GEMM (ur_igemm with swapped operands), so there is no
// transposing store anywhere.  The running max/sum live per lane (the 4 lanes sharing a query hold
// identical maxima, partial sums are combined once at the end).
#include <type_traits>

#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

template <int CPR>
__device__ __forceinline__ int k_swz(int row) {
    // XOR key applied to the 16-byte chunk index of a K-tile row.  Chosen with tools/lds_bank_check.py
    // for the fragment read pattern rows {0-3, 8-11, 16-19, 24-27} + const: conflict-free for 64/128/192/
    // 320-byte rows, 2-way for 256-byte rows.  The key must stay inside an aligned group of CPR chunks.
    if (CPR % 8 == 0) return (row & 3) | (((row >> 3) & 1) << 2);
    return (row ^ (row >> 2)) & 3;
}

// max over the 4 lanes {l15 + 16*q'} that share a query, without LDS permutes: gfx950 swaps whole 16-lane rows
// (v_permlane16_swap: odd rows of a <-> even rows of b) and 32-lane halves (v_permlane32_swap).
__device__ __forceinline__ float xor16_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <typename T, int D>
__global__ void __launch_bounds__(256) attention_kernel(const ur_attn_desc p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int DK = (D + 31) / 32 * 32;  // contraction length of Q.K^T, zero padded
    constexpr int DV = (D + 15) / 16 * 16;  // output rows of O^T
    constexpr int KSTEPS = DK / 32;
    constexpr int DFR = DV / 16;
    constexpr int CPR = DK / 8;          // 16-byte chunks per K-tile row
    constexpr int KT_BYTES = 64 * DK * 2;
    constexpr int VT_BYTES = DV * 128;
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int KI = CPR / 4;          // K-tile LDS-DMA instructions per wave
    constexpr int VROWS8 = DV / 8;       // V^T tile 8-row groups (one instruction each)
    constexpr int VI = (VROWS8 + 3) / 4;
    // Head dims with zero-padded V^T rows (d = 40 -> 48 rows) get the softmax denominator from the MFMA for free:
    // padding row D of the LDS tile is all ones, so O^T row D accumulates sum_k P[k] (of the SAME rounded P that
    // multiplies V) and follows every rescale of the accumulator.  The padding rows are written once, not loaded.
    constexpr bool MFMA_ROWSUM = (DV > D) && (D % 8 == 0);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, qq = lane >> 4;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);  // a (b,h) stays on one XCD
    const int bh = lid / gridDim.x, qt = lid - bh * gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 128 + wave * 32;

    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Tq * p.ldq + p.q_off + h * D;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk * p.ldk + p.k_off + h * D;
    const T* VT = reinterpret_cast<const T*>(p.vt) + (int64_t)b * p.vt_bstride + (int64_t)h * D * p.ldvt;
    const T* zp = reinterpret_cast<const T*>(p.zero_page);

    // ---- Q fragments (B operand of S^T): lane (query l15 of fragment qf, qq) holds 8 d-values per k-step
    vec8 qf[2][KSTEPS];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int qrow = q0 + f * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int dc = ks * 32 + qq * 8;
            const T* src = (qrow < p.Tq && dc < D) ? Q + (int64_t)qrow * p.ldq + dc : zp;
            qf[f][ks] = *reinterpret_cast<const vec8*>(src);
        }
    }

    // ---- loader state, hoisted out of the key loop: every lane owns fixed (row, chunk) slots of the K and V^T
    // tiles and walks them with one pointer each (+64 keys per tile; zero-page lanes stand still).  Only a ragged
    // LAST tile re-tests its key rows against Tk.
    const char* zpc = reinterpret_cast<const char*>(zp);
    const char* kptr[KI];
    int kinc[KI], krow[KI];
    const int kstep = 64 * (int)p.ldk * (int)sizeof(T);
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int g = (wave * KI + it) * 64 + lane;  // linear 16-byte chunk of the K tile
        const int row = g / CPR, c = g - row * CPR;
        const int cl = c ^ k_swz<CPR>(row);
        const bool real = cl * 8 < D;  // padded head-dim chunks always read zeros
        krow[it] = real ? row : (1 << 30);
        kptr[it] = real ? reinterpret_cast<const char*>(K + (int64_t)row * p.ldk + cl * 8) : zpc;
        kinc[it] = real ? kstep : 0;
    }
    const char* vptr[VI];
    int vinc[VI];
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int row = (wave * VI + it) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ (lane >> 3);
        const bool real = row < D;
        vptr[it] = real ? reinterpret_cast<const char*>(VT + (int64_t)row * p.ldvt + cl * 8) : zpc;
        vinc[it] = real ? 128 : 0;
    }

    auto stage = [&](int buf, int kt, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        char* ks_ = smem + buf * STAGE;
        char* vs_ = ks_ + KT_BYTES;
#pragma unroll
        for (int it = 0; it < KI; ++it) {
            const char* src = kptr[it];
            if (tail) src = (kt * 64 + krow[it] < p.Tk) ? src : zpc;
            glds16(src, ks_ + (wave * KI + it) * 1024);
            kptr[it] += kinc[it];
        }
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int ii = wave * VI + it;  // 8-row group of the V^T tile
            if (ii < VROWS8 && !(MFMA_ROWSUM && ii * 8 >= D)) glds16(vptr[it], vs_ + ii * 1024);
            vptr[it] += vinc[it];
        }
    };

    f32x4 acc_o[DFR][2];
#pragma unroll
    for (int i = 0; i < DFR; ++i) {
        acc_o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr float NEG_BIG = -1.0e30f;
    constexpr float RESCALE_THR = 8.0f;
    const float cs = (p.scale > 0.f) ? p.scale * 1.44269504088896341f : 1.0f;  // softmax in base 2; scale <= 0: Q.K^T is already in log2 units
    float m_run[2] = {NEG_BIG, NEG_BIG};
    float mc_run[2] = {-NEG_BIG * cs, -NEG_BIG * cs};  // -m_run * cs, the addend of p = exp2(s*cs - m*cs)
    float l_run[2] = {0.f, 0.f};

    const int nkt = (p.Tk + 63) / 64;
    if (MFMA_ROWSUM) {
        constexpr int PER = (DV - D) * 64;  // padding elements per stage buffer
        for (int i = tid; i < 2 * PER; i += 256) {
            const int buf = i / PER, rem = i - buf * PER;
            const int row = D + (rem >> 6), col = rem & 63;
            reinterpret_cast<T*>(smem + buf * STAGE + KT_BYTES + row * 128)[col] = (T)(row == D ? 1.0f : 0.0f);
        }
    }
    const int nfull = p.Tk >> 6;
    if (nfull > 0) stage(0, 0, std::false_type{});
    else stage(0, 0, std::true_type{});
    __syncthreads();
    // The tile body is instantiated twice: full tiles carry no key masking at all; only a ragged last tile
    // (Tk % 64 != 0, e.g. the 77 prompt tokens) pays for the per-score compare/select.
    auto tile = [&](const int kt, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        if (kt + 1 < nfull) stage((kt + 1) & 1, kt + 1, std::false_type{});
        else if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1, std::true_type{});
        const char* ks_ = smem + (kt & 1) * STAGE;
        const char* vs_ = ks_ + KT_BYTES;

        // ---- S^T = K . Q^T : acc_s[kb][sub][f], MFMA row i = 4q'+r  <->  key 32*kb + 8q' + 4*sub + r
        f32x4 acc_s[2][2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                acc_s[kb][sub][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc_s[kb][sub][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int krow = kb * 32 + (l15 >> 2) * 8 + sub * 4 + (l15 & 3);
                    const int c = (ks * 4 + qq) ^ k_swz<CPR>(krow);
                    const vec8 kf = *reinterpret_cast<const vec8*>(ks_ + (krow * CPR + c) * 16);
                    acc_s[kb][sub][0] = mfma16(kf, qf[0][ks], acc_s[kb][sub][0]);
                    acc_s[kb][sub][1] = mfma16(kf, qf[1][ks], acc_s[kb][sub][1]);
                }
        }

        // ---- online softmax (per query fragment f; this lane's query is l15 of that fragment).
        // Raw scores are kept unscaled: max in the raw domain (scale > 0), p = exp2(s*cs - m*cs) is ONE fma
        // feeding v_exp_f32.  Masked keys use a large finite negative (exp2 -> 0) so no inf/NaN arithmetic.
        vec8 pf[2][2];  // [kb][f] : P^T fragment = 8 consecutive keys of this lane's query
        float mx[2];
        if (tail) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + kb * 32 + qq * 8 + sub * 4 + r;
                        if (key >= p.Tk) {
                            acc_s[kb][sub][0][r] = NEG_BIG;
                            acc_s[kb][sub][1][r] = NEG_BIG;
                        }
                    }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // two independent v_max3 chains over the lane's 16 scores
            auto e = [&](int kb, int i) { return acc_s[kb][i >> 2][f][i & 3]; };
            float ma = fmaxf(e(0, 0), e(0, 7)), mb = fmaxf(e(1, 0), e(1, 7));
#pragma unroll
            for (int i = 1; i < 7; i += 2) {
                ma = fmaxf(fmaxf(ma, e(0, i)), e(0, i + 1));
                mb = fmaxf(fmaxf(mb, e(1, i)), e(1, i + 1));
            }
            float m = fmaxf(ma, mb);
            // the 4 lanes (l15, q' = 0..3) sharing a query: gfx950 row / half swaps instead of LDS permutes
            m = xor16_max(m);
            mx[f] = xor32_max(m);
        }
        // Lazy rescale: the reference maximum of a query only moves when some score of this tile exceeds it by
        // more than 2^RESCALE_THR (then p would outgrow the fp16/bf16 range); otherwise P keeps the old reference
        // and the accumulator / row-sum rescale is skipped.  The test is wave-uniform, so it is one scalar branch.
        const bool grow = (fmaf(mx[0], cs, mc_run[0]) > RESCALE_THR) || (fmaf(mx[1], cs, mc_run[1]) > RESCALE_THR);
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float m_new = fmaxf(m_run[f], mx[f]);
                const float alpha = __builtin_amdgcn_exp2f((m_run[f] - m_new) * cs);
                m_run[f] = m_new;
                mc_run[f] = -m_new * cs;
                l_run[f] *= alpha;
#pragma unroll
                for (int i = 0; i < DFR; ++i) acc_o[i][f] *= alpha;
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float mc = mc_run[f];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_exp2f(fmaf(acc_s[kb][i >> 2][f][i & 3], cs, mc));
                if (!MFMA_ROWSUM) l_run[f] += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
                vec8 pv;
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = (T)t[i];
                pf[kb][f] = pv;
            }
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int df = 0; df < DFR; ++df) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int vrow = df * 16 + l15;
                const int c = (kb * 4 + qq) ^ (vrow & 7);
                const vec8 vf = *reinterpret_cast<const vec8*>(vs_ + vrow * 128 + c * 16);
                acc_o[df][0] = mfma16(vf, pf[kb][0], acc_o[df][0]);
                acc_o[df][1] = mfma16(vf, pf[kb][1], acc_o[df][1]);
            }
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nfull, std::true_type{});

    // ---- finalize: combine the partial row sums of the 4 lanes sharing a query, normalise, store
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        float l;
        if (MFMA_ROWSUM) {
            l = __shfl(acc_o[DFR - 1][f][(D % 16) % 4], ((D % 16) / 4) * 16 + l15, 64);
        } else {
            l = l_run[f];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qrow = q0 + f * 16 + l15;
        if (p.lse && qq == 0 && qrow < p.Tq)  // log2-unit log-sum-exp of the row: reference + log2(sum of exp2(s - reference))
            p.lse[((int64_t)b * p.H + h) * p.Tq + qrow] = __builtin_amdgcn_logf(l) - mc_run[f];
        if (qrow < p.Tq) {
            T* orow = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Tq + qrow) * p.ldo + h * D;
#pragma unroll
            for (int df = 0; df < DFR; ++df) {
                const int dd = df * 16 + qq * 4;
                if (dd + 4 <= D) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    vec4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (T)(acc_o[df][f][r] * inv);
                    *reinterpret_cast<vec4*>(orow + dd) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Head dims <= 64 (SD-1.x level 0: d = 40, the 4096-token self-attention that dominates the step) run on the
// 32x32x16 MFMA: the same matrix work in half the instructions (7 per 32 keys instead of 14), i.e. half the MFMA issue
// slots competing with the softmax VALU stream; measured 280 -> 272 us at B.H = 64, T = 4096.  (An
// earlier comment blamed a "27-cycle" 16x16x32: that was a micro-benchmark artefact -- both shapes issue at their nominal
// 16 / 32 cycles, profiles/r03_mfma_rate.txt.)
//
// Same transposed formulation, one 32-query block per wave:
//   S^T[key][query] = K . Q^T : A = 32 key rows, B = Q rows.  Lane (query j = l&31, h = l>>5) receives, in register
//       v = 4g + r, MFMA row 8g + 4h + r.  The A fragment of MFMA row i reads LDS key row
//       16*(g>>1) + 8h + 4*(g&1) + r, so that the lane's registers 8*s .. 8*s+7 are the 8 CONSECUTIVE keys
//       16s + 8h + (0..7) -- the B operand of P.V k-step s without any cross-lane traffic.
//   O^T[d][query] = V^T . P^T : A = 32 rows of the V^T tile (d = 40 -> 2 blocks, rows 40.. are padding; row 40 is
//       all ones so that O^T row 40 is the softmax denominator), B = P.
// K tile rows are 128 B (64 halfs, head dim zero padded), V^T rows 128 B (64 keys); both XOR-swizzled by
// ((row >> 1) & 7) on the LDS-DMA source side (conflict-free for 32-consecutive-row b128 reads, lds_bank_check.py).
template <typename T, int D, bool SLOT>
__global__ void __launch_bounds__(256) attention32_kernel(const ur_attn_desc p) {
    typedef typename Vec8<T>::type vec8;
    static_assert(D <= 64 && D % 8 == 0, "32x32 path: head dim <= 64");
    constexpr int KSTEPS = (D + 15) / 16;  // k-steps of Q.K^T
    constexpr int DB = (D + 31) / 32;      // 32-row blocks of O^T
    constexpr int DV = DB * 32;
    constexpr int KT_BYTES = 64 * 128;
    constexpr int VT_BYTES = DV * 128;
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int VROWS8 = D / 8;  // 8-row groups of V^T that hold data
    constexpr int VI = (VROWS8 + 3) / 4;
    constexpr bool MFMA_ROWSUM = DV > D;
    // SLOT (scores pre-scaled to log2 units by the projections, and a zero-padded k-slot at d = D): Q carries
    // -m_ref of its query in that slot and K a constant 1, so S^T comes out of the MFMA already shifted by the
    // softmax reference and P = exp2(S) needs NO per-score multiply-add.  Any reference within 2^RESCALE_THR of the
    // true maximum is valid, so m_ref is kept at storage-type precision.
    static_assert(!SLOT || (KSTEPS * 16 > D), "the reference slot needs a padded k column");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int bh = lid / gridDim.x, qt = lid - bh * gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 128 + wave * 32;

    // head-major images (q_hstride / k_hstride > 0: [sample][head][token][D], round 6): a head's rows are D elements apart and
    // contiguous, so a 64-key tile is ONE 64*D*2-byte run instead of 64 slices of 2*D bytes inside (H*D*2)-byte token rows
    const int64_t qrs = p.q_hstride > 0 ? D : p.ldq, krs = p.k_hstride > 0 ? D : p.ldk;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Tq * p.ldq + p.q_off + (p.q_hstride > 0 ? h * p.q_hstride : (int64_t)h * D);
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk * p.ldk + p.k_off + (p.k_hstride > 0 ? h * p.k_hstride : (int64_t)h * D);
    const T* VT = reinterpret_cast<const T*>(p.vt) + (int64_t)b * p.vt_bstride + (int64_t)h * D * p.ldvt;
    const char* zpc = reinterpret_cast<const char*>(p.zero_page);

    // Q fragments (B operand): lane (query l31, half hh) holds d = 16*ks + 8*hh .. +7
    vec8 qf[KSTEPS];
    {
        const int qrow = q0 + l31;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int dc = ks * 16 + hh * 8;
            const void* src = (qrow < p.Tq && dc < D) ? (const void*)(Q + (int64_t)qrow * qrs + dc) : (const void*)zpc;
            qf[ks] = *reinterpret_cast<const vec8*>(src);
        }
    }

    // loader: lane owns (row = 8*piece + lane/8, LDS chunk lane%8) of every 1 KiB piece; pointers walk 64 keys per tile
    const char* kptr[2];
    int kinc[2], krow[2];
    const int kstep = 64 * (int)krs * (int)sizeof(T);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (wave * 2 + it) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((row >> 1) & 7);
        const bool real = cl * 8 < D;
        krow[it] = real ? row : (1 << 30);
        kptr[it] = real ? reinterpret_cast<const char*>(K + (int64_t)row * krs + cl * 8) : zpc;
        kinc[it] = real ? kstep : 0;
    }
    const char* vptr[VI];
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int row = (wave * VI + it) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((row >> 1) & 7);
        vptr[it] = reinterpret_cast<const char*>(VT + (int64_t)min(row, D - 1) * p.ldvt + cl * 8);
    }

    auto stage = [&](int buf, int kt, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        char* ks_ = smem + buf * STAGE;
        char* vs_ = ks_ + KT_BYTES;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const char* src = kptr[it];
            if (tail) src = (kt * 64 + krow[it] < p.Tk) ? src : zpc;
            if (SLOT) {  // padded chunks keep their one-time image (lane-masked DMA)
                if (kinc[it] != 0) glds16(src, ks_ + (wave * 2 + it) * 1024);
            } else {
                glds16(src, ks_ + (wave * 2 + it) * 1024);
            }
            kptr[it] += kinc[it];
        }
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int ii = wave * VI + it;
            if (ii < VROWS8) glds16(vptr[it], vs_ + ii * 1024);
            vptr[it] += 128;
        }
    };

    f32x16 acc_o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc_o[i][v] = 0.f;
    constexpr float NEG_BIG = -1.0e30f;
    constexpr float RESCALE_THR = 8.0f;
    const float cs = (p.scale > 0.f) ? p.scale * 1.44269504088896341f : 1.0f;  // scale <= 0: already log2 units
    float m_run = NEG_BIG, mc_run = -NEG_BIG * cs, l_run = 0.f;
    float m_ref = 0.f;  // SLOT: reference currently stored (negated) in the Q slot

    const int nkt = (p.Tk + 63) / 64;
    const int nfull = p.Tk >> 6;
    if (DV > D) {  // padding rows of the V^T tiles: written once (row D = ones when it carries the row sums)
        constexpr int PER = (DV - D) * 64;
        for (int i = tid; i < 2 * PER; i += 256) {
            const int buf = i / PER, rem = i - buf * PER;
            const int row = D + (rem >> 6), col = rem & 63;
            reinterpret_cast<T*>(smem + buf * STAGE + KT_BYTES + row * 128)[col] = (T)((MFMA_ROWSUM && row == D) ? 1.0f : 0.0f);
        }
    }
    if (SLOT) {  // K tile chunk d = D..D+7 of every row: {1, 0, ...}; never touched by the DMA afterwards
        for (int i = tid; i < 2 * 64; i += 256) {
            const int buf = i >> 6, row = i & 63;
            vec8 one;
#pragma unroll
            for (int j = 0; j < 8; ++j) one[j] = (T)(j == 0 ? 1.0f : 0.0f);
            *reinterpret_cast<vec8*>(smem + buf * STAGE + row * 128 + (((D / 8) ^ ((row >> 1) & 7)) << 4)) = one;
        }
    }
    if (nfull > 0) stage(0, 0, std::false_type{});
    else stage(0, 0, std::true_type{});
    __syncthreads();

    // LDS byte offsets of this lane's fragments (without the k-step chunk)
    const int krow_l = 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);  // key row of MFMA row l31
    const int kswz = (krow_l >> 1) & 7, vswz = (l31 >> 1) & 7;

    auto tile = [&](const int kt, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        if (kt + 1 < nfull) stage((kt + 1) & 1, kt + 1, std::false_type{});
        else if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1, std::true_type{});
        const char* ks_ = smem + (kt & 1) * STAGE;
        const char* vs_ = ks_ + KT_BYTES;

        f32x16 acc_s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc_s[kb][v] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const vec8 kf = *reinterpret_cast<const vec8*>(ks_ + (kb * 32 + krow_l) * 128 + (((ks * 2 + hh) ^ kswz) << 4));
                acc_s[kb] = mfma32(kf, qf[ks], acc_s[kb]);
            }
        if (tail) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int key = kt * 64 + kb * 32 + (v >> 3) * 16 + hh * 8 + (v & 7);
                    if (key >= p.Tk) acc_s[kb][v] = NEG_BIG;
                }
        }
        float ma = fmaxf(acc_s[0][0], acc_s[0][15]), mb = fmaxf(acc_s[1][0], acc_s[1][15]);
#pragma unroll
        for (int v = 1; v < 15; v += 2) {
            ma = fmaxf(fmaxf(ma, acc_s[0][v]), acc_s[0][v + 1]);
            mb = fmaxf(fmaxf(mb, acc_s[1][v]), acc_s[1][v + 1]);
        }
        const float mx = xor32_max(fmaxf(ma, mb));  // the two lanes (l31, hh = 0/1) of a query
        if (SLOT) {
            // acc_s is already relative to m_ref.  Move the reference only when a score outgrows it by 2^THR (or on
            // the first tile, which sets it); the scores of THIS tile are then shifted by hand (rare path).
            if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0) {
                const float up = (kt == 0) ? mx : fmaxf(mx, 0.f);
                const float ref_new = (float)(T)(m_ref + up);
                const float delta = ref_new - m_ref;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref = ref_new;
                if (hh == 1) qf[KSTEPS - 1][0] = (T)(-ref_new);  // slot d = D lives in the upper half's fragment
#pragma unroll
                for (int i = 0; i < DB; ++i) acc_o[i] *= alpha;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) acc_s[kb] -= delta;
            }
        } else if (__builtin_amdgcn_ballot_w64(fmaf(mx, cs, mc_run) > RESCALE_THR) != 0) {  // lazy rescale, wave-uniform
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
            m_run = m_new;
            mc_run = -m_new * cs;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i) acc_o[i] *= alpha;
        }
        vec8 pf[2][2];  // [kb][s]: keys 32kb + 16s + 8hh + (0..7) of this lane's query
        auto make_p = [&](int kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    t[i] = SLOT ? __builtin_amdgcn_exp2f(acc_s[kb][s2 * 8 + i])
                                : __builtin_amdgcn_exp2f(fmaf(acc_s[kb][s2 * 8 + i], cs, mc_run));
                if (!MFMA_ROWSUM) l_run += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
                vec8 pv;
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = (T)t[i];
                pf[kb][s2] = pv;
            }
        };
        auto pv_mfma = [&](int kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const vec8 vf = *reinterpret_cast<const vec8*>(vs_ + (db * 32 + l31) * 128 + (((kb * 4 + s2 * 2 + hh) ^ vswz) << 4));
                    acc_o[db] = mfma32(vf, pf[kb][s2], acc_o[db]);
                }
        };
        make_p(0);
        make_p(1);
        pv_mfma(0);
        pv_mfma(1);
        __syncthreads();
    };
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nfull, std::true_type{});

    // finalize: lane (query l31, half hh) holds O^T rows d = 32*db + 8g + 4hh + r in register 4g + r
    float l;
    if (MFMA_ROWSUM) {
        constexpr int R = D % 32;  // row of the ones inside the last block: register 4*(R/8) on half (R/4)&1
        l = __shfl(acc_o[DB - 1][4 * (R / 8) + (R % 4)], ((R / 4) & 1) * 32 + l31, 64);
    } else {
        l = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l;
    const int qrow = q0 + l31;
    if (!SLOT && p.lse && hh == 0 && qrow < p.Tq)
        p.lse[((int64_t)b * p.H + h) * p.Tq + qrow] = __builtin_amdgcn_logf(l) - mc_run;
    if (qrow < p.Tq) {
        T* orow = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Tq + qrow) * p.ldo + h * D;
        typedef T vec4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = db * 32 + g * 8 + hh * 4;
                if (dd + 4 <= D) {
                    vec4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (T)(acc_o[db][4 * g + r] * inv);
                    *reinterpret_cast<vec4*>(orow + dd) = o;
                }
            }
    }
}

template <typename T, int D>
static int launch_attn32(const ur_attn_desc& d, hipStream_t s) {
    constexpr size_t lds = 2 * (64 * 128 + (D + 31) / 32 * 32 * 128);
    constexpr bool HAS_SLOT = ((D + 15) / 16 * 16 > D) && ((D + 31) / 32 * 32 > D);
    static std::atomic<uint64_t> done_plain{0}, done_slot{0};  // per (instantiation, device)
    set_lds_limit_once(done_plain, reinterpret_cast<const void*>(&attention32_kernel<T, D, false>), (int)lds);
    if (HAS_SLOT) set_lds_limit_once(done_slot, reinterpret_cast<const void*>(&attention32_kernel<T, D, HAS_SLOT>), (int)lds);
    dim3 grid((d.Tq + 127) / 128, d.B * d.H);
    if (HAS_SLOT && d.scale <= 0.f)
        hipLaunchKernelGGL((attention32_kernel<T, D, HAS_SLOT>), grid, dim3(256), lds, s, d);
    else
        hipLaunchKernelGGL((attention32_kernel<T, D, false>), grid, dim3(256), lds, s, d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T, int D>
static int launch_attn(const ur_attn_desc& d, hipStream_t s) {
    if (d.q_hstride > 0 || d.k_hstride > 0) return UR_E_UNSUPPORTED;  // head-major images: the d <= 64 kernel only
    constexpr int DK = (D + 31) / 32 * 32, DV = (D + 15) / 16 * 16;
    constexpr size_t lds = 2 * (64 * DK * 2 + DV * 128);
    static std::atomic<uint64_t> done{0};  // per (instantiation, device)
    set_lds_limit_once(done, reinterpret_cast<const void*>(&attention_kernel<T, D>), (int)lds);
    dim3 grid((d.Tq + 127) / 128, d.B * d.H);
    hipLaunchKernelGGL((attention_kernel<T, D>), grid, dim3(256), lds, s, d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
static int launch_attn_d(const ur_attn_desc& d, hipStream_t s) {
    switch (d.d) {
        case 32: return launch_attn32<T, 32>(d, s);
        case 40: return launch_attn32<T, 40>(d, s);
        case 64: return launch_attn32<T, 64>(d, s);
        case 80: return launch_attn<T, 80>(d, s);
        case 128: return launch_attn<T, 128>(d, s);
        case 160: return launch_attn<T, 160>(d, s);
    }
    return UR_E_UNSUPPORTED;
}

}  // namespace ur

extern "C" int ur_attention(const ur_attn_desc* d, void* stream) {
    using namespace ur;
    if (!d || !d->q || !d->k || !d->vt || !d->o || !d->zero_page) return UR_E_BADARG;
    if (d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0) return UR_E_BADARG;
    if ((d->ldq & 7) || (d->ldk & 7) || (d->ldvt & 63) || (d->ldo & 3) || (d->q_off & 7) || (d->k_off & 7))
        return UR_E_BADARG;
    if (d->ldvt < (d->Tk + 63) / 64 * 64 || (d->vt_bstride & 7)) return UR_E_BADARG;
    if (d->lse && !(d->scale > 0.f)) return UR_E_BADARG;  // the pre-scaled (reference slot) mode keeps no row reference
    if (d->q_hstride < 0 || d->k_hstride < 0 || (d->q_hstride & 7) || (d->k_hstride & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == UR_DT_F16) return launch_attn_d<f16>(*d, s);
    if (d->dtype == UR_DT_BF16) return launch_attn_d<bf16>(*d, s);
    return UR_E_BADARG;
}
