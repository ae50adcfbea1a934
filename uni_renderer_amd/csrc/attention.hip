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
    // tiles; per tile only a wave-uniform offset is added (64 keys further) and the ragged-tail test re-done.
    const char* kbase[KI];
    int krow[KI];
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int g = (wave * KI + it) * 64 + lane;  // linear 16-byte chunk of the K tile
        const int row = g / CPR, c = g - row * CPR;
        const int cl = c ^ k_swz<CPR>(row);
        krow[it] = (cl * 8 < D) ? row : (1 << 30);  // padded head-dim chunks always read zeros
        kbase[it] = reinterpret_cast<const char*>(K + (int64_t)row * p.ldk + cl * 8);
    }
    const char* vbase[VI];
    bool vok[VI];
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int row = (wave * VI + it) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ (lane >> 3);
        vok[it] = row < D;
        vbase[it] = reinterpret_cast<const char*>(VT + (int64_t)row * p.ldvt + cl * 8);
    }
    const char* zpc = reinterpret_cast<const char*>(zp);
    const int64_t kstep = (int64_t)64 * p.ldk * (int64_t)sizeof(T);

    auto stage = [&](int buf, int kt) {
        char* ks_ = smem + buf * STAGE;
        char* vs_ = ks_ + KT_BYTES;
        const int key0 = kt * 64;
        const int64_t koff = (int64_t)kt * kstep;
#pragma unroll
        for (int it = 0; it < KI; ++it) {
            const char* src = (key0 + krow[it] < p.Tk) ? kbase[it] + koff : zpc;
            glds16(src, ks_ + (wave * KI + it) * 1024);
        }
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int ii = wave * VI + it;  // 8-row group of the V^T tile
            if (ii < VROWS8 && !(MFMA_ROWSUM && ii * 8 >= D)) {
                const char* src = vok[it] ? vbase[it] + key0 * (int)sizeof(T) : zpc;
                glds16(src, vs_ + ii * 1024);
            }
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
    const float cs = p.scale * 1.44269504088896341f;  // softmax in base 2
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
    stage(0, 0);
    __syncthreads();
    // The tile body is instantiated twice: full tiles carry no key masking at all; only a ragged last tile
    // (Tk % 64 != 0, e.g. the 77 prompt tokens) pays for the per-score compare/select.
    auto tile = [&](const int kt, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
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
            float m4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = acc_s[i >> 1][i & 1][f];
                m4[i] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            }
            float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
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
    const int nfull = p.Tk >> 6;
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

template <typename T, int D>
static int launch_attn(const ur_attn_desc& d, hipStream_t s) {
    constexpr int DK = (D + 31) / 32 * 32, DV = (D + 15) / 16 * 16;
    constexpr size_t lds = 2 * (64 * DK * 2 + DV * 128);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<T, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    dim3 grid((d.Tq + 127) / 128, d.B * d.H);
    hipLaunchKernelGGL((attention_kernel<T, D>), grid, dim3(256), lds, s, d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
static int launch_attn_d(const ur_attn_desc& d, hipStream_t s) {
    switch (d.d) {
        case 32: return launch_attn<T, 32>(d, s);
        case 40: return launch_attn<T, 40>(d, s);
        case 64: return launch_attn<T, 64>(d, s);
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
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == UR_DT_F16) return launch_attn_d<f16>(*d, s);
    if (d->dtype == UR_DT_BF16) return launch_attn_d<bf16>(*d, s);
    return UR_E_BADARG;
}
