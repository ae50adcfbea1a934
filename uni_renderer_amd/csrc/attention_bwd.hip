// Flash backward of o = softmax(q k^T * scale) v for the training path (device op 11; the reference reaches it through
// torch autograd over F.scaled_dot_product_attention, diffusers AttnProcessor2_0 called from
// /root/reference models/attention.py BasicTransformerBlock).  P is never materialised: two kernels recompute the
// score blocks from q, k on the MFMA and keep P / dS in registers.
//
//   inputs   q, o, do | k, v   [B][Tq][ld] | [B][Tk][ld] token matrices, head h at columns h*d .. h*d+d-1 (the reference's
//                               layout: no per-head copies; the q | k | v parts of a fused projection are column offsets)
//            qt, dot | kt       [B][H*d][ldt] transposes of those matrices (ur_transpose2d), ldt >= the padded token count
//   outputs  dq | dk, dv        same layouts as q | k, v
//            stats              [2][S][Tq]  fp32 workspace: row log-sum-exp (log2 units) | D = rowsum(do * o), S = B*H
//   Head dims are zero-extended to DP (64 / 96 / 160 for d = 40 / 80 / 160) and key rows to a multiple of 64 by the
//   loaders (bounds-checked 16-byte chunks), not in memory.
//
//   kernel 1 (dq):    a wave owns 16*NB queries (MFMA columns) and streams 64-key tiles: pass A the row log-sum-exp,
//                     pass B  S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - D),  dQ^T += K^T dS^T
//   kernel 2 (dkdv):  a wave owns 16*NB keys and streams 64-query tiles:
//                     S = Q K^T,  dP = dO V^T,  P = exp2(S s2 - lse),  dS = P o (dP - D),  dV^T += dO^T P,  dK^T += Q^T dS
//
// The score block leaves the 16x16x32 MFMA with lane (j = lane & 15, g = lane >> 4) holding rows 4g .. 4g+3 of column
// j.  Two such blocks ARE the B operand of the next MFMA when its k index is read as
//       k = 8g + i  ->  row 4g + i of block 0 (i < 4),  row 4g + i - 4 of block 1 (i >= 4)
// and the A operand (the transposed tile in LDS) is fetched with the same permutation (two 8-byte reads): no LDS round
// trip, no shuffles for P / dS.  Every sum is in a fixed order: deterministic, no atomics (dq and dk / dv come from
// different kernels instead of one kernel with atomic dq).
#include <cstdlib>

#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

struct AttnBwdArgs {
    const void *q, *k, *v, *o, *dout, *qt, *kt, *dot;  // pre-offset to the first column (row) of head 0 of their part
    int64_t ldq, ldk, ldv, ldo, lddo;                  // row strides (elements)
    int64_t ldqt, ldkt, lddot;                         // row strides of the transposed matrices
    float* stats;
    void *dq, *dk, *dv;
    int64_t lddq, lddk, lddv;
    float* part;       // [2][G][S][Tk][DP] fp32 partial dk | dv of the query splits (G > 1), else unused
    int S, H, d;       // S = B * H slices, head dim d (multiple of 8)
    int Tq, Tk;        // query rows (multiple of 64), key rows rounded up to a multiple of 64
    int Tk_valid;      // real key rows: rows >= Tk_valid read as zeros and P = 0 there (cross-attention: 77 of 128)
    int Tk_rows;       // rows per batch of the k / v / dk / dv matrices (>= Tk_valid)
    int G;             // query splits of the dk / dv kernel (blockIdx.z)
    float scale;
};

// LDS images: row-major tiles [64][DP] with rows padded by 16 bytes (row stride 2*DP + 16: the 16 rows of an MFMA
// fragment read start on 16 distinct 4-bank groups for DP = 64 / 96 / 160), transposed tiles [DP][64] with 144-byte rows.
template <int DP> struct BwdLds {
    static constexpr int RS = 2 * DP + 16;   // bytes per row of a [64][DP] tile
    static constexpr int TS = 144;           // bytes per row of a [DP][64] tile
    static constexpr int ROWS = 64 * RS;
    static constexpr int TRN = DP * TS;
};

// Staging of one tile is split into its global loads (into registers) and its LDS stores, so that the loads of tile
// t + 1 are in flight while tile t is multiplied (PF) or at least while the other waves reach the barrier.
#define UR_ROWREGS(DP) ((64 * ((DP) / 8)) / 256)
#define UR_TRNREGS(DP) (((DP) * 8) / 256)
template <typename T, int DP, bool GEN>
__device__ __forceinline__ void gload_rows(const T* __restrict__ g, int64_t ld, int rows_valid, int d,
                                           u32x4 (&r)[UR_ROWREGS(DP)], int tid) {
    constexpr int CPR = DP / 8;
    if (!GEN || (rows_valid >= 64 && d == DP)) {  // whole tile in bounds: plain loads (GEN = false: known at compile time)
#pragma unroll
        for (int i = 0; i < (64 * CPR) / 256; ++i) {
            const int e = tid + i * 256, row = e / CPR, c = e - row * CPR;
            r[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)row * ld + c * 8);
        }
    } else {  // out-of-range chunks read chunk 0 of the tile (always valid) and are zeroed: no divergent loads
#pragma unroll
        for (int i = 0; i < (64 * CPR) / 256; ++i) {
            const int e = tid + i * 256, row = e / CPR, c = e - row * CPR;
            const bool ok = row < rows_valid && c * 8 < d;
            const u32x4 v = *reinterpret_cast<const u32x4*>(g + (ok ? (int64_t)row * ld + c * 8 : 0));
            r[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
}
template <int DP>
__device__ __forceinline__ void lstore_rows(char* lds, const u32x4 (&r)[UR_ROWREGS(DP)], int tid) {
    constexpr int CPR = DP / 8;
#pragma unroll
    for (int i = 0; i < (64 * CPR) / 256; ++i) {
        const int e = tid + i * 256, row = e / CPR, c = e - row * CPR;
        *reinterpret_cast<u32x4*>(lds + row * BwdLds<DP>::RS + c * 16) = r[i];
    }
}
template <typename T, int DP, bool GEN>
__device__ __forceinline__ void gload_trn(const T* __restrict__ g, int64_t ld, int d, u32x4 (&r)[UR_TRNREGS(DP)], int tid) {
    if (!GEN || d == DP) {
#pragma unroll
        for (int i = 0; i < (DP * 8) / 256; ++i) {
            const int e = tid + i * 256, row = e >> 3, c = e & 7;
            r[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)row * ld + c * 8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < (DP * 8) / 256; ++i) {
            const int e = tid + i * 256, row = e >> 3, c = e & 7;
            const bool ok = row < d;
            const u32x4 v = *reinterpret_cast<const u32x4*>(g + (ok ? (int64_t)row * ld : 0) + c * 8);
            r[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
}
template <int DP>
__device__ __forceinline__ void lstore_trn(char* lds, const u32x4 (&r)[UR_TRNREGS(DP)], int tid) {
#pragma unroll
    for (int i = 0; i < (DP * 8) / 256; ++i) {
        const int e = tid + i * 256, row = e >> 3, c = e & 7;
        *reinterpret_cast<u32x4*>(lds + row * BwdLds<DP>::TS + c * 16) = r[i];
    }
}
// A fragment of a row-major tile: row (16 * blk + j), k = 32 * ks + 8 * g .. + 7
template <typename T, int DP>
__device__ __forceinline__ typename Vec8<T>::type frag_rows(const char* lds, int blk, int ks, int j, int g) {
    return *reinterpret_cast<const typename Vec8<T>::type*>(lds + (16 * blk + j) * BwdLds<DP>::RS + (ks * 4 + g) * 16);
}
// A fragment of a transposed tile with the permuted k order: row (16 * blk + j), columns 32 * ks + 4g .. +3 and + 16
template <typename T, int DP>
__device__ __forceinline__ typename Vec8<T>::type frag_trn(const char* lds, int blk, int ks, int j, int g) {
    const char* r = lds + (16 * blk + j) * BwdLds<DP>::TS + ks * 64 + g * 8;
    const uint2 a = *reinterpret_cast<const uint2*>(r), b = *reinterpret_cast<const uint2*>(r + 32);
    const uint4 u = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(typename Vec8<T>::type, u);
}
// Round 4: the same fragments WITHOUT a transposed tile (and without the transposed copies q^T, k^T, dO^T in memory): the LDS
// transpose read gathers them from the row-major tile that is staged for the other GEMMs anyway.  Within a 16-lane group, lane
// 4 rr + q4 supplies the address of tile row (k index) rr, columns 4 q4 .. 4 q4 + 3, and lane i receives column i of that
// 4 x 16 block (ds_read_b64_tr_b16; tools/ubench/tr_probe.hip): two reads = the 8 permuted k values.  -DUR_ATTN_BWD_TRN=1
// builds the former transposed-tile form (the A/B of tools/experiments/r04_run36.sh).
#ifndef UR_ATTN_BWD_TRN
#define UR_ATTN_BWD_TRN 0
#endif
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x4_t lds_tr16(const char* q) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)q);
}
// MFMA 16x16x32 A operand: row = tile column 16 blk + (lane & 15), k = tile rows 32 ks + 4 g + (0..3) and + 16
template <typename T, int DP>
__device__ __forceinline__ typename Vec8<T>::type frag_tr(const char* lds, int blk, int ks, int lane) {
    const int i16 = lane & 15, g = lane >> 4;
    const char* r = lds + (32 * ks + 4 * g + (i16 >> 2)) * BwdLds<DP>::RS + (16 * blk + 4 * (i16 & 3)) * 2;
    const s16x4_t a = lds_tr16(r), b = lds_tr16(r + 16 * BwdLds<DP>::RS);
    return __builtin_bit_cast(typename Vec8<T>::type, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// MFMA 32x32x16 A operand: row = tile column 32 blk + (lane & 31), k = tile rows row0 + 4 hh + (0..3) and + 8 (hh = lane >> 5)
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type frag_tr32(const char* lds, int blk, int row0, int lane) {
    const int i16 = lane & 15, gi = lane >> 4;
    const char* r = lds + (row0 + 4 * (gi >> 1) + (i16 >> 2)) * BwdLds<64>::RS + (32 * blk + 16 * (gi & 1) + 4 * (i16 & 3)) * 2;
    const s16x4_t a = lds_tr16(r), b = lds_tr16(r + 8 * BwdLds<64>::RS);
    return __builtin_bit_cast(typename Vec8<T>::type, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type pack2(const f32x4& a, const f32x4& b) {
    typename Vec8<T>::type v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (T)a[i]; v[4 + i] = (T)b[i]; }
    return v;
}
template <typename T, typename V, bool GEN>
__device__ __forceinline__ V load_or_zero(const T* p, bool ok) {
    if (!GEN) return *reinterpret_cast<const V*>(p);
    V v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)0.0f;
    if (ok) v = *reinterpret_cast<const V*>(p);
    return v;
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const f32x4& a, float scale) {
    T h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (T)(a[i] * scale);
    uint2 u;
    __builtin_memcpy(&u, h, 8);
    *reinterpret_cast<uint2*>(p) = u;
}

// ---------------------------------------------------------------------------------------------------------------
// kernel 1: row statistics + dq.  grid (T / (64 * NB), S), 4 waves, wave w owns queries (4 * bx + w) * 16 * NB ...
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int DP, int NB, bool PF, bool HAS_LSE, bool MASK, bool GEN>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const AttnBwdArgs p) {
    typedef typename Vec8<T>::type vec8;
    typedef BwdLds<DP> L;
    constexpr int KS = DP / 32, DB = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + L::ROWS;
    char* Kts = smem + 2 * L::ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    const int s = blockIdx.y, Tn = p.Tk, Tq = p.Tq;  // Tn: the streamed (key) side
    const int b = s / p.H, hd = (s - b * p.H) * p.d, d = p.d, C = p.H * p.d;
    const int qbase = (blockIdx.x * 4 + wave) * 16 * NB;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tq * p.ldq + hd;
    const T* O = reinterpret_cast<const T*>(p.o) + (int64_t)b * Tq * p.ldo + hd;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (int64_t)b * Tq * p.lddo + hd;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk_rows * p.ldk + hd;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Tk_rows * p.ldv + hd;
    const T* Kt = reinterpret_cast<const T*>(p.kt) + ((int64_t)b * C + hd) * p.ldkt;  // rows = head dims, columns = keys
    const float s2 = p.scale * 1.44269504088896341f;

    // stationary B operands: this lane's query rows, k = 32 ks + 8 g .. + 7; D = rowsum(dO o O) on the way
    vec8 qf[NB][KS], dof[NB][KS];
    float dsum[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int64_t row = qbase + 16 * nb + j;
        float a = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int col = 32 * ks + 8 * g;
            qf[nb][ks] = load_or_zero<T, vec8, GEN>(Q + row * p.ldq + col, col < d);
            dof[nb][ks] = load_or_zero<T, vec8, GEN>(dO + row * p.lddo + col, col < d);
            const vec8 of = load_or_zero<T, vec8, GEN>(O + row * p.ldo + col, col < d);
#pragma unroll
            for (int i = 0; i < 8; ++i) a = fmaf((float)dof[nb][ks][i], (float)of[i], a);
        }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        dsum[nb] = a;
    }

    // ---- pass A: log-sum-exp of every query row (log2 units); skipped when the forward kernel handed it over ----
    u32x4 rk[UR_ROWREGS(DP)], rv[UR_ROWREGS(DP)], rkt[UR_TRNREGS(DP)];
    float lse[NB];
    if constexpr (HAS_LSE) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            lse[nb] = p.stats[(int64_t)s * Tq + qbase + 16 * nb + j];
            if (g == 0) p.stats[(int64_t)(p.S + s) * Tq + qbase + 16 * nb + j] = dsum[nb];
        }
    } else {
        float mx[NB], ls[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { mx[nb] = -1e30f; ls[nb] = 0.f; }
        if (PF) gload_rows<T, DP, GEN>(K, p.ldk, p.Tk_valid, d, rk, tid);
        for (int kt = 0; kt < Tn; kt += 64) {
            if (!PF) gload_rows<T, DP, GEN>(K + (int64_t)kt * p.ldk, p.ldk, p.Tk_valid - kt, d, rk, tid);
            __syncthreads();
            lstore_rows<DP>(Ks, rk, tid);
            __syncthreads();
            if (PF && kt + 64 < Tn) gload_rows<T, DP, GEN>(K + (int64_t)(kt + 64) * p.ldk, p.ldk, p.Tk_valid - kt - 64, d, rk, tid);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f32x4 sc[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) sc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 a = frag_rows<T, DP>(Ks, kb, ks, j, g);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) sc[nb] = mfma16(a, qf[nb][ks], sc[nb]);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v0 = sc[nb][0] * s2, v1 = sc[nb][1] * s2, v2 = sc[nb][2] * s2, v3 = sc[nb][3] * s2;
                    if (MASK) {  // padded keys leave the softmax
                        const int key = kt + 16 * kb + 4 * g;
                        if (key + 0 >= p.Tk_valid) v0 = -1e30f;
                        if (key + 1 >= p.Tk_valid) v1 = -1e30f;
                        if (key + 2 >= p.Tk_valid) v2 = -1e30f;
                        if (key + 3 >= p.Tk_valid) v3 = -1e30f;
                    }
                    const float mn = fmaxf(fmaxf(mx[nb], fmaxf(v0, v1)), fmaxf(v2, v3));
                    ls[nb] = ls[nb] * __builtin_amdgcn_exp2f(mx[nb] - mn) + __builtin_amdgcn_exp2f(v0 - mn) +
                             __builtin_amdgcn_exp2f(v1 - mn) + __builtin_amdgcn_exp2f(v2 - mn) + __builtin_amdgcn_exp2f(v3 - mn);
                    mx[nb] = mn;
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {  // the four lanes (g = 0..3) of a query, fixed order
                const float mo = __shfl_xor(mx[nb], o, 64), lo = __shfl_xor(ls[nb], o, 64);
                const float mn = fmaxf(mx[nb], mo);
                ls[nb] = ls[nb] * __builtin_amdgcn_exp2f(mx[nb] - mn) + lo * __builtin_amdgcn_exp2f(mo - mn);
                mx[nb] = mn;
            }
            lse[nb] = mx[nb] + __builtin_amdgcn_logf(ls[nb]);  // v_log_f32 = log2
            if (g == 0) {
                p.stats[(int64_t)s * Tq + qbase + 16 * nb + j] = lse[nb];
                p.stats[(int64_t)(p.S + s) * Tq + qbase + 16 * nb + j] = dsum[nb];
            }
        }
    }

    // ---- pass B: dQ^T[d][query] += K^T[d][key] dS^T[key][query] ----
    f32x4 acc[DB][NB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[db][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#define UR_GLOAD_KV(kt_)                                          \
    do {                                                          \
        gload_rows<T, DP, GEN>(K + (int64_t)(kt_) * p.ldk, p.ldk, p.Tk_valid - (kt_), d, rk, tid); \
        gload_rows<T, DP, GEN>(V + (int64_t)(kt_) * p.ldv, p.ldv, p.Tk_valid - (kt_), d, rv, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(Kt + (kt_), p.ldkt, d, rkt, tid); \
    } while (0)
    if (PF) UR_GLOAD_KV(0);
    for (int kt = 0; kt < Tn; kt += 64) {
        if (!PF) UR_GLOAD_KV(kt);
        __syncthreads();
        lstore_rows<DP>(Ks, rk, tid);
        lstore_rows<DP>(Vs, rv, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(Kts, rkt, tid);
        __syncthreads();
        if (PF && kt + 64 < Tn) UR_GLOAD_KV(kt + 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // 32 keys at a time = one k step of the dQ MFMA
            f32x4 sc[2][NB], dp[2][NB];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { sc[b][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[b][nb] = sc[b][nb]; }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 ka = frag_rows<T, DP>(Ks, 2 * h + b, ks, j, g);
                    const vec8 va = frag_rows<T, DP>(Vs, 2 * h + b, ks, j, g);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        sc[b][nb] = mfma16(ka, qf[nb][ks], sc[b][nb]);
                        dp[b][nb] = mfma16(va, dof[nb][ks], dp[b][nb]);
                    }
                }
            vec8 dsf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = __builtin_amdgcn_exp2f(fmaf(sc[b][nb][r], s2, -lse[nb]));
                        if (MASK && kt + 16 * (2 * h + b) + 4 * g + r >= p.Tk_valid) pr = 0.f;
                        sc[b][nb][r] = pr * (dp[b][nb][r] - dsum[nb]);
                    }
                dsf[nb] = pack2<T>(sc[0][nb], sc[1][nb]);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const vec8 a = UR_ATTN_BWD_TRN ? frag_trn<T, DP>(Kts, db, h, j, g) : frag_tr<T, DP>(Ks, db, h, lane);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma16(a, dsf[nb], acc[db][nb]);
            }
        }
    }
    T* dQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * Tq * p.lddq + hd;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
            if (!GEN || 16 * db + 4 * g + 4 <= d)
                store4<T>(dQ + (int64_t)(qbase + 16 * nb + j) * p.lddq + 16 * db + 4 * g, acc[db][nb], p.scale);
}

// ---------------------------------------------------------------------------------------------------------------
// kernel 2: dk, dv.  grid (T / (64 * NB), S), wave w owns keys (4 * bx + w) * 16 * NB ...
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int DP, int NB, bool PF, bool SPLIT, bool GEN>
__global__ void __launch_bounds__(256) attn_bwd_dkdv_kernel(const AttnBwdArgs p) {
    typedef typename Vec8<T>::type vec8;
    typedef BwdLds<DP> L;
    constexpr int KS = DP / 32, DB = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* dOs = smem + L::ROWS;
    char* Qts = smem + 2 * L::ROWS;
    char* dOts = Qts + L::TRN;
    float* st = reinterpret_cast<float*>(UR_ATTN_BWD_TRN ? dOts + L::TRN : smem + 2 * L::ROWS);  // [0..63] lse, [64..127] D of the query tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    const int s = blockIdx.y, Tn = p.Tq, Tk = p.Tk;  // Tn: the streamed (query) side
    const int b = s / p.H, hd = (s - b * p.H) * p.d, d = p.d, C = p.H * p.d;
    const int kbase = (blockIdx.x * 4 + wave) * 16 * NB;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ldq + hd;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (int64_t)b * Tn * p.lddo + hd;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk_rows * p.ldk + hd;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Tk_rows * p.ldv + hd;
    const T* Qt = reinterpret_cast<const T*>(p.qt) + ((int64_t)b * C + hd) * p.ldqt;
    const T* dOt = reinterpret_cast<const T*>(p.dot) + ((int64_t)b * C + hd) * p.lddot;
    const float* lse_g = p.stats + (int64_t)s * Tn;
    const float* dsum_g = p.stats + (int64_t)(p.S + s) * Tn;
    // SPLIT: blockIdx.z owns one of G contiguous query ranges and leaves fp32 partial sums (few keys, many queries: the
    // 77-key cross-attention would otherwise run on S workgroups)
    const int qlen = SPLIT ? (Tn / 64 + p.G - 1) / p.G * 64 : Tn;
    const int q_beg = SPLIT ? (int)blockIdx.z * qlen : 0, q_end = SPLIT ? min(Tn, q_beg + qlen) : Tn;
    const float s2 = p.scale * 1.44269504088896341f;

    vec8 kf[NB][KS], vf[NB][KS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int64_t row = kbase + 16 * nb + j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int col = 32 * ks + 8 * g;
            const bool ok = row < p.Tk_valid && col < d;  // GEN = false: the rows exist (zero padded) and d == DP
            kf[nb][ks] = load_or_zero<T, vec8, GEN>(K + row * p.ldk + col, ok);
            vf[nb][ks] = load_or_zero<T, vec8, GEN>(V + row * p.ldv + col, ok);
        }
    }
    f32x4 dk[DB][NB], dv[DB][NB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { dk[db][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db][nb] = dk[db][nb]; }

    u32x4 rq[UR_ROWREGS(DP)], rdo[UR_ROWREGS(DP)], rqt[UR_TRNREGS(DP)], rdot[UR_TRNREGS(DP)];
    float rst = 0.f;
    const float* st_g = tid < 64 ? lse_g + tid : dsum_g + (tid & 63);  // tid < 128 stage the row statistics
#define UR_GLOAD_Q(qt_)                                           \
    do {                                                          \
        gload_rows<T, DP, GEN>(Q + (int64_t)(qt_) * p.ldq, p.ldq, 64, d, rq, tid);     \
        gload_rows<T, DP, GEN>(dO + (int64_t)(qt_) * p.lddo, p.lddo, 64, d, rdo, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(Qt + (qt_), p.ldqt, d, rqt, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(dOt + (qt_), p.lddot, d, rdot, tid); \
        if (tid < 128) rst = st_g[qt_];                           \
    } while (0)
    if (PF && q_beg < q_end) UR_GLOAD_Q(q_beg);
    for (int qt = q_beg; qt < q_end; qt += 64) {
        if (!PF) UR_GLOAD_Q(qt);
        __syncthreads();
        lstore_rows<DP>(Qs, rq, tid);
        lstore_rows<DP>(dOs, rdo, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(Qts, rqt, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(dOts, rdot, tid);
        if (tid < 128) st[tid] = rst;
        __syncthreads();
        if (PF && qt + 64 < q_end) UR_GLOAD_Q(qt + 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // 32 queries at a time
            f32x4 sc[2][NB], dp[2][NB];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { sc[b][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[b][nb] = sc[b][nb]; }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 qa = frag_rows<T, DP>(Qs, 2 * h + b, ks, j, g);
                    const vec8 da = frag_rows<T, DP>(dOs, 2 * h + b, ks, j, g);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        sc[b][nb] = mfma16(qa, kf[nb][ks], sc[b][nb]);
                        dp[b][nb] = mfma16(da, vf[nb][ks], dp[b][nb]);
                    }
                }
            vec8 pf[NB], dsf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float4 l4 = *reinterpret_cast<const float4*>(st + 32 * h + 16 * b + 4 * g);
                    const float4 d4 = *reinterpret_cast<const float4*>(st + 64 + 32 * h + 16 * b + 4 * g);
                    const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(fmaf(sc[b][nb][r], s2, -lr[r]));
                        sc[b][nb][r] = pr;
                        dp[b][nb][r] = pr * (dp[b][nb][r] - dr[r]);
                    }
                }
                pf[nb] = pack2<T>(sc[0][nb], sc[1][nb]);
                dsf[nb] = pack2<T>(dp[0][nb], dp[1][nb]);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const vec8 da = UR_ATTN_BWD_TRN ? frag_trn<T, DP>(dOts, db, h, j, g) : frag_tr<T, DP>(dOs, db, h, lane);
                const vec8 qa = UR_ATTN_BWD_TRN ? frag_trn<T, DP>(Qts, db, h, j, g) : frag_tr<T, DP>(Qs, db, h, lane);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    dv[db][nb] = mfma16(da, pf[nb], dv[db][nb]);
                    dk[db][nb] = mfma16(qa, dsf[nb], dk[db][nb]);
                }
            }
        }
    }
    if constexpr (SPLIT) {
        float* pk = p.part + (((int64_t)blockIdx.z * p.S + s) * Tk) * DP;
        float* pv = pk + (int64_t)p.G * p.S * Tk * DP;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int64_t o = (int64_t)(kbase + 16 * nb + j) * DP + 16 * db + 4 * g;
                *reinterpret_cast<f32x4*>(pk + o) = dk[db][nb];
                *reinterpret_cast<f32x4*>(pv + o) = dv[db][nb];
            }
    } else {
        T* dK = reinterpret_cast<T*>(p.dk) + (int64_t)b * p.Tk_rows * p.lddk + hd;
        T* dV = reinterpret_cast<T*>(p.dv) + (int64_t)b * p.Tk_rows * p.lddv + hd;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int64_t row = kbase + 16 * nb + j;
                const int col = 16 * db + 4 * g;
                if (!GEN || (row < p.Tk_valid && col + 4 <= d)) {
                    store4<T>(dK + row * p.lddk + col, dk[db][nb], p.scale);
                    store4<T>(dV + row * p.lddv + col, dv[db][nb], 1.0f);
                }
            }
    }
}

// dk | dv = sum over the G query splits, in split order (fixed order); one thread per 4 elements of [S][Tk][DP]
template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_fold_kernel(const AttnBwdArgs p, int DP, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const bool is_v = blockIdx.y == 1;
    const int col = (int)((i * 4) % DP);
    const int64_t sr = (i * 4) / DP;
    const int row = (int)(sr % p.Tk), s = (int)(sr / p.Tk);
    if (row >= p.Tk_valid || col + 4 > p.d) return;
    const float* src = p.part + (is_v ? (int64_t)p.G * n4 * 4 : 0) + i * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    for (int z = 1; z < p.G; ++z) a += *reinterpret_cast<const f32x4*>(src + (int64_t)z * n4 * 4);
    const int b = s / p.H, hd = (s - b * p.H) * p.d;
    T* dst = reinterpret_cast<T*>(is_v ? p.dv : p.dk);
    const int64_t ld = is_v ? p.lddv : p.lddk;
    store4<T>(dst + ((int64_t)b * p.Tk_rows + row) * ld + hd + col, a, is_v ? 1.0f : p.scale);
}

// ---------------------------------------------------------------------------------------------------------------
// The same two kernels on the 32x32x16 MFMA for the 64-wide padded head dim (d = 40, the 4096-token level where the
// time is).  The 32x32x16 shape does the same work in half the MFMA issue slots (both shapes run at their nominal 16 / 32
// cycles, profiles/r03_mfma_rate.txt); LDS bytes per FLOP are the same as the NB = 2 kernels above.
// A wave owns 32 columns (queries / keys); lane (j = lane & 31, hh = lane >> 5) receives, in register v, row
// 8 (v >> 2) + 4 hh + (v & 3) of column j: registers 8t .. 8t+7 ARE the B operand of k step t of the next MFMA when its k
// index is read as k = 8 hh + i -> row 16 t + 4 hh + i (i < 4), 16 t + 8 + 4 hh + i - 4 (i >= 4); the transposed tile is
// read with that permutation (two 8-byte reads).  The dq kernel needs the forward's log-sum-exp (has_lse).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type frag_rows32(const char* lds, int blk, int ks, int j, int hh) {
    return *reinterpret_cast<const typename Vec8<T>::type*>(lds + (32 * blk + j) * BwdLds<64>::RS + (ks * 2 + hh) * 16);
}
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type frag_trn32(const char* lds, int blk, int col0, int j, int hh) {
    const char* r = lds + (32 * blk + j) * BwdLds<64>::TS + (col0 + 4 * hh) * 2;
    const uint2 a = *reinterpret_cast<const uint2*>(r), b = *reinterpret_cast<const uint2*>(r + 16);
    const uint4 u = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(typename Vec8<T>::type, u);
}
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type pack8(const f32x16& a, int t) {
    typename Vec8<T>::type v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)a[8 * t + i];
    return v;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
template <typename T>
__device__ __forceinline__ void store4v(T* p, const f32x16& a, int g4, float scale) {
    T h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (T)(a[4 * g4 + i] * scale);
    uint2 u;
    __builtin_memcpy(&u, h, 8);
    *reinterpret_cast<uint2*>(p) = u;
}

template <typename T, bool MASK, bool GEN>
__global__ void __launch_bounds__(256) attn_bwd_dq32_kernel(const AttnBwdArgs p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int DP = 64;
    typedef BwdLds<DP> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + L::ROWS;
    char* Kts = smem + 2 * L::ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int s = blockIdx.y, Tn = p.Tk, Tq = p.Tq;
    const int b = s / p.H, hd = (s - b * p.H) * p.d, d = p.d, C = p.H * p.d;
    const int64_t row = (blockIdx.x * 4 + wave) * 32 + j;  // this lane's query
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tq * p.ldq + hd;
    const T* O = reinterpret_cast<const T*>(p.o) + (int64_t)b * Tq * p.ldo + hd;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (int64_t)b * Tq * p.lddo + hd;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk_rows * p.ldk + hd;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Tk_rows * p.ldv + hd;
    const T* Kt = reinterpret_cast<const T*>(p.kt) + ((int64_t)b * C + hd) * p.ldkt;
    const float s2 = p.scale * 1.44269504088896341f;

    vec8 qf[4], dof[4];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int col = 16 * ks + 8 * hh;
        qf[ks] = load_or_zero<T, vec8, GEN>(Q + row * p.ldq + col, col < d);
        dof[ks] = load_or_zero<T, vec8, GEN>(dO + row * p.lddo + col, col < d);
        const vec8 of = load_or_zero<T, vec8, GEN>(O + row * p.ldo + col, col < d);
#pragma unroll
        for (int i = 0; i < 8; ++i) dsum = fmaf((float)dof[ks][i], (float)of[i], dsum);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse = p.stats[(int64_t)s * Tq + row];
    if (hh == 0) p.stats[(int64_t)(p.S + s) * Tq + row] = dsum;

    f32x16 acc[2] = {zero16(), zero16()};
    u32x4 rk[UR_ROWREGS(DP)], rv[UR_ROWREGS(DP)], rkt[UR_TRNREGS(DP)];
#define UR_GLOAD_KV32(kt_)                                        \
    do {                                                          \
        gload_rows<T, DP, GEN>(K + (int64_t)(kt_) * p.ldk, p.ldk, p.Tk_valid - (kt_), d, rk, tid); \
        gload_rows<T, DP, GEN>(V + (int64_t)(kt_) * p.ldv, p.ldv, p.Tk_valid - (kt_), d, rv, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(Kt + (kt_), p.ldkt, d, rkt, tid); \
    } while (0)
    UR_GLOAD_KV32(0);
    for (int kt = 0; kt < Tn; kt += 64) {
        __syncthreads();
        lstore_rows<DP>(Ks, rk, tid);
        lstore_rows<DP>(Vs, rv, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(Kts, rkt, tid);
        __syncthreads();
        if (kt + 64 < Tn) UR_GLOAD_KV32(kt + 64);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {  // 32 keys at a time
            f32x16 sc = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sc = mfma32(frag_rows32<T>(Ks, kb, ks, j, hh), qf[ks], sc);
                dp = mfma32(frag_rows32<T>(Vs, kb, ks, j, hh), dof[ks], dp);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float pr = __builtin_amdgcn_exp2f(fmaf(sc[v], s2, -lse));
                if (MASK && kt + 32 * kb + 8 * (v >> 2) + 4 * hh + (v & 3) >= p.Tk_valid) pr = 0.f;
                sc[v] = pr * (dp[v] - dsum);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const vec8 dsf = pack8<T>(sc, t);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc[db] = mfma32(UR_ATTN_BWD_TRN ? frag_trn32<T>(Kts, db, 32 * kb + 16 * t, j, hh) : frag_tr32<T>(Ks, db, 32 * kb + 16 * t, lane), dsf, acc[db]);
            }
        }
    }
    T* dQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * Tq * p.lddq + hd;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int col = 32 * db + 8 * g4 + 4 * hh;
            if (!GEN || col + 4 <= d) store4v<T>(dQ + row * p.lddq + col, acc[db], g4, p.scale);
        }
}

template <typename T, bool SPLIT, bool GEN>
__global__ void __launch_bounds__(256) attn_bwd_dkdv32_kernel(const AttnBwdArgs p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int DP = 64;
    typedef BwdLds<DP> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* dOs = smem + L::ROWS;
    char* Qts = smem + 2 * L::ROWS;
    char* dOts = Qts + L::TRN;
    float* st = reinterpret_cast<float*>(UR_ATTN_BWD_TRN ? dOts + L::TRN : smem + 2 * L::ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int s = blockIdx.y, Tn = p.Tq, Tk = p.Tk;
    const int b = s / p.H, hd = (s - b * p.H) * p.d, d = p.d, C = p.H * p.d;
    const int64_t row = (blockIdx.x * 4 + wave) * 32 + j;  // this lane's key
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ldq + hd;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (int64_t)b * Tn * p.lddo + hd;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Tk_rows * p.ldk + hd;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Tk_rows * p.ldv + hd;
    const T* Qt = reinterpret_cast<const T*>(p.qt) + ((int64_t)b * C + hd) * p.ldqt;
    const T* dOt = reinterpret_cast<const T*>(p.dot) + ((int64_t)b * C + hd) * p.lddot;
    const float* lse_g = p.stats + (int64_t)s * Tn;
    const float* dsum_g = p.stats + (int64_t)(p.S + s) * Tn;
    const float s2 = p.scale * 1.44269504088896341f;
    const int qlen = SPLIT ? (Tn / 64 + p.G - 1) / p.G * 64 : Tn;
    const int q_beg = SPLIT ? (int)blockIdx.z * qlen : 0, q_end = SPLIT ? min(Tn, q_beg + qlen) : Tn;

    vec8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int col = 16 * ks + 8 * hh;
        const bool ok = row < p.Tk_valid && col < d;
        kf[ks] = load_or_zero<T, vec8, GEN>(K + row * p.ldk + col, ok);
        vf[ks] = load_or_zero<T, vec8, GEN>(V + row * p.ldv + col, ok);
    }
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    u32x4 rq[UR_ROWREGS(DP)], rdo[UR_ROWREGS(DP)], rqt[UR_TRNREGS(DP)], rdot[UR_TRNREGS(DP)];
    float rst = 0.f;
    const float* st_g = tid < 64 ? lse_g + tid : dsum_g + (tid & 63);
#define UR_GLOAD_Q32(qt_)                                         \
    do {                                                          \
        gload_rows<T, DP, GEN>(Q + (int64_t)(qt_) * p.ldq, p.ldq, 64, d, rq, tid);     \
        gload_rows<T, DP, GEN>(dO + (int64_t)(qt_) * p.lddo, p.lddo, 64, d, rdo, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(Qt + (qt_), p.ldqt, d, rqt, tid); \
        if (UR_ATTN_BWD_TRN) gload_trn<T, DP, GEN>(dOt + (qt_), p.lddot, d, rdot, tid); \
        if (tid < 128) rst = st_g[qt_];                           \
    } while (0)
    if (q_beg < q_end) UR_GLOAD_Q32(q_beg);
    for (int qt = q_beg; qt < q_end; qt += 64) {
        __syncthreads();
        lstore_rows<DP>(Qs, rq, tid);
        lstore_rows<DP>(dOs, rdo, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(Qts, rqt, tid);
        if (UR_ATTN_BWD_TRN) lstore_trn<DP>(dOts, rdot, tid);
        if (tid < 128) st[tid] = rst;
        __syncthreads();
        if (qt + 64 < q_end) UR_GLOAD_Q32(qt + 64);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {  // 32 queries at a time
            f32x16 sc = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sc = mfma32(frag_rows32<T>(Qs, qb, ks, j, hh), kf[ks], sc);
                dp = mfma32(frag_rows32<T>(dOs, qb, ks, j, hh), vf[ks], dp);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 l4 = *reinterpret_cast<const float4*>(st + 32 * qb + 8 * g4 + 4 * hh);
                const float4 d4 = *reinterpret_cast<const float4*>(st + 64 + 32 * qb + 8 * g4 + 4 * hh);
                const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(sc[4 * g4 + r], s2, -lr[r]));
                    sc[4 * g4 + r] = pr;
                    dp[4 * g4 + r] = pr * (dp[4 * g4 + r] - dr[r]);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const vec8 pf = pack8<T>(sc, t), dsf = pack8<T>(dp, t);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = mfma32(UR_ATTN_BWD_TRN ? frag_trn32<T>(dOts, db, 32 * qb + 16 * t, j, hh) : frag_tr32<T>(dOs, db, 32 * qb + 16 * t, lane), pf, dv[db]);
                    dk[db] = mfma32(UR_ATTN_BWD_TRN ? frag_trn32<T>(Qts, db, 32 * qb + 16 * t, j, hh) : frag_tr32<T>(Qs, db, 32 * qb + 16 * t, lane), dsf, dk[db]);
                }
            }
        }
    }
    if constexpr (SPLIT) {
        float* pk = p.part + (((int64_t)blockIdx.z * p.S + s) * Tk) * DP;
        float* pv = pk + (int64_t)p.G * p.S * Tk * DP;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int64_t o = row * DP + 32 * db + 8 * g4 + 4 * hh;
                *reinterpret_cast<f32x4*>(pk + o) = f32x4{dk[db][4 * g4], dk[db][4 * g4 + 1], dk[db][4 * g4 + 2], dk[db][4 * g4 + 3]};
                *reinterpret_cast<f32x4*>(pv + o) = f32x4{dv[db][4 * g4], dv[db][4 * g4 + 1], dv[db][4 * g4 + 2], dv[db][4 * g4 + 3]};
            }
    } else {
        T* dK = reinterpret_cast<T*>(p.dk) + (int64_t)b * p.Tk_rows * p.lddk + hd;
        T* dV = reinterpret_cast<T*>(p.dv) + (int64_t)b * p.Tk_rows * p.lddv + hd;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * db + 8 * g4 + 4 * hh;
                if (!GEN || (row < p.Tk_valid && col + 4 <= d)) {
                    store4v<T>(dK + row * p.lddk + col, dk[db], g4, p.scale);
                    store4v<T>(dV + row * p.lddv + col, dv[db], g4, 1.0f);
                }
            }
    }
}

template <typename T, int DP, int NB, bool HAS_LSE, bool MASK, bool GEN>
static void launch_dq(const AttnBwdArgs& a, hipStream_t st) {
    typedef BwdLds<DP> L;
    constexpr bool PF = DP <= 64;  // register prefetch of the next tile: 32 VGPRs at DP = 64, too many above
    constexpr int lds_dq = 2 * L::ROWS + (UR_ATTN_BWD_TRN ? L::TRN : 0);
    static std::atomic<uint64_t> done{0};
    set_lds_limit_once(done, reinterpret_cast<const void*>(&attn_bwd_dq_kernel<T, DP, NB, PF, HAS_LSE, MASK, GEN>), lds_dq);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DP, NB, PF, HAS_LSE, MASK, GEN>), dim3(a.Tq / (64 * NB), a.S), dim3(256), lds_dq, st, a);
}
template <typename T, int DP, int NB, bool SPLIT, bool GEN>
static void launch_dkdv(const AttnBwdArgs& a, hipStream_t st) {
    typedef BwdLds<DP> L;
    constexpr bool PF = DP <= 64;
    constexpr int lds_kv = 2 * L::ROWS + (UR_ATTN_BWD_TRN ? 2 * L::TRN : 0) + 512;
    static std::atomic<uint64_t> done{0};
    set_lds_limit_once(done, reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<T, DP, NB, PF, SPLIT, GEN>), lds_kv);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, DP, NB, PF, SPLIT, GEN>), dim3(a.Tk / (64 * NB), a.S, SPLIT ? a.G : 1), dim3(256),
                       lds_kv, st, a);
    if (SPLIT) {
        const int64_t n4 = (int64_t)a.S * a.Tk * DP / 4;
        hipLaunchKernelGGL((attn_bwd_fold_kernel<T>), dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, st, a, DP, n4);
    }
}

static constexpr bool flash_m32() { return true; }  // the 32x32x16 kernels for d <= 64 (an environment toggle during their A/B runs)
// GEN = false: every tile is fully in bounds (d == DP, key rows allocated up to the padded count): the loaders, operand
// loads and stores carry no predicates -- the per-head-copy mode of the host side
static bool full_tiles(const AttnBwdArgs& a, int dp) { return a.d == dp && a.Tk_rows >= a.Tk; }

template <typename T, bool GEN>
static int launch_bwd32(const AttnBwdArgs& a, hipStream_t st) {
    typedef BwdLds<64> L;
    constexpr int lds_dq = 2 * L::ROWS + (UR_ATTN_BWD_TRN ? L::TRN : 0), lds_kv = 2 * L::ROWS + (UR_ATTN_BWD_TRN ? 2 * L::TRN : 0) + 512;
    const dim3 gq(a.Tq / 128, a.S), gk(a.Tk / 128, a.S, a.G);
    if (a.Tk_valid < a.Tk) hipLaunchKernelGGL((attn_bwd_dq32_kernel<T, true, GEN>), gq, dim3(256), lds_dq, st, a);
    else hipLaunchKernelGGL((attn_bwd_dq32_kernel<T, false, GEN>), gq, dim3(256), lds_dq, st, a);
    if (a.G > 1) {
        hipLaunchKernelGGL((attn_bwd_dkdv32_kernel<T, true, GEN>), gk, dim3(256), lds_kv, st, a);
        const int64_t n4 = (int64_t)a.S * a.Tk * 64 / 4;
        hipLaunchKernelGGL((attn_bwd_fold_kernel<T>), dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, st, a, 64, n4);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkdv32_kernel<T, false, GEN>), gk, dim3(256), lds_kv, st, a);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T, int DP, int NBQ, int NBK, bool HAS_LSE, bool GEN>
static int launch_bwd_g(const AttnBwdArgs& a, hipStream_t st) {
    if (a.Tk_valid < a.Tk) launch_dq<T, DP, NBQ, HAS_LSE, true, GEN>(a, st);
    else launch_dq<T, DP, NBQ, HAS_LSE, false, GEN>(a, st);
    if (a.G > 1) launch_dkdv<T, DP, NBK, true, GEN>(a, st);
    else launch_dkdv<T, DP, NBK, false, GEN>(a, st);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
template <typename T, int DP, int NBQ, int NBK, bool HAS_LSE>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
    return full_tiles(a, DP) ? launch_bwd_g<T, DP, NBQ, NBK, HAS_LSE, false>(a, st)
                             : launch_bwd_g<T, DP, NBQ, NBK, HAS_LSE, true>(a, st);
}

template <typename T, bool HAS_LSE>
static int dispatch_bwd(const AttnBwdArgs& a, int dp, hipStream_t st) {
    const bool wq = (a.Tq % 128) == 0, wk = (a.Tk % 128) == 0;
    switch (dp) {
        case 32: return launch_bwd<T, 32, 1, 1, HAS_LSE>(a, st);
        case 64:
            if (wq && wk && HAS_LSE && flash_m32())
                return full_tiles(a, 64) ? launch_bwd32<T, false>(a, st) : launch_bwd32<T, true>(a, st);
            if (wq && wk) return launch_bwd<T, 64, 2, 2, HAS_LSE>(a, st);
            if (wq) return launch_bwd<T, 64, 2, 1, HAS_LSE>(a, st);
            if (wk) return launch_bwd<T, 64, 1, 2, HAS_LSE>(a, st);
            return launch_bwd<T, 64, 1, 1, HAS_LSE>(a, st);
        case 96: return launch_bwd<T, 96, 1, 1, HAS_LSE>(a, st);
        case 160: return launch_bwd<T, 160, 1, 1, HAS_LSE>(a, st);
        default: return UR_E_BADARG;
    }
}

}  // namespace ur

extern "C" int ur_attention_backward_needs_transposes(void) { return UR_ATTN_BWD_TRN; }

extern "C" int ur_attention_backward_splits(int S, int Tq, int Tk, int dp) {
    // query splits of the dk / dv kernel: enough workgroups to fill the chip when there are few keys
    const int nbk = (dp == 64 && (Tk % 128) == 0) ? 2 : 1;
    const int64_t wgs = (int64_t)(Tk / (64 * nbk)) * S;
    int G = 1;
    while (wgs * G < 256 && G < 16 && (Tq / 64) >= 4 * G) G *= 2;
    return G;
}

extern "C" int ur_attention_backward(const ur_attn_bwd_desc* dsc, void* stream) {
    if (!dsc) return UR_E_BADARG;
    const ur_attn_bwd_desc& x = *dsc;
    if (UR_ATTN_BWD_TRN && (!x.qt || !x.kt || !x.dot)) return UR_E_BADARG;  // the transposed copies: not read any more (ABI 9)
    if (!x.q || !x.k || !x.v || !x.o || !x.dout || !x.stats || !x.dq || !x.dk || !x.dv || x.B <= 0 ||
        x.H <= 0 || x.d <= 0 || (x.d & 7) || x.Tq <= 0 || (x.Tq & 63) || x.Tk <= 0 || (x.Tk_rows != 0 && x.Tk_rows < x.Tk))
        return UR_E_BADARG;
    const int64_t lds[] = {x.ldq, x.ldk, x.ldv, x.ldo, x.lddo, x.lddq, x.lddk, x.lddv};
    for (int64_t ld : lds)
        if (ld <= 0 || (ld & 7)) return UR_E_BADARG;
    const int S = x.B * x.H, Tkp = (x.Tk + 63) / 64 * 64, dp = (x.d + 31) / 32 * 32;
    if (S > 65535) return UR_E_BADARG;
    if (UR_ATTN_BWD_TRN) {
        const int64_t ldt[] = {x.ldqt, x.ldkt, x.lddot};
        for (int64_t ld : ldt)
            if (ld <= 0 || (ld & 7)) return UR_E_BADARG;
        if (x.ldqt < x.Tq || x.lddot < x.Tq || x.ldkt < Tkp) return UR_E_BADARG;
    }
    const int G = ur_attention_backward_splits(S, x.Tq, Tkp, dp);
    if (G > 1 && !x.part) return UR_E_BADARG;
    ur::AttnBwdArgs a{x.q, x.k, x.v, x.o, x.dout, x.qt, x.kt, x.dot, x.ldq, x.ldk, x.ldv, x.ldo, x.lddo, x.ldqt, x.ldkt, x.lddot,
                      x.stats, x.dq, x.dk, x.dv, x.lddq, x.lddk, x.lddv, x.part, S, x.H, x.d, x.Tq, Tkp, x.Tk,
                      x.Tk_rows > 0 ? x.Tk_rows : x.Tk, G, x.scale};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool has_lse = x.has_lse != 0;
    if (x.dtype == UR_DT_F16) return has_lse ? ur::dispatch_bwd<ur::f16, true>(a, dp, st) : ur::dispatch_bwd<ur::f16, false>(a, dp, st);
    if (x.dtype == UR_DT_BF16) return has_lse ? ur::dispatch_bwd<ur::bf16, true>(a, dp, st) : ur::dispatch_bwd<ur::bf16, false>(a, dp, st);
    return UR_E_BADARG;
}

extern "C" int ur_attention_backward_supported(int Tq, int Tk, int d) {
    const int dp = (d + 31) / 32 * 32;
    return Tq > 0 && Tk > 0 && (Tq & 63) == 0 && d > 0 && (d & 7) == 0 && (dp == 32 || dp == 64 || dp == 96 || dp == 160);
}
