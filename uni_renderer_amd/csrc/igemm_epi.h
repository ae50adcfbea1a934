// Epilogue shared by the implicit-GEMM kernels (igemm.hip: LDS-tiled lock-step tiles; igemm_pp.hip: 8-wave ping-pong
// tiles): a lane holds 16 consecutive output channels of one pixel and applies bias, per-sample time-embedding row,
// SiLU / GEGLU, residual (+ low part), scale, and writes 32-byte vectors (+ the rounding remainder of the (hi, lo) stream).
#pragma once
#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

constexpr int BK = 64;  // elements per K chunk (128 bytes)

struct V16 { float v[16]; };
// the scalars of the descriptor the masked epilogue needs, passed BY VALUE: handing the kernel-argument struct to
// an out-of-line function by reference would force a scratch copy of it and turn every p.field into a scratch load
struct EpiArgs { int N, n_store, rows_per_b, ld_rowadd, act; int64_t ldc, ldres; float out_scale; };
// low parts of the (hi, lo) residual stream (include/ur_kernels.h): same leading dimensions as res / out
template <typename T> struct HiLo { const lo_t<T>* res_lo; lo_t<T>* out_lo; };  // low parts: e5m2 bytes (fp16) / bf16

// Rare path (ragged N tile, conv_out with 4 / 28 channels, unaligned leading dimensions): element-wise with
// masks.  Kept OUT OF LINE so that the hot kernels carry only the straight-line vector epilogue.
template <typename T>
__device__ __noinline__ void epilogue16_slow(EpiArgs p, T* __restrict__ outz, const float* biasz, const T* rowaddz,
                                             const T* resz, int m, int nc, V16 a, HiLo<T> hl) {
    float (&v)[16] = a.v;
    if (biasz) {
        for (int i = 0; i < 16; ++i)
            if (nc + i < p.N) v[i] += biasz[nc + i];
    }
    if (rowaddz) {
        const T* ra = rowaddz + (int64_t)(m / p.rows_per_b) * p.ld_rowadd + nc;
        for (int i = 0; i < 16; ++i)
            if (nc + i < p.N) v[i] += to_f(ra[i]);
    }
    if (p.act == ACT_GEGLU) {
        const int oc = nc >> 1;
        T* dst = outz + (int64_t)m * p.ldc + oc;
        for (int i = 0; i < 8; ++i) {
            const int src = (i < 4) ? i : 4 + i;  // value columns 0-3 / 8-11, gates 4 columns further
            if (oc + i < (p.N >> 1)) dst[i] = from_f<T>(v[src] * gelu_erf_f(v[src + 4]));
        }
        return;
    }
    if (p.act == ACT_SILU)
        for (int i = 0; i < 16; ++i) v[i] = silu_f(v[i]);
    if (resz) {
        const T* rp = resz + (int64_t)m * p.ldres + nc;
        for (int i = 0; i < 16; ++i)
            if (nc + i < p.N) v[i] += to_f(rp[i]);
        if (hl.res_lo) {
            const lo_t<T>* rl = hl.res_lo + (int64_t)m * p.ldres + nc;
            for (int i = 0; i < 16; ++i)
                if (nc + i < p.N) v[i] += lo_to_f(rl[i]);
        }
    }
    T* dst = outz + (int64_t)m * p.ldc + nc;
    lo_t<T>* dlo = hl.out_lo ? hl.out_lo + (int64_t)m * p.ldc + nc : nullptr;
    for (int i = 0; i < 16; ++i)
        if (nc + i < p.n_store) {
            const float y = v[i] * p.out_scale;
            const T h = from_f<T>(y);
            dst[i] = h;
            if (dlo) dlo[i] = lo_from_f<lo_t<T>>(y - to_f(h));
        }
}

// LEAN = true (the 3x3 conv instantiations): no transposed side output and no GEGLU -- neither exists for a conv (igemm_run rejects
// the combination), and every byte of epilogue code is inlined once per accumulator fragment of the wave tile (4 - 10 copies) and
// fetched cold by every workgroup of a launch: code size is launch latency (profiles/r06_slab_binary_ab.txt: +6 KB = +1.3 % step time).
template <typename T, bool LEAN = false>
__device__ __forceinline__ void epilogue16(const ur_igemm_desc& p, T* __restrict__ outz, const float* biasz,
                                           const T* rowaddz, const T* resz, int m, int nc, float (&v)[16], HiLo<T> hl,
                                           T* __restrict__ vtz = nullptr) {
    if (m >= p.M) return;
    if (!LEAN && vtz != nullptr && nc >= p.vt_n0) {
        // transposed side output (ur_igemm_desc.out_vt): this lane's 16 channels of token m go to 16 rows of V^T; the
        // lanes of a 16- / 32-lane group hold consecutive tokens, so every store instruction writes 32- / 64-byte runs
        if (nc + 16 > p.N) return;  // vt_n0 and N - vt_n0 are multiples of 16 (checked on the host)
        if (biasz) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += biasz[nc + i];
        }
        const int b = m / p.vt_rows, t = m - b * p.vt_rows;
        T* dst = vtz + (int64_t)b * p.vt_bstride + (int64_t)(nc - p.vt_n0) * p.ldvt + t;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(int64_t)i * p.ldvt] = from_f<T>(v[i]);
        return;
    }
    const bool vec = (((p.ldc | p.ldres | (int64_t)p.ld_rowadd) & 7) == 0);
    const int n_out_end = (!LEAN && p.act == ACT_GEGLU) ? (nc >> 1) + 8 : nc + 16;
    if (__builtin_expect(!(vec && nc + 16 <= p.N && n_out_end <= p.n_store), 0)) {
        V16 a;
#pragma unroll
        for (int i = 0; i < 16; ++i) a.v[i] = v[i];
        const EpiArgs e{p.N, p.n_store, p.rows_per_b, p.ld_rowadd, p.act, p.ldc, p.ldres, p.out_scale};
        epilogue16_slow<T>(e, outz, biasz, rowaddz, resz, m, nc, a, hl);
        return;
    }
    if (biasz) {
        const float4* b4 = reinterpret_cast<const float4*>(biasz + nc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 b = b4[i];
            v[4 * i + 0] += b.x; v[4 * i + 1] += b.y; v[4 * i + 2] += b.z; v[4 * i + 3] += b.w;
        }
    }
    if (rowaddz) {
        const T* ra = rowaddz + (int64_t)(m / p.rows_per_b) * p.ld_rowadd + nc;
        float t[8];
        load8(ra, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
        load8(ra + 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 + i] += t[i];
    }
    if (!LEAN && p.act == ACT_GEGLU) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = v[i] * gelu_erf_f(v[4 + i]);
            o[4 + i] = v[8 + i] * gelu_erf_f(v[12 + i]);
        }
        store8(outz + (int64_t)m * p.ldc + (nc >> 1), o);
        return;
    }
    if (p.act == ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = silu_f(v[i]);
    }
    if (resz) {
        const T* rp = resz + (int64_t)m * p.ldres + nc;
        float t[8];
        load8(rp, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
        load8(rp + 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 + i] += t[i];
        if (hl.res_lo) {
            const lo_t<T>* rl = hl.res_lo + (int64_t)m * p.ldres + nc;
            load_lo<8>(rl, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += t[i];
            load_lo<8>(rl + 8, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[8 + i] += t[i];
        }
    }
    if (p.out_scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= p.out_scale;
    }
    T* dst = outz + (int64_t)m * p.ldc + nc;
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = v[i];
    store8(dst, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = v[8 + i];
    store8(dst + 8, t);
    if (hl.out_lo) {  // rounding remainders v - float(T(v)) (exact in fp32), stored as e5m2 bytes / bf16
        lo_t<T>* dlo = hl.out_lo + (int64_t)m * p.ldc + nc;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[i];
        store_lo8<T>(dlo, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[8 + i];
        store_lo8<T>(dlo + 8, t);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue in two halves for the split-K second pass (round 6).  igemm_splitk_reduce is one thread per (row, 16 columns) with only
// ~2.5 waves per SIMD on the 16x16 / 8x8 maps, so its run time was the LENGTH OF ITS LOAD CHAIN: split-K slab after slab (a
// runtime-length load -> add loop), then bias -> wait -> time-embedding row -> wait -> residual -> wait -> low part -> wait ->
// store: 8 - 20 dependent memory round trips for 10 us of a launch that moves 47 MB.  epi_preload issues every epilogue
// operand load up front and UNCONDITIONALLY (an absent operand reads the zero page and is ignored), the caller then issues its
// slab loads, and epi_apply does the arithmetic in epilogue16's order (bit-identical results) and stores.  Straight-line case
// only (epi_split_ok); anything else goes through epilogue16.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct EpiOperands {
    float4 b4[4];                            // bias, 16 channels
    typename Vec8<T>::type ra[2], rs[2];     // time-embedding row, residual: 16 channels each
    uint2 lo[sizeof(lo_t<T>) == 1 ? 2 : 4];  // low part of the residual in 8-byte pieces (e5m2 rows are only 8-byte aligned)
};

template <typename T>
__device__ __forceinline__ bool epi_split_ok(const ur_igemm_desc& p, int nc, const void* vtz) {
    return (((p.ldc | p.ldres | (int64_t)p.ld_rowadd) & 7) == 0) && p.act != ACT_GEGLU && vtz == nullptr && nc + 16 <= p.N &&
           nc + 16 <= p.n_store;
}

template <typename T>
__device__ __forceinline__ void epi_preload(const ur_igemm_desc& p, EpiOperands<T>& op, const float* biasz, const T* rowaddz,
                                            const T* resz, const lo_t<T>* res_lo, int m, int nc, const void* zero_page) {
    typedef typename Vec8<T>::type vec8;
    const char* zp = reinterpret_cast<const char*>(zero_page);
    const float4* bs = biasz ? reinterpret_cast<const float4*>(biasz + nc) : reinterpret_cast<const float4*>(zp);
#pragma unroll
    for (int i = 0; i < 4; ++i) op.b4[i] = bs[i];
    const vec8* ra = rowaddz ? reinterpret_cast<const vec8*>(rowaddz + (int64_t)(m / p.rows_per_b) * p.ld_rowadd + nc)
                             : reinterpret_cast<const vec8*>(zp);
    const vec8* rs = resz ? reinterpret_cast<const vec8*>(resz + (int64_t)m * p.ldres + nc) : reinterpret_cast<const vec8*>(zp);
    op.ra[0] = ra[0]; op.ra[1] = ra[1];
    op.rs[0] = rs[0]; op.rs[1] = rs[1];
    const char* rl = res_lo ? reinterpret_cast<const char*>(res_lo + (int64_t)m * p.ldres + nc) : zp;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(op.lo) / sizeof(uint2)); ++i) op.lo[i] = reinterpret_cast<const uint2*>(rl)[i];
}

template <typename T>
__device__ __forceinline__ void epi_apply(const ur_igemm_desc& p, const EpiOperands<T>& op, T* __restrict__ outz, bool has_b,
                                          bool has_ra, bool has_rs, bool has_lo, lo_t<T>* out_lo, int m, int nc, float (&x)[16]) {
    if (has_b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[4 * i + 0] += op.b4[i].x; x[4 * i + 1] += op.b4[i].y; x[4 * i + 2] += op.b4[i].z; x[4 * i + 3] += op.b4[i].w;
        }
    }
    if (has_ra) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] += (float)op.ra[0][i]; x[8 + i] += (float)op.ra[1][i]; }
    }
    if (p.act == ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = silu_f(x[i]);
    }
    if (has_rs) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] += (float)op.rs[0][i]; x[8 + i] += (float)op.rs[1][i]; }
        if (has_lo) {
            lo_t<T> l[16];
            __builtin_memcpy(l, op.lo, sizeof(op.lo));
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] += lo_to_f(l[i]);
        }
    }
    if (p.out_scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] *= p.out_scale;
    }
    T* dst = outz + (int64_t)m * p.ldc + nc;
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = x[i];
    store8(dst, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = x[8 + i];
    store8(dst + 8, t);
    if (out_lo) {
        lo_t<T>* dlo = out_lo + (int64_t)m * p.ldc + nc;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = x[i];
        store_lo8<T>(dlo, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = x[8 + i];
        store_lo8<T>(dlo + 8, t);
    }
}

}  // namespace ur
