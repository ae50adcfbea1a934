// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace ur {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum : int { UR_F16 = 0, UR_BF16 = 1, UR_F32 = 2 };
enum : int { ACT_NONE = 0, ACT_SILU = 1, ACT_GEGLU = 2 };

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef f16x8 type; };
template <> struct Vec8<bf16> { typedef bf16x8 type; };

// D[i][j] += sum_k A[i][k] * B[k][j]; lane l feeds A[l&15][8*(l>>4)..+7] and B[8*(l>>4)..+7][l&15],
// and receives D[4*(l>>4)+r][l&15], r = 0..3 (cdna_hip_programming.md §3).
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 32x32x16: lane l feeds A[l&31][8*(l>>5)..+7] and B[8*(l>>5)..+7][l&31] and receives, for v = 0..15,
// D[8*(v>>2) + 4*(l>>5) + (v&3)][l&31].  Both shapes sustain 1.6-1.85 PFLOP/s on random operands (tools/ubench/mfma_rate.hip,
// profiles/r03_mfma_rate.txt); the igemm tiles are not MFMA-issue bound: a 32x32x16 build of them measured 9 % slower
// over the step's heaviest problems, so they stay on mfma16.
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Asynchronous 16-byte-per-lane global -> LDS copy.  The LDS destination is wave-uniform; the
// hardware adds lane*16.  The global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware workgroup remap (speed only, never correctness): the dispatcher places workgroup `bid` on
// XCD bid % 8 and each XCD has a private L2.  Returning logical ids so that every XCD owns ONE contiguous
// range of them keeps the tiles that share an operand panel on one L2 instead of replicating the panel
// over all eight.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU of diffusers' GEGLU.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, far below
// the fp16/bf16 rounding of the product): one v_exp, one v_rcp and 8 FMAs instead of libm's erff (~3x the
// instructions) -- the GEGLU epilogue of the K = 320 feed-forward GEMMs is a third of that kernel.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
    const float erf_abs = fmaf(-pl * t, e, 1.0f);            // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;             // x * Phi(x): the sign of erf folds into |x|
}

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&o)[8]) {
    typename Vec8<T>::type v = *reinterpret_cast<const typename Vec8<T>::type*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&o)[8]) {
    typename Vec8<T>::type v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)o[i];
    *reinterpret_cast<typename Vec8<T>::type*>(p) = v;
}


// ---------------------------------------------------------------------------------------------------------------
// Low parts of the (hi, lo) residual stream (include/ur_kernels.h).  The low part of an fp16 value v is
// lo = v - float(half(v)), |lo| <= ulp(hi)/2; three significant bits of it already push the pair's rounding error to
// 2^-14 relative, so fp16 streams store it as ONE byte: e5m2 = the high byte of the fp16 encoding of lo (round to
// nearest even).  bf16 streams keep a bf16 low part (e5m2 cannot hold bf16's exponent range).
// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct LoT { typedef T type; };
template <> struct LoT<f16> { typedef unsigned char type; };
template <typename T> using lo_t = typename LoT<T>::type;

__device__ __forceinline__ float lo_to_f(unsigned char b) {
    const unsigned short bits = (unsigned short)((unsigned)b << 8);
    return (float)__builtin_bit_cast(f16, bits);
}
__device__ __forceinline__ float lo_to_f(bf16 v) { return (float)v; }
__device__ __forceinline__ float lo_to_f(f16 v) { return (float)v; }
template <typename L> __device__ __forceinline__ L lo_from_f(float v);
template <> __device__ __forceinline__ unsigned char lo_from_f<unsigned char>(float v) {
    const unsigned bits = __builtin_bit_cast(unsigned short, (f16)v);
    return (unsigned char)((bits + 0x7Fu + ((bits >> 8) & 1u)) >> 8);  // round to nearest even on the dropped byte
}
template <> __device__ __forceinline__ bf16 lo_from_f<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 lo_from_f<f16>(float v) { return (f16)v; }

// N consecutive low parts (N = 2, 4, 8) -> float, one vector load
template <int N>
__device__ __forceinline__ void load_lo(const unsigned char* p, float (&v)[N]) {
    unsigned char b[N];
    if constexpr (N == 8) { const uint2 r = *reinterpret_cast<const uint2*>(p); __builtin_memcpy(b, &r, 8); }
    else if constexpr (N == 4) { const unsigned r = *reinterpret_cast<const unsigned*>(p); __builtin_memcpy(b, &r, 4); }
    else { const unsigned short r = *reinterpret_cast<const unsigned short*>(p); __builtin_memcpy(b, &r, 2); }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = lo_to_f(b[i]);
}
template <int N>
__device__ __forceinline__ void load_lo(const bf16* p, float (&v)[N]) {
    bf16 h[N];
    if constexpr (N == 8) { const uint4 r = *reinterpret_cast<const uint4*>(p); __builtin_memcpy(h, &r, 16); }
    else if constexpr (N == 4) { const uint2 r = *reinterpret_cast<const uint2*>(p); __builtin_memcpy(h, &r, 8); }
    else { const unsigned r = *reinterpret_cast<const unsigned*>(p); __builtin_memcpy(h, &r, 4); }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (float)h[i];
}
// v[i] - float(T(v[i])) for 8 values -> low parts, one vector store
template <typename T>
__device__ __forceinline__ void store_lo8(lo_t<T>* p, const float (&v)[8]) {
    lo_t<T> b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = lo_from_f<lo_t<T>>(v[i] - to_f(from_f<T>(v[i])));
    if constexpr (sizeof(lo_t<T>) == 1) { uint2 r; __builtin_memcpy(&r, b, 8); *reinterpret_cast<uint2*>(p) = r; }
    else { uint4 r; __builtin_memcpy(&r, b, 16); *reinterpret_cast<uint4*>(p) = r; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Host side: raise the dynamic-LDS limit of kernel `fn` on the CURRENT device once per (call site, device).  `done`
// is a `static std::atomic<uint64_t>` owned by the calling template instantiation, one bit per device ordinal.
// Setting the attribute twice is harmless (two host threads racing here set the same value), so there is no lock;
// it is the only mutable state in the library and it only caches an idempotent driver call.
inline void set_lds_limit_once(std::atomic<uint64_t>& done, const void* fn, int lds) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    done.fetch_or(bit, std::memory_order_release);
}

}  // namespace ur
