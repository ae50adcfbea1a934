// Building blocks of the backward pass (device op 11) for gfx950.  The FLOP-heavy gradients are
// GEMM-shaped and run on the forward implicit-GEMM kernel (ur_igemm) over transposed operands:
//     dX = dY . W            -> ur_igemm(x0 = dY [M][N],   w = W^T [K][N])
//     dW = dY^T . X          -> ur_igemm(x0 = dY^T [N][M], w = X^T [K][M])          (contraction over the M rows)
//     conv dX                -> ur_igemm conv3x3 of dY with the 180-degree rotated, channel-transposed weights
//     conv dW                -> ur_igemm(x0 = dY^T [N][P], w = im2col(X)^T [9C][P])
// This file holds what that needs around the GEMM -- an LDS-tiled transpose, the transposed im2col gather, column
// sums (bias / time-embedding gradients) -- and the backward of the memory-bound ops: SiLU, GEGLU, GroupNorm(+SiLU),
// LayerNorm.  All reductions are fixed-order (deterministic), statistics and sums in fp32.
#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

// ---------------------------------------------------------------------------------------------------------------
// dst[b][c][r] = src[b][r][c]   (R x C tiles of 64 x 64 through LDS, both sides in 16-byte vectors)
// rowmap != null: source row r is rowmap-free here; the im2col variant below gathers rows instead.
// ---------------------------------------------------------------------------------------------------------------
// Rows are packed in pairs on the way into LDS -- word (c, r/2) = (src[r][c], src[r+1][c]) -- so that the transposed
// side leaves with one 16-byte LDS read per thread (8 b32 writes + 1 b128 read per 16 elements instead of 16 + 16
// 2-byte accesses).  Word index inside a column is XOR-swizzled in 4-word groups to spread the banks.
__device__ __forceinline__ int tp_word(int c, int r2) { return c * 32 + ((((r2 >> 2) ^ (c >> 3)) & 7) << 2 | (r2 & 3)); }

template <typename T>
__device__ __forceinline__ float bits_to_f(uint16_t b) {
    T v;
    __builtin_memcpy(&v, &b, 2);
    return (float)v;
}

template <typename T, typename LoadRow>
__device__ __forceinline__ void transpose_tile_64x64(LoadRow load_row, uint32_t* tile, T* __restrict__ dst, int64_t ld_dst,
                                                     int c_valid, int r_valid8, int t) {
    typedef typename Vec8<T>::type vec8;
    {   // thread = (row pair r2 = t >> 3, 8 columns cv): two 16-byte global loads, eight packed words
        const int r2 = t >> 3, cv = (t & 7) * 8;
        const vec8 a = load_row(2 * r2, cv), b = load_row(2 * r2 + 1, cv);
        uint32_t ua[4], ub[4];  // (element 2j, element 2j+1) per word
        __builtin_memcpy(ua, &a, 16);
        __builtin_memcpy(ub, &b, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[tp_word(cv + 2 * j, r2)] = (ua[j] & 0xffffu) | (ub[j] << 16);
            tile[tp_word(cv + 2 * j + 1, r2)] = (ua[j] >> 16) | (ub[j] & 0xffff0000u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = pass * 32 + (t >> 3), g = t & 7;  // 8 destination elements = rows 8g .. 8g+7 of column c
        if (c < c_valid && g * 8 < r_valid8) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + c * 32 + (((g ^ (c >> 3)) & 7) << 2));
            *reinterpret_cast<uint4*>(dst + (int64_t)c * ld_dst + g * 8) = v;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) transpose2d_kernel(const T* __restrict__ src, int64_t ld_src, int64_t bs_src,
                                                          T* __restrict__ dst, int64_t ld_dst, int64_t bs_dst, int R,
                                                          int C) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * 32];
    const int t = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    src += (int64_t)blockIdx.z * bs_src;
    dst += (int64_t)blockIdx.z * bs_dst;
    typedef typename Vec8<T>::type vec8;
    auto load_row = [&](int r, int cv) {
        vec8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)0.0f;
        if (r0 + r < R && c0 + cv < C) v = *reinterpret_cast<const vec8*>(src + (int64_t)(r0 + r) * ld_src + c0 + cv);
        return v;
    };
    // rows R .. ceil8(R) of the source read as zeros
    transpose_tile_64x64<T>(load_row, tile, dst + (int64_t)c0 * ld_dst + r0, ld_dst, C - c0, ((R + 7) & ~7) - r0, t);
}

// Up to UR_TRANSPOSE_MAX independent (batched) transposes in ONE launch: the three operand transposes of a linear
// backward (w, dy, x) or q / k / dO of the flash attention backward are a few microseconds each and were paying one launch
// apiece.  Descriptors are kernel arguments; blockIdx.x walks the concatenated tile lists.
struct TransposeMultiArgs {
    ur_transpose_desc d[UR_TRANSPOSE_MAX];
    int tile0[UR_TRANSPOSE_MAX + 1];
    int n;
};
template <typename T>
__global__ void __launch_bounds__(256) transpose2d_multi_kernel(const TransposeMultiArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * 32];
    int k = 0, hi = a.n;  // last descriptor with tile0 <= blockIdx.x
    while (hi - k > 1) {
        const int mid = (k + hi) >> 1;
        if (a.tile0[mid] <= (int)blockIdx.x) k = mid; else hi = mid;
    }
    const ur_transpose_desc d = a.d[k];
    const int tc = (d.C + 63) / 64, tr = (d.R + 63) / 64;
    int id = (int)blockIdx.x - a.tile0[k];
    const int b = id / (tc * tr);
    id -= b * tc * tr;
    const int r0 = (id / tc) * 64, c0 = (id % tc) * 64;
    const T* src = reinterpret_cast<const T*>(d.src) + (int64_t)b * d.bs_src;
    T* dst = reinterpret_cast<T*>(d.dst) + (int64_t)b * d.bs_dst;
    const int t = threadIdx.x, R = d.R, C = d.C;
    const int64_t ld_src = d.ld_src;
    typedef typename Vec8<T>::type vec8;
    auto load_row = [&](int r, int cv) {
        vec8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)0.0f;
        if (r0 + r < R && c0 + cv < C) v = *reinterpret_cast<const vec8*>(src + (int64_t)(r0 + r) * ld_src + c0 + cv);
        return v;
    };
    transpose_tile_64x64<T>(load_row, tile, dst + (int64_t)c0 * d.ld_dst + r0, d.ld_dst, C - c0,
                            (d.rows_out > 0 ? d.rows_out : ((R + 7) & ~7)) - r0, t);
    if (!d.colsum) return;
    // ---- fused column sums of the source (the bias gradient of a linear / conv backward: dy is read here anyway) ----
    // The tile in LDS holds column c as 32 packed words (rows 2 w, 2 w + 1; rows >= R are zeros).  Thread c < 64 adds them
    // in fp32, starting at word c & 31 (bank spread; the order is fixed per column), and writes the tile's partial to
    // ws[row tile][column].  The last workgroup of a column tile (device-scope counter, reset for the next launch) adds
    // the partials in row-tile order: deterministic, no separate reduction launch.
    // Cross-workgroup visibility WITHOUT a fence: an agent-scope release (__threadfence) writes back the XCD's whole L2
    // -- measured +110 us per launch with 256+ workgroups doing it.  Instead the partials are stored / loaded with
    // relaxed AGENT-scope atomics (they bypass the non-coherent L2s), each storing thread waits for its store to be
    // acknowledged (vmcnt) before the workgroup barrier, and only then is the counter bumped.
    __shared__ float red[4][64];
    __shared__ int last;
    const int ct = c0 >> 6, rt = r0 >> 6;
    if (t < 64) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const uint32_t w = tile[t * 32 + ((j + t) & 31)];
            acc += bits_to_f<T>((uint16_t)(w & 0xffffu)) + bits_to_f<T>((uint16_t)(w >> 16));
        }
        if (c0 + t < C) __hip_atomic_store(d.colsum_ws + (int64_t)rt * C + c0 + t, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) {
        const unsigned prev = __hip_atomic_fetch_add(d.colsum_cnt + ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == (unsigned)(tr - 1);
        if (last) __hip_atomic_store(d.colsum_cnt + ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all others have arrived
    }
    __syncthreads();
    if (!last) return;
    const int nl = t & 63, q = t >> 6, n = c0 + nl;
    const int len = (tr + 3) >> 2, k0 = q * len, k1 = min(tr, k0 + len);
    float fs = 0.f;
    if (n < C) {
        // eight partials in flight per round trip (an agent-scope load goes to the memory side: ~1.5 us each way)
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = __hip_atomic_load(d.colsum_ws + (int64_t)(k + i) * C + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 8; ++i) fs += v[i];
        }
        for (; k < k1; ++k) fs += __hip_atomic_load(d.colsum_ws + (int64_t)k * C + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[q][nl] = fs;
    __syncthreads();
    if (q == 0 && n < C) d.colsum[n] = ((red[0][nl] + red[1][nl]) + red[2][nl]) + red[3][nl];
}

// Transposed im2col of a 3x3 / pad 1 convolution: out[(tap*C + c)][p] = x[pixel(p, tap)][c] (0 outside the image),
// p = (b, oy, ox) row-major; columns p >= P (padding to ld_out) are written as zeros by the caller's memset.
template <typename T>
__global__ void __launch_bounds__(256) im2col3x3_t_kernel(const T* __restrict__ x, int B, int H, int W, int C, int Ho,
                                                          int Wo, int stride, T* __restrict__ out, int64_t ld_out) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * 32];
    const int t = threadIdx.x;
    const int P = B * Ho * Wo;
    const int p0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tap = blockIdx.z;
    const int dy = tap / 3, dx = tap - dy * 3;
    typedef typename Vec8<T>::type vec8;
    auto load_row = [&](int r, int cv) {
        const int p = p0 + r;
        vec8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)0.0f;
        if (p < P && c0 + cv < C) {
            const int b = p / (Ho * Wo), rem = p - b * (Ho * Wo);
            const int oy = rem / Wo, ox = rem - oy * Wo;
            const int iy = oy * stride - 1 + dy, ix = ox * stride - 1 + dx;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const vec8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c0 + cv);
        }
        return v;
    };
    transpose_tile_64x64<T>(load_row, tile, out + ((int64_t)tap * C + c0) * ld_out + p0, ld_out, C - c0, (int)ld_out - p0, t);
}

// out[g][n] = sum over the rows m in [g*rpg, (g+1)*rpg) of x[m][n] (fp32).  grid = (N/64 column blocks, slices, groups):
// every group's rows are cut into ``slices`` contiguous pieces whose sums land in dst[(g*slices + s)][n]; with
// slices > 1 dst is the caller's workspace and colsum_fold_kernel adds the pieces in a fixed order.
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float (&v)[8]) { load8(p, v); }
template <>
__device__ __forceinline__ void load8f<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, int64_t ldx, int M, int N, int rpg,
                                                     int slices, float* __restrict__ dst, float* __restrict__ out,
                                                     unsigned* __restrict__ counters) {
    __shared__ float red[32][64 + 1];
    const int t = threadIdx.x, cv = (t & 7) * 8, rl = t >> 3;  // 8 vector columns x 32 row lanes
    const int n0 = blockIdx.x * 64, sl = blockIdx.y, g = blockIdx.z;
    const int gbeg = g * rpg, gend = min(M, gbeg + rpg);
    const int rps = (gend - gbeg + slices - 1) / slices;
    const int mbeg = gbeg + sl * rps, mend = min(gend, mbeg + rps);
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    if (n0 + cv < N) {
        int m = mbeg + rl;
        for (; m + 32 < mend; m += 64) {  // two rows in flight
            float v[8], w[8];
            load8f(x + (int64_t)m * ldx + n0 + cv, v);
            load8f(x + (int64_t)(m + 32) * ldx + n0 + cv, w);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += w[i];
        }
        if (m < mend) {
            float v[8];
            load8f(x + (int64_t)m * ldx + n0 + cv, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cv + i] = s[i];
    __syncthreads();
    if (!counters) {  // two-launch form: colsum_fold_kernel adds the slices
        if (t < 64 && n0 + t < N) {
            float a = 0.f;
            for (int k = 0; k < 32; ++k) a += red[k][t];  // fixed order
            dst[((int64_t)g * slices + sl) * N + n0 + t] = a;
        }
        return;
    }
    // one-launch form: the slice sums go to the workspace with agent-scope stores (they bypass the non-coherent L2s), the
    // last workgroup of a (group, column block) -- device-scope counter, left at zero -- adds them exactly as
    // colsum_fold_kernel would.  (The same hand-off cost every one of the thousands of small transposing workgroups a
    // memory round trip, tools/experiments/r03_run14.sh; here there are <= 512 long-running workgroups and the fold launch goes away.)
    __shared__ int last;
    if (t < 64 && n0 + t < N) {
        float a = 0.f;
        for (int k = 0; k < 32; ++k) a += red[k][t];  // fixed order
        __hip_atomic_store(dst + ((int64_t)g * slices + sl) * N + n0 + t, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) {
        unsigned* c = counters + (int64_t)g * gridDim.x + blockIdx.x;
        const unsigned prev = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == (unsigned)(slices - 1);
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    float (*r4)[64] = reinterpret_cast<float (*)[64]>(&red[0][0]);  // red is free again: [4][64] of it
    const int nl = t & 63, q = t >> 6, n = n0 + nl;
    const int len = (slices + 3) >> 2, k0 = q * len, k1 = min(slices, k0 + len);
    float a = 0.f;
    if (n < N)
        for (int k = k0; k < k1; ++k)
            a += __hip_atomic_load(dst + ((int64_t)g * slices + k) * N + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    r4[q][nl] = a;
    __syncthreads();
    if (q == 0 && n < N) out[(int64_t)g * N + n] = ((r4[0][nl] + r4[1][nl]) + r4[2][nl]) + r4[3][nl];
}

// 64 columns per workgroup, four 64-lane groups over contiguous quarters of the slices, partial sums added in group order
__global__ void __launch_bounds__(256) colsum_fold_kernel(const float* __restrict__ ws, int N, int slices,
                                                          float* __restrict__ out) {
    __shared__ float red[4][64];
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl, g = blockIdx.y;
    const int len = (slices + 3) >> 2, k0 = q * len, k1 = min(slices, k0 + len);
    float a = 0.f;
    if (n < N)
        for (int k = k0; k < k1; ++k) a += ws[((int64_t)g * slices + k) * N + n];
    red[q][nl] = a;
    __syncthreads();
    if (q == 0 && n < N) out[(int64_t)g * N + n] = ((red[0][nl] + red[1][nl]) + red[2][nl]) + red[3][nl];
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {  // d/dx x*sigmoid(x)
    const float s = sigmoid_f(x);
    return s * (1.0f + x * (1.0f - s));
}

template <typename T>
__global__ void __launch_bounds__(256) silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       T* __restrict__ dx, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], g[8];
        load8(x + i * 8, a);
        load8(dy + i * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = g[k] * silu_grad_f(a[k]);
        store8(dx + i * 8, a);
    }
}

// GEGLU in the reference's layout (diffusers GEGLU: hidden, gate = proj(x).chunk(2, -1)): h[m][0..D) values,
// h[m][D..2D) gates; forward y = value * gelu(gate) (erf), backward dvalue = dy*gelu(gate), dgate = dy*value*gelu'(gate).
template <typename T>
__global__ void __launch_bounds__(256) geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, int64_t M, int D) {
    const int dv = D >> 3;
    const int64_t total = M * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / dv;
        const int c = (int)(i - m * dv) * 8;
        float a[8], g[8];
        load8(h + m * 2 * D + c, a);
        load8(h + m * 2 * D + D + c, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = a[k] * (0.5f * g[k] * (1.0f + erff(g[k] * 0.70710678118654752f)));
        store8(y + m * D + c, a);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dy,
                                                        T* __restrict__ dh, int64_t M, int D) {
    const int dv = D >> 3;
    const int64_t total = M * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / dv;
        const int c = (int)(i - m * dv) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        load8(h + m * 2 * D + c, a);
        load8(h + m * 2 * D + D + c, g);
        load8(dy + m * D + c, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float phi = 0.5f * (1.0f + erff(g[k] * 0.70710678118654752f));             // Phi(g)
            const float pdf = 0.3989422804014327f * __expf(-0.5f * g[k] * g[k]);             // phi(g)
            da[k] = d[k] * g[k] * phi;
            dg[k] = d[k] * a[k] * (phi + g[k] * pdf);
        }
        store8(dh + m * 2 * D + c, da);
        store8(dh + m * 2 * D + D + c, dg);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward.  y = act(xhat * gamma + beta), xhat = (x - mean) * rstd per (sample, group).
//   pass 1 (grid = (nchunks, B)): per channel partial sums over the chunk's rows of dz and dz * xhat
//           (dz = dy * act'(z)) -> chan_part[b][chunk][C] (float2).  Their sum over (b, chunk) is (dbeta, dgamma).
//   pass 2 (grid = (nchunks, B)): per group A = sum_c gamma_c * sum dz / n, Bg = sum_c gamma_c * sum dz*xhat / n
//           (reduced from pass 1's partials in the prologue), dx = rstd * (dz * gamma - A - xhat * Bg).
// mean / rstd come from the forward statistics partials (ur_groupnorm_stats), reduced in fixed order.
// Thread mapping as in the forward kernels: a thread owns one 8-channel vector and strides over rows.  C <= 2048.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void gn_stat_reduce(const float* __restrict__ partial, int b, int nstat, int groups, int rows,
                                               int cpg, float eps, float2* stat, float2 (*red)[64], int t) {
    const int g = t & 63, j = t >> 6;
    float s = 0.f, ss = 0.f;
    if (g < groups) {
        const float2* src = reinterpret_cast<const float2*>(partial) + (int64_t)b * nstat * groups + g;
        for (int k = j; k < nstat; k += 4) {
            float2 a = src[(int64_t)k * groups];
            s += a.x;
            ss += a.y;
        }
    }
    red[j][g] = make_float2(s, ss);
    __syncthreads();
    if (t < groups) {
        float2 a0 = red[0][t], a1 = red[1][t], a2 = red[2][t], a3 = red[3][t];
        const float sum = (a0.x + a1.x) + (a2.x + a3.x), sq = (a0.y + a1.y) + (a2.y + a3.y);
        const float n = (float)rows * (float)cpg;
        const float mean = sum / n;
        const float var = fmaxf(sq / n - mean * mean, 0.f);
        stat[t] = make_float2(mean, rsqrtf(var + eps));
    }
    __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, int C,
                                                            int rows, int groups, int nstat, int nchunks,
                                                            const float* __restrict__ partial,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int silu,
                                                            float* __restrict__ chan_part) {
    __shared__ float2 stat[64];
    __shared__ float2 red[4][64];
    __shared__ float2 acc[256];  // one entry per thread = (row lane, vector column), folded per channel below
    const int nvec = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    gn_stat_reduce<T>(partial, b, nstat, groups, rows, cpg, eps, stat, red, t);
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int tpr = min(nvec, 256), rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    const T* xb = x + (int64_t)b * rows * C;
    const T* db = dy + (int64_t)b * rows * C;
    float2* outp = reinterpret_cast<float2*>(chan_part) + ((int64_t)b * nchunks + chunk) * C;
    for (int cvb = 0; cvb < nvec; cvb += tpr) {
        const int cv = cvb + cvl;
        float sd[8], sx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sd[i] = 0.f; sx[i] = 0.f; }
        if (rsub < rs && cv < nvec) {
            float gm[8], bt[8], mu[8], rsd[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cv * 8 + i;
                const float2 st = stat[c / cpg];
                gm[i] = gamma[c]; bt[i] = beta[c]; mu[i] = st.x; rsd[i] = st.y;
            }
            for (int r = rbeg + rsub; r < rend; r += rs) {
                float xv[8], dv[8];
                load8(xb + (int64_t)r * C + cv * 8, xv);
                load8(db + (int64_t)r * C + cv * 8, dv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (xv[i] - mu[i]) * rsd[i];
                    const float dz = silu ? dv[i] * silu_grad_f(xh * gm[i] + bt[i]) : dv[i];
                    sd[i] += dz;
                    sx[i] += dz * xh;
                }
            }
        }
        // fold the row lanes (fixed order) and write the per-channel partials of this chunk
        for (int i = 0; i < 8; ++i) {
            __syncthreads();
            acc[t] = make_float2(sd[i], sx[i]);
            __syncthreads();
            if (rsub == 0 && cv < nvec) {
                float a = 0.f, a2 = 0.f;
                for (int k = 0; k < rs; ++k) { a += acc[k * tpr + cvl].x; a2 += acc[k * tpr + cvl].y; }
                outp[cv * 8 + i] = make_float2(a, a2);
            }
        }
    }
}

// GroupNorm backward in ONE launch for the small maps (<= 1024 pixels per sample: the 32x32 / 16x16 / 8x8 levels): one
// 1024-thread workgroup per (sample, group) owns the group's [rows][cpg] strip -- statistics, the per-channel sums
// (sum dz, sum dz * xhat) and dx need no other workgroup.  Three sweeps over x and two over dy, all but the first from the
// XCD's L2 (workgroups of one sample sit on one XCD: neighbouring groups share the 128-byte lines of the strip, as in
// gn_fused_kernel).  Replaces ur_groupnorm_stats + gn_bwd_reduce + gn_bwd_fold + gn_bwd_apply (the forward of these maps is
// the one-launch GroupNorm, which keeps no statistics).  A thread keeps ONE piece of P channels for all its rows, so the
// per-channel sums stay in registers until one fixed-order fold through LDS.
constexpr int GNB_THREADS = 1024;

template <typename T, int P>
__device__ __forceinline__ void gnb_load(const T* p, float (&v)[P]) {
    if constexpr (P == 8) {
        load8(p, v);
    } else if constexpr (P == 4) {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
        T h[4];
        __builtin_memcpy(h, &raw, 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = to_f(h[i]);
    } else {
        const unsigned raw = *reinterpret_cast<const unsigned*>(p);
        T h[2];
        __builtin_memcpy(h, &raw, 4);
        v[0] = to_f(h[0]);
        v[1] = to_f(h[1]);
    }
}
template <typename T, int P>
__device__ __forceinline__ void gnb_store(T* p, const float (&v)[P]) {
    if constexpr (P == 8) {
        store8(p, v);
    } else {
        T h[P];
#pragma unroll
        for (int i = 0; i < P; ++i) h[i] = from_f<T>(v[i]);
        if constexpr (P == 4) { uint2 raw; __builtin_memcpy(&raw, h, 8); *reinterpret_cast<uint2*>(p) = raw; }
        else { unsigned raw; __builtin_memcpy(&raw, h, 4); *reinterpret_cast<unsigned*>(p) = raw; }
    }
}

template <typename T, int P>
__global__ void __launch_bounds__(GNB_THREADS) gn_bwd_fused_kernel(const T* __restrict__ x, const T* __restrict__ dy, int C, int rows,
                                                                   int groups, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, int silu,
                                                                   float* __restrict__ chan_sum, T* __restrict__ dx) {
    __shared__ float2 red[GNB_THREADS / 64];
    __shared__ float2 csum[128];                 // per channel of the group: (sum dz, sum dz * xhat)
    __shared__ float2 part[GNB_THREADS];         // one (sum dz, sum dz * xhat) per thread, one channel of the piece at a time
    __shared__ float2 gab;                       // (A, Bg) of the group
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int t = threadIdx.x, g = lid % gridDim.x, b = lid / gridDim.x;
    const int cpg = C / groups, ppr = cpg / P;   // pieces per row
    const int R = GNB_THREADS / ppr;             // rows per sweep step; threads >= R * ppr idle
    const int pc = t % ppr, r0 = t / ppr;
    const bool active = r0 < R;
    const int c0 = g * cpg + pc * P;
    const T* xb = x + (int64_t)b * rows * C + c0;
    const T* db = dy + (int64_t)b * rows * C + c0;
    T* ob = dx + (int64_t)b * rows * C + c0;

    // ---- sweep 1: statistics of the strip ----
    float s1 = 0.f, s2 = 0.f;
    if (active) {
        for (int r = r0; r < rows; r += R) {
            float v[P];
            gnb_load<T, P>(xb + (int64_t)r * C, v);
#pragma unroll
            for (int k = 0; k < P; ++k) { s1 += v[k]; s2 += v[k] * v[k]; }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((t & 63) == 0) red[t >> 6] = make_float2(s1, s2);
    __syncthreads();
    const float n = (float)rows * (float)cpg;
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int w = 0; w < GNB_THREADS / 64; ++w) { sum += red[w].x; sq += red[w].y; }  // fixed order
    const float mean = sum / n;
    const float rstd = rsqrtf(fmaxf(sq / n - mean * mean, 0.f) + eps);

    float gm[P], bt[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { gm[k] = gamma[c0 + k]; bt[k] = beta[c0 + k]; }

    // ---- sweep 2: per-channel sums of dz and dz * xhat ----
    float sd[P], sx[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { sd[k] = 0.f; sx[k] = 0.f; }
    if (active) {
        for (int r = r0; r < rows; r += R) {
            float xv[P], dv[P];
            gnb_load<T, P>(xb + (int64_t)r * C, xv);
            gnb_load<T, P>(db + (int64_t)r * C, dv);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float xh = (xv[k] - mean) * rstd;
                const float dz = silu ? dv[k] * silu_grad_f(xh * gm[k] + bt[k]) : dv[k];
                sd[k] += dz;
                sx[k] += dz * xh;
            }
        }
    }
    // fold the R row lanes of every channel in a fixed order: channel k of piece pc is summed by thread (pc, row lane 0)
    // (two levels: 16 threads per piece each add every 16th row lane, then one thread adds the 16 partial sums in order)
    __shared__ float2 part2[16 * 64];
    const int fq = t / ppr, fp = t - fq * ppr;   // t < 16 * ppr: second-level lane fq of piece fp (ppr <= 64)
#pragma unroll
    for (int k = 0; k < P; ++k) {
        __syncthreads();
        part[t] = make_float2(sd[k], sx[k]);
        __syncthreads();
        if (fq < 16) {
            float a = 0.f, a2 = 0.f;
            for (int j = fq; j < R; j += 16) { const float2 v = part[j * ppr + fp]; a += v.x; a2 += v.y; }
            part2[fq * 64 + fp] = make_float2(a, a2);
        }
        __syncthreads();
        if (t < ppr) {
            float a = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float2 v = part2[j * 64 + t]; a += v.x; a2 += v.y; }
            csum[t * P + k] = make_float2(a, a2);
        }
    }
    __syncthreads();
    if (t < cpg) reinterpret_cast<float2*>(chan_sum)[(int64_t)b * C + g * cpg + t] = csum[t];
    if (t == 0) {
        float a = 0.f, a2 = 0.f;
        for (int c = 0; c < cpg; ++c) { a += gamma[g * cpg + c] * csum[c].x; a2 += gamma[g * cpg + c] * csum[c].y; }
        gab = make_float2(a / n, a2 / n);
    }
    __syncthreads();
    const float A = gab.x, Bg = gab.y;

    // ---- sweep 3: dx ----
    if (active) {
        for (int r = r0; r < rows; r += R) {
            float xv[P], dv[P];
            gnb_load<T, P>(xb + (int64_t)r * C, xv);
            gnb_load<T, P>(db + (int64_t)r * C, dv);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float xh = (xv[k] - mean) * rstd;
                const float dz = silu ? dv[k] * silu_grad_f(xh * gm[k] + bt[k]) : dv[k];
                xv[k] = rstd * (dz * gm[k] - A - xh * Bg);
            }
            gnb_store<T, P>(ob + (int64_t)r * C, xv);
        }
    }
}

// chan_sum[b][c] = sum over the nred chunks of chan_part[b][chunk][c] (fixed order); one thread per (channel, component)
// A workgroup folds 32 columns: eight 32-lane groups each sum a contiguous eighth of the chunks (up to 512 dependent adds
// in one thread made this 16 us on a dozen workgroups), then the eight partial sums are added in group order.
__global__ void __launch_bounds__(256) gn_bwd_fold_kernel(const float* __restrict__ chan_part, int C2, int nred,
                                                          float* __restrict__ chan_sum) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el, b = blockIdx.y;
    const int len = (nred + 7) >> 3, k0 = q * len, k1 = min(nred, k0 + len);
    float a = 0.f;
    if (e < C2) {
        const float* src = chan_part + (int64_t)b * nred * C2 + e;
        for (int k = k0; k < k1; ++k) a += src[(int64_t)k * C2];
    }
    red[q][el] = a;
    __syncthreads();
    if (q == 0 && e < C2) {
        float t = red[0][el];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i][el];
        chan_sum[(int64_t)b * C2 + e] = t;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, int C,
                                                           int rows, int groups, int nstat, int nred, int nchunks,
                                                           const float* __restrict__ partial,
                                                           const float* __restrict__ chan_part,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int silu,
                                                           T* __restrict__ dx) {
    __shared__ float2 stat[64];
    __shared__ float2 red[4][64];
    __shared__ float2 gsum[64];  // per group: (A, Bg)
    __shared__ float2 gpart[256];
    const int nvec = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    gn_stat_reduce<T>(partial, b, nstat, groups, rows, cpg, eps, stat, red, t);
    {   // per group sums of gamma_c * (sum dz, sum dz*xhat): 4 threads per group over its channels and the nred chunks
        const int g = t >> 2, q = t & 3;
        float a = 0.f, a2 = 0.f;
        if (g < groups) {
            const float2* src = reinterpret_cast<const float2*>(chan_part) + (int64_t)b * nred * C;
            for (int e = q; e < cpg * nred; e += 4) {
                const int k = e / cpg, c = g * cpg + (e - k * cpg);
                const float2 v = src[(int64_t)k * C + c];
                a += gamma[c] * v.x;
                a2 += gamma[c] * v.y;
            }
        }
        gpart[t] = make_float2(a, a2);
        __syncthreads();
        if (t < groups) {
            const float2 p0 = gpart[4 * t], p1 = gpart[4 * t + 1], p2 = gpart[4 * t + 2], p3 = gpart[4 * t + 3];
            const float n = (float)rows * (float)cpg;
            gsum[t] = make_float2(((p0.x + p1.x) + (p2.x + p3.x)) / n, ((p0.y + p1.y) + (p2.y + p3.y)) / n);
        }
        __syncthreads();
    }
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int tpr = min(nvec, 256), rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    if (rsub >= rs) return;
    const T* xb = x + (int64_t)b * rows * C;
    const T* db = dy + (int64_t)b * rows * C;
    T* ob = dx + (int64_t)b * rows * C;
    for (int cv = cvl; cv < nvec; cv += tpr) {
        float gm[8], bt[8], mu[8], rsd[8], A[8], Bg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cv * 8 + i, g = c / cpg;
            const float2 st = stat[g], gs = gsum[g];
            gm[i] = gamma[c]; bt[i] = beta[c]; mu[i] = st.x; rsd[i] = st.y; A[i] = gs.x; Bg[i] = gs.y;
        }
        for (int r = rbeg + rsub; r < rend; r += rs) {
            float xv[8], dv[8];
            load8(xb + (int64_t)r * C + cv * 8, xv);
            load8(db + (int64_t)r * C + cv * 8, dv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (xv[i] - mu[i]) * rsd[i];
                const float dz = silu ? dv[i] * silu_grad_f(xh * gm[i] + bt[i]) : dv[i];
                xv[i] = rsd[i] * (dz * gm[i] - A[i] - xh * Bg[i]);
            }
            store8(ob + (int64_t)r * C + cv * 8, xv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row block of RPW rows; dx per row, and per-wave partial (dgamma, dbeta) rows
// part[wave][2][C] (fp32) that a column sum reduces.  C <= 2048 (MAXV vectors of 8 per lane).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                     const float* __restrict__ gamma, float eps, int rows, int C,
                                                     int rpw, T* __restrict__ dx, float* __restrict__ part,
                                                     const T* __restrict__ skip) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nvec = C >> 3;
    float dg[MAXV][8], dbt[MAXV][8];
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) { dg[k][i] = 0.f; dbt[k][i] = 0.f; }
    const int rbeg = wave * rpw, rend = min(rows, rbeg + rpw);
    for (int row = rbeg; row < rend; ++row) {
        float xv[MAXV][8], dv[MAXV][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
                load8(x + (int64_t)row * C + cv * 8, xv[k]);
                load8(dy + (int64_t)row * C + cv * 8, dv[k]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += xv[k][i];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k)
            if (lane + k * 64 < nvec) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = xv[k][i] - mean;
                    ss += d * d;
                }
            }
        const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
        float a = 0.f, b2 = 0.f;  // sum dxhat, sum dxhat * xhat
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
                const float4* g4 = reinterpret_cast<const float4*>(gamma + cv * 8);
                const float4 g0 = g4[0], g1 = g4[1];
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (xv[k][i] - mean) * rstd;
                    dg[k][i] += dv[k][i] * xh;
                    dbt[k][i] += dv[k][i];
                    const float dxh = dv[k][i] * g[i];
                    a += dxh;
                    b2 += dxh * xh;
                    xv[k][i] = xh;      // keep xhat
                    dv[k][i] = dxh;     // keep dxhat
                }
            }
        }
        a = wave_sum(a) / (float)C;
        b2 = wave_sum(b2) / (float)C;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rstd * (dv[k][i] - a - xv[k][i] * b2);
                if (skip) {  // the gradient that reaches x around the norm (residual path): added here, not by a launch of its own
                    float sk[8];
                    load8(skip + (int64_t)row * C + cv * 8, sk);
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] += sk[i];
                }
                store8(dx + (int64_t)row * C + cv * 8, o);
            }
        }
    }
    if (rbeg < rows) {
        float* pg = part + (int64_t)wave * 2 * C;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pg[cv * 8 + i] = dg[k][i];
                    pg[C + cv * 8 + i] = dbt[k][i];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention backward helpers (the P = softmax(Q K^T) matrix is recomputed and materialised per (batch, head); the
// five GEMMs of the gradient run on ur_igemm, z-batched over batch*heads).
//   split_heads: x [B][T][ld] (head h at columns off + h*d) -> out [B*H][Tp][dp], zero padded rows / columns
//   merge_heads: the inverse (accumulating nothing: plain store of the d real columns)
//   softmax_rows / softmax_bwd_rows: row-wise over the first ncols columns of [rows][ld], fp32 inside, in place
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) split_heads_kernel(const T* __restrict__ x, int64_t ld, int off, int B, int T_,
                                                          int H, int d, T* __restrict__ out, int Tp, int dp) {
    const int dv = dp >> 3;
    const int64_t total = (int64_t)B * H * Tp * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % dv) * 8;
        const int64_t r = i / dv;
        const int t = (int)(r % Tp);
        const int bh = (int)(r / Tp), b = bh / H, h = bh - b * H;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (t < T_ && c < d) load8(x + ((int64_t)b * T_ + t) * ld + off + h * d + c, v);
        store8(out + i * 8, v);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) merge_heads_kernel(const T* __restrict__ g, int Tp, int dp, int B, int T_, int H,
                                                          int d, T* __restrict__ out, int64_t ld, int off) {
    const int dv = d >> 3;
    const int64_t total = (int64_t)B * H * T_ * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % dv) * 8;
        const int64_t r = i / dv;
        const int t = (int)(r % T_);
        const int bh = (int)(r / T_), b = bh / H, h = bh - b * H;
        float v[8];
        load8(g + ((int64_t)bh * Tp + t) * dp + c, v);
        store8(out + ((int64_t)b * T_ + t) * ld + off + h * d + c, v);
    }
}

// The same copies for up to UR_HEADS_MAX tensors in ONE launch (blockIdx.y = tensor): q, k, v, o, dO -> per-head padded copies
// before the d = 40 flash backward, dq, dk, dv back afterwards (five + three launches per attention otherwise).
struct HeadsMultiArgs {
    ur_heads_desc t[UR_HEADS_MAX];
    int n, B, H, d, dp;
};
template <typename T>
__global__ void __launch_bounds__(256) split_heads_multi_kernel(const HeadsMultiArgs a) {
    const ur_heads_desc e = a.t[blockIdx.y];
    const T* x = reinterpret_cast<const T*>(e.tok);
    T* out = reinterpret_cast<T*>(e.heads);
    const int dv = a.dp >> 3;
    const int64_t total = (int64_t)a.B * a.H * e.Tp * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % dv) * 8;
        const int64_t r = i / dv;
        const int t = (int)(r % e.Tp);
        const int bh = (int)(r / e.Tp), b = bh / a.H, h = bh - b * a.H;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (t < e.T && c < a.d) load8(x + ((int64_t)b * e.T + t) * e.ld + e.off + h * a.d + c, v);
        store8(out + i * 8, v);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) merge_heads_multi_kernel(const HeadsMultiArgs a) {
    const ur_heads_desc e = a.t[blockIdx.y];
    const T* g = reinterpret_cast<const T*>(e.heads);
    T* out = reinterpret_cast<T*>(e.tok);
    const int dv = a.d >> 3;
    const int64_t total = (int64_t)a.B * a.H * e.T * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % dv) * 8;
        const int64_t r = i / dv;
        const int t = (int)(r % e.T);
        const int bh = (int)(r / e.T), b = bh / a.H, h = bh - b * a.H;
        float v[8];
        load8(g + ((int64_t)bh * e.Tp + t) * a.dp + c, v);
        store8(out + ((int64_t)b * e.T + t) * e.ld + e.off + h * a.d + c, v);
    }
}

// one wave per row; columns in 16-byte vectors; three passes over the row (max, sum, write) out of L2
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(T* __restrict__ s, int64_t ld, int64_t rows, int ncols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    T* p = s + row * ld;
    const int nv = (ncols + 7) >> 3;
    float mx = -3.0e38f;
    for (int cv = lane; cv < nv; cv += 64) {
        float v[8];
        load8(p + cv * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (cv * 8 + k < ncols) mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int cv = lane; cv < nv; cv += 64) {
        float v[8];
        load8(p + cv * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (cv * 8 + k < ncols) sum += __expf(v[k] - mx);
    }
    const float inv = 1.0f / wave_sum(sum);
    const int nvl = (int)(ld >> 3);
    for (int cv = lane; cv < nvl; cv += 64) {  // padding columns become exact zeros
        float v[8];
        load8(p + cv * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (cv * 8 + k < ncols) ? __expf(v[k] - mx) * inv : 0.f;
        store8(p + cv * 8, v);
    }
}

// dS = P * (dP - sum_k dP*P) * scale, written over dP
template <typename T>
__global__ void __launch_bounds__(256) softmax_bwd_rows_kernel(const T* __restrict__ pm, T* __restrict__ dp, int64_t ld,
                                                               int64_t rows, int ncols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* p = pm + row * ld;
    T* g = dp + row * ld;
    const int nv = (ncols + 7) >> 3;
    float dot = 0.f;
    for (int cv = lane; cv < nv; cv += 64) {
        float a[8], b[8];
        load8(p + cv * 8, a);
        load8(g + cv * 8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (cv * 8 + k < ncols) dot += a[k] * b[k];
    }
    dot = wave_sum(dot);
    const int nvl = (int)(ld >> 3);
    for (int cv = lane; cv < nvl; cv += 64) {
        float a[8], b[8];
        load8(p + cv * 8, a);
        load8(g + cv * 8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = (cv * 8 + k < ncols) ? a[k] * (b[k] - dot) * scale : 0.f;
        store8(g + cv * 8, b);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        load8(x + i * 8, a);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = silu_f(a[k]);
        store8(y + i * 8, a);
    }
}

// Resampling glue of the training path over NHWC [B][H][W][C] (C % 8 == 0):
//   mode 0  nearest 2x upsample        out[b][y][x] = in[b][y/2][x/2]                 (forward of Upsample2D)
//   mode 1  2x2 sum pooling            out[b][y][x] = sum in[b][2y+i][2x+j]            (its backward)
//   mode 2  zero insertion             out[b][2y][2x] = in[b][y][x], 0 elsewhere       (dgrad of the stride-2 conv)
template <typename T>
__global__ void __launch_bounds__(256) resample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Ho,
                                                         int Wo, int C, int mode) {
    const int cv = C >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8;
        const int64_t pix = i / cv;
        const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (mode == 0) {
            load8(in + (((int64_t)b * (Ho / 2) + y / 2) * (Wo / 2) + x / 2) * C + c, v);
        } else if (mode == 1) {
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    float t[8];
                    load8(in + (((int64_t)b * (Ho * 2) + 2 * y + dy) * (Wo * 2) + 2 * x + dx) * C + c, t);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += t[k];
                }
        } else if (((x | y) & 1) == 0) {
            load8(in + (((int64_t)b * (Ho / 2) + y / 2) * (Wo / 2) + x / 2) * C + c, v);
        }
        store8(out + i * 8, v);
    }
}

static inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace ur

using namespace ur;

// Conv weight master (fp32 [Co][Ci][3][3], the nn.Conv2d layout) <-> packed compute-dtype matrix [Co][9 * Cpad]
// (k = tap * Cpad + c, channels Ci .. Cpad zero): cast + repack in ONE pass each way.  A thread owns one (co, c): the
// nine taps are 36 contiguous bytes of the master and nine coalesced 2-byte accesses of the packed rows.
template <typename T>
__global__ void __launch_bounds__(256) pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Co,
                                                               int Ci, int Cpad) {
    // (an LDS-staged form with coalesced 4-byte loads of the master measured SLOWER, 16.1 vs 12.7 us per launch: the nine
    // loads of a thread already use every byte of the lines they touch, and the staging adds two barriers per 256 pairs;
    // the reverse direction, unpack_conv_weight_kernel below, does gain from it)
    const int64_t total = (int64_t)Co * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const int64_t co = i / Cpad;
        T* o = out + co * 9 * Cpad + c;
        if (c < Ci) {
            const float* src = w + (co * Ci + c) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) o[(int64_t)t * Cpad] = (T)src[t];
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) o[(int64_t)t * Cpad] = (T)0.0f;
        }
    }
}

// sum of a per-thread value over the workgroup in a fixed order (wave sums, then waves 0..3): the per-workgroup partial of
// the fused gradient-norm (ur_*_sumsq entry points)
__device__ __forceinline__ void block_partial_store(float v, float* dst) {
    __shared__ float wsum[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) dst[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// A workgroup moves runs of 256 (co, ci) pairs = 2304 consecutive output floats: the nine taps of a pair are read with
// coalesced 2-byte loads (lanes along ci), staged in LDS as [pair][tap], and leave as fully coalesced 4-byte stores over the
// contiguous run (a thread writing its own nine floats is a 36-byte lane stride: 18 partial lines per store instruction).
template <typename T>
__global__ void __launch_bounds__(256) unpack_conv_weight_kernel(const T* __restrict__ dwp, int64_t ld, float* __restrict__ out,
                                                                 int Co, int Ci, int Cpad, float* __restrict__ sumsq) {
    __shared__ float tile[256 * 9];
    const int64_t total = (int64_t)Co * Ci;
    float sq = 0.f;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < total; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        if (i < total) {
            const int c = (int)(i % Ci);
            const int64_t co = i / Ci;
            const T* src = dwp + co * ld + c;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float g = (float)src[(int64_t)t * Cpad];
                tile[threadIdx.x * 9 + t] = g;   // lane stride 9 words: odd, conflict-free
                sq = fmaf(g, g, sq);
            }
        }
        __syncthreads();
        const int64_t n = (total - i0 < 256 ? total - i0 : 256) * 9;
        float* o = out + i0 * 9;
        for (int e = threadIdx.x; e < n; e += 256) o[e] = tile[e];
        __syncthreads();
    }
    if (sumsq) block_partial_store(sq, sumsq);
}

#define UR_DISPATCH(dtype, CALL)                          \
    if ((dtype) == UR_DT_F16) { typedef f16 T; CALL; }    \
    else if ((dtype) == UR_DT_BF16) { typedef bf16 T; CALL; } \
    else return UR_E_BADARG;

static int last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_transpose2d(const void* src, int64_t ld_src, int64_t bs_src, void* dst, int64_t ld_dst, int64_t bs_dst,
                              int R, int C, int batch, int dtype, void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || batch <= 0 || (C & 7) || (ld_src & 7) || (ld_dst & 7) || (bs_src & 7) ||
        (bs_dst & 7) || ld_dst < ((R + 7) & ~7))
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((transpose2d_kernel<T>), grid, dim3(256), 0, s, (const T*)src, ld_src, bs_src,
                                          (T*)dst, ld_dst, bs_dst, R, C));
    return last_error();
}

extern "C" int ur_sizeof_transpose_desc(void) { return (int)sizeof(ur_transpose_desc); }

extern "C" int ur_transpose2d_multi(const ur_transpose_desc* descs, int n, int dtype, void* stream) {
    if (!descs || n <= 0 || n > UR_TRANSPOSE_MAX) return UR_E_BADARG;
    TransposeMultiArgs a;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        const ur_transpose_desc& d = descs[i];
        if (!d.src || !d.dst || d.R <= 0 || d.C <= 0 || d.batch <= 0 || (d.C & 7) || (d.ld_src & 7) || (d.ld_dst & 7) ||
            (d.bs_src & 7) || (d.bs_dst & 7) || d.ld_dst < ((d.R + 7) & ~7))
            return UR_E_BADARG;
        if (d.colsum && (d.batch != 1 || !d.colsum_ws || !d.colsum_cnt)) return UR_E_BADARG;
        if (d.rows_out && ((d.rows_out & 7) || d.rows_out < d.R || d.rows_out > ((d.R + 63) & ~63) || d.ld_dst < d.rows_out)) return UR_E_BADARG;
        a.d[i] = d;
        a.tile0[i] = (int)tiles;
        tiles += (int64_t)((d.C + 63) / 64) * ((d.R + 63) / 64) * d.batch;
        if (tiles > 0x7fffffff) return UR_E_BADARG;
    }
    a.tile0[n] = (int)tiles;
    a.n = n;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((transpose2d_multi_kernel<T>), dim3((unsigned)tiles), dim3(256), 0, s, a));
    return last_error();
}

extern "C" int ur_im2col3x3_t(const void* x, int B, int H, int W, int C, int stride, void* out, int64_t ld_out, int dtype,
                              void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (stride != 1 && stride != 2) || (ld_out & 7))
        return UR_E_BADARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    if (ld_out < (int64_t)B * Ho * Wo) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C + 63) / 64, (int)((ld_out + 63) / 64), 9);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((im2col3x3_t_kernel<T>), grid, dim3(256), 0, s, (const T*)x, B, H, W, C, Ho, Wo,
                                          stride, (T*)out, ld_out));
    return last_error();
}

__global__ void __launch_bounds__(256) pairsum_rows_kernel(const float* __restrict__ in, int B, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b2 = 0.f;
    for (int b = 0; b < B; ++b) {  // fixed order
        const float2 v = *reinterpret_cast<const float2*>(in + ((int64_t)b * C + c) * 2);
        a += v.x;
        b2 += v.y;
    }
    out[c] = a;
    out[C + c] = b2;
}

extern "C" int ur_pairsum_rows(const float* in, int B, int C, float* out, void* stream) {
    if (!in || !out || B <= 0 || C <= 0) return UR_E_BADARG;
    hipLaunchKernelGGL(pairsum_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, B, C, out);
    return last_error();
}

static int colsum_slices(int M, int N, int rpg) {
    // enough workgroups to fill the chip (~512) while a slice keeps >= 64 rows; <= 32 slices keeps the fold short
    const int groups = (M + rpg - 1) / rpg, cols = (N + 63) / 64;
    int s = 512 / (groups * cols);
    const int max_s = (rpg < M ? rpg : M) / 64;
    if (s > max_s) s = max_s;
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

extern "C" int64_t ur_colsum_workspace_floats(int M, int N, int rows_per_group) {
    if (M <= 0 || N <= 0) return 0;
    const int rpg = rows_per_group > 0 ? rows_per_group : M;
    const int sl = colsum_slices(M, N, rpg);
    return sl > 1 ? (int64_t)((M + rpg - 1) / rpg) * sl * N : 0;
}

extern "C" int ur_colsum_counters(int M, int N, int rows_per_group) {
    if (M <= 0 || N <= 0) return 0;
    const int rpg = rows_per_group > 0 ? rows_per_group : M;
    return ((M + rpg - 1) / rpg) * ((N + 63) / 64);
}

// counters: NULL = partial sums + a second (fold) launch; else ur_colsum_counters() zero-initialised device counters that
// the launch leaves zero again: ONE launch, the last workgroup of every column block folds (same order, same bits)
extern "C" int ur_colsum_fused(const void* x, int64_t ldx, int M, int N, int rows_per_group, float* out, float* workspace,
                               unsigned* counters, int dtype, void* stream) {
    if (!x || !out || M <= 0 || N <= 0 || (N & 7) || (ldx & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rpg = rows_per_group > 0 ? rows_per_group : M;
    const int groups = (M + rpg - 1) / rpg;
    const int sl = colsum_slices(M, N, rpg);
    if (sl > 1 && !workspace) return UR_E_BADARG;
    float* dst = sl > 1 ? workspace : out;
    unsigned* cnt = sl > 1 ? counters : nullptr;
    dim3 grid((N + 63) / 64, sl, groups);
    if (dtype == UR_DT_F32) {
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)x, ldx, M, N, rpg, sl, dst, out, cnt);
    } else {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, s, (const T*)x, ldx, M, N, rpg, sl, dst, out, cnt));
    }
    if (sl > 1 && !cnt) hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 63) / 64, groups), dim3(256), 0, s, workspace, N, sl, out);
    return last_error();
}

extern "C" int ur_colsum(const void* x, int64_t ldx, int M, int N, int rows_per_group, float* out, float* workspace,
                         int dtype, void* stream) {
    return ur_colsum_fused(x, ldx, M, N, rows_per_group, out, workspace, nullptr, dtype, stream);
}

extern "C" int ur_silu_backward(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
    if (!x || !dy || !dx || n <= 0 || (n & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((silu_bwd_kernel<T>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const T*)x,
                                          (const T*)dy, (T*)dx, n / 8));
    return last_error();
}

extern "C" int ur_geglu_forward(const void* h, void* y, int64_t M, int D, int dtype, void* stream) {
    if (!h || !y || M <= 0 || D <= 0 || (D & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((geglu_fwd_kernel<T>), dim3(grid_for(M * (D / 8))), dim3(256), 0, s, (const T*)h,
                                          (T*)y, M, D));
    return last_error();
}

extern "C" int ur_geglu_backward(const void* h, const void* dy, void* dh, int64_t M, int D, int dtype, void* stream) {
    if (!h || !dy || !dh || M <= 0 || D <= 0 || (D & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((geglu_bwd_kernel<T>), dim3(grid_for(M * (D / 8))), dim3(256), 0, s, (const T*)h,
                                          (const T*)dy, (T*)dh, M, D));
    return last_error();
}

extern "C" int ur_groupnorm_backward(const void* x, const void* dy, int C, int B, int rows, int groups, int nstat,
                                     const float* partial, const float* gamma, const float* beta, float eps, int silu,
                                     int nred, float* chan_part, float* chan_sum, int nchunks, void* dx, int dtype,
                                     void* stream) {
    if (!x || !dy || !partial || !gamma || !beta || !chan_part || !chan_sum || !dx) return UR_E_BADARG;
    if (C <= 0 || (C & 7) || C > 4096 || B <= 0 || rows <= 0 || groups <= 0 || groups > 64 || (C % groups) || nstat <= 0 ||
        nchunks <= 0 || nchunks > 65535 || nred <= 0 || nred > 65535)
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, {
        hipLaunchKernelGGL((gn_bwd_reduce_kernel<T>), dim3(nred, B), dim3(256), 0, s, (const T*)x, (const T*)dy, C, rows,
                           groups, nstat, nred, partial, gamma, beta, eps, silu, chan_part);
        hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3((2 * C + 31) / 32, B), dim3(256), 0, s, chan_part, 2 * C, nred, chan_sum);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(nchunks, B), dim3(256), 0, s, (const T*)x, (const T*)dy, C, rows,
                           groups, nstat, 1, nchunks, partial, chan_sum, gamma, beta, eps, silu, (T*)dx);
    });
    return last_error();
}

extern "C" int ur_groupnorm_backward_fused(const void* x, const void* dy, int C, int B, int rows, int groups, const float* gamma,
                                           const float* beta, float eps, int silu, float* chan_sum, void* dx, int dtype,
                                           void* stream) {
    if (!x || !dy || !gamma || !beta || !chan_sum || !dx || C <= 0 || B <= 0 || rows <= 0 || groups <= 0 || C % groups)
        return UR_E_BADARG;
    const int cpg = C / groups;
    if (cpg > 128 || (cpg & 1) || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) || (C & 7)) return UR_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(groups, B);
#define UR_GNB(PP)                                                                                                        \
    UR_DISPATCH(dtype, hipLaunchKernelGGL((gn_bwd_fused_kernel<T, PP>), grid, dim3(GNB_THREADS), 0, s, (const T*)x, (const T*)dy, C, \
                                          rows, groups, gamma, beta, eps, silu, chan_sum, (T*)dx))
    if (cpg % 8 == 0) { UR_GNB(8); }
    else if (cpg % 4 == 0) { UR_GNB(4); }
    else { UR_GNB(2); }
#undef UR_GNB
    return last_error();
}

extern "C" int ur_layernorm_backward_skip(const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                                          int rows_per_wave, void* dx, float* part, const void* skip, int dtype, void* stream) {
    if (!x || !dy || !gamma || !dx || !part || rows <= 0 || C <= 0 || (C & 7) || C > 2048 || rows_per_wave <= 0)
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int waves = (rows + rows_per_wave - 1) / rows_per_wave;
    dim3 grid((waves + 3) / 4);
    if (C <= 512) {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, gamma,
                                              eps, rows, C, rows_per_wave, (T*)dx, part, (const T*)skip));
    } else if (C <= 1024) {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, 2>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, gamma,
                                              eps, rows, C, rows_per_wave, (T*)dx, part, (const T*)skip));
    } else {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, gamma,
                                              eps, rows, C, rows_per_wave, (T*)dx, part, (const T*)skip));
    }
    return last_error();
}

extern "C" int ur_layernorm_backward(const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                                     int rows_per_wave, void* dx, float* part, int dtype, void* stream) {
    return ur_layernorm_backward_skip(x, dy, gamma, eps, rows, C, rows_per_wave, dx, part, nullptr, dtype, stream);
}

extern "C" int ur_split_heads(const void* x, int64_t ld, int off, int B, int T_, int H, int d, void* out, int Tp, int dp,
                              int dtype, void* stream) {
    if (!x || !out || B <= 0 || T_ <= 0 || H <= 0 || d <= 0 || (d & 7) || (dp & 7) || dp < d || Tp < T_ || (ld & 7) || (off & 7))
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * Tp * (dp / 8);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((split_heads_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, ld, off,
                                          B, T_, H, d, (T*)out, Tp, dp));
    return last_error();
}

extern "C" int ur_merge_heads(const void* g, int Tp, int dp, int B, int T_, int H, int d, void* out, int64_t ld, int off,
                              int dtype, void* stream) {
    if (!g || !out || B <= 0 || T_ <= 0 || H <= 0 || d <= 0 || (d & 7) || (dp & 7) || dp < d || Tp < T_ || (ld & 7) || (off & 7))
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * T_ * (d / 8);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((merge_heads_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)g, Tp, dp, B,
                                          T_, H, d, (T*)out, ld, off));
    return last_error();
}

// Column sums of up to UR_COLSUM_MULTI_MAX fp32 matrices [M][N] in ONE launch (the parameter gradients of every LayerNorm /
// GroupNorm of a network, deferred to the end of its backward: backward.NormSums): a workgroup owns 32 columns of one matrix,
// eight row lanes per column, folded in a fixed order.  pair: the columns are (channel, component) pairs [C][2] and the
// result is planar, out[k * C + c] (ur_pairsum_rows).
struct ColsumMultiArgs {
    ur_colsum_item t[UR_COLSUM_MULTI_MAX];
    int blk0[UR_COLSUM_MULTI_MAX + 1];
    int n;
};
__global__ void __launch_bounds__(256) colsum_multi_kernel(const ColsumMultiArgs a) {
    __shared__ float red[8][32];
    int lo = 0, hi = a.n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const ur_colsum_item e = a.t[lo];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;   // 32 columns x 8 row lanes: 128-byte row segments
    const int col = ((int)blockIdx.x - a.blk0[lo]) * 32 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (col < e.N) {
        int r = q;
        for (; r + 8 < e.M; r += 16) { s0 += e.in[(int64_t)r * e.N + col]; s1 += e.in[(int64_t)(r + 8) * e.N + col]; }
        if (r < e.M) s0 += e.in[(int64_t)r * e.N + col];
    }
    red[q][cl] = s0 + s1;
    __syncthreads();
    if (q == 0 && col < e.N) {
        float v = red[0][cl];
#pragma unroll
        for (int i = 1; i < 8; ++i) v += red[i][cl];
        e.out[e.pair ? (col & 1) * (e.N >> 1) + (col >> 1) : col] = v;
    }
}
extern "C" int ur_colsum_multi(const ur_colsum_item* items, int n, void* stream) {
    if (!items || n <= 0 || n > UR_COLSUM_MULTI_MAX) return UR_E_BADARG;
    ColsumMultiArgs a;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].in || !items[i].out || items[i].M <= 0 || items[i].N <= 0 || (items[i].pair && (items[i].N & 1))) return UR_E_BADARG;
        a.t[i] = items[i];
        a.blk0[i] = blocks;
        blocks += (items[i].N + 31) / 32;
    }
    a.blk0[n] = blocks;
    a.n = n;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return last_error();
}
extern "C" int ur_sizeof_colsum_item(void) { return (int)sizeof(ur_colsum_item); }

static int heads_multi(const ur_heads_desc* descs, int n, int B, int H, int d, int dp, int dtype, void* stream, bool split) {
    if (!descs || n <= 0 || n > UR_HEADS_MAX || B <= 0 || H <= 0 || d <= 0 || (d & 7) || (dp & 7) || dp < d) return UR_E_BADARG;
    HeadsMultiArgs a;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        const ur_heads_desc& e = descs[i];
        if (!e.tok || !e.heads || e.T <= 0 || e.Tp < e.T || (e.ld & 7) || (e.off & 7)) return UR_E_BADARG;
        a.t[i] = e;
        const int64_t total = (int64_t)B * H * (split ? (int64_t)e.Tp * (dp / 8) : (int64_t)e.T * (d / 8));
        if (total > most) most = total;
    }
    a.n = n; a.B = B; a.H = H; a.d = d; a.dp = dp;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(most), n);
    if (split) { UR_DISPATCH(dtype, hipLaunchKernelGGL((split_heads_multi_kernel<T>), grid, dim3(256), 0, s, a)); }
    else { UR_DISPATCH(dtype, hipLaunchKernelGGL((merge_heads_multi_kernel<T>), grid, dim3(256), 0, s, a)); }
    return last_error();
}
extern "C" int ur_split_heads_multi(const ur_heads_desc* descs, int n, int B, int H, int d, int dp, int dtype, void* stream) {
    return heads_multi(descs, n, B, H, d, dp, dtype, stream, true);
}
extern "C" int ur_merge_heads_multi(const ur_heads_desc* descs, int n, int B, int H, int d, int dp, int dtype, void* stream) {
    return heads_multi(descs, n, B, H, d, dp, dtype, stream, false);
}
extern "C" int ur_sizeof_heads_desc(void) { return (int)sizeof(ur_heads_desc); }

extern "C" int ur_softmax_rows(void* s_, int64_t ld, int64_t rows, int ncols, int dtype, void* stream) {
    if (!s_ || rows <= 0 || ncols <= 0 || (ld & 7) || ld < ncols) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (T*)s_, ld,
                                          rows, ncols));
    return last_error();
}

extern "C" int ur_softmax_backward_rows(const void* p, void* dp, int64_t ld, int64_t rows, int ncols, float scale, int dtype,
                                        void* stream) {
    if (!p || !dp || rows <= 0 || ncols <= 0 || (ld & 7) || ld < ncols) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((softmax_bwd_rows_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                                          (const T*)p, (T*)dp, ld, rows, ncols, scale));
    return last_error();
}

extern "C" int ur_silu_forward(const void* x, void* y, int64_t n, int dtype, void* stream) {
    if (!x || !y || n <= 0 || (n & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((silu_fwd_kernel<T>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const T*)x, (T*)y, n / 8));
    return last_error();
}

extern "C" int ur_resample2x(const void* in, void* out, int B, int Hout, int Wout, int C, int mode, int dtype, void* stream) {
    if (!in || !out || B <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || (C & 7) || mode < 0 || mode > 2) return UR_E_BADARG;
    if (mode != 1 && ((Hout | Wout) & 1)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * Hout * Wout * (C / 8);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((resample2x_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)in, (T*)out, B,
                                          Hout, Wout, C, mode));
    return last_error();
}

extern "C" int ur_pack_conv_weight(const float* w, void* out, int Co, int Ci, int Cpad, int dtype, void* stream) {
    if (!w || !out || Co <= 0 || Ci <= 0 || Cpad < Ci) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)Co * Cpad;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((pack_conv_weight_kernel<T>), dim3(grid), dim3(256), 0, s, w, (T*)out, Co, Ci, Cpad));
    return last_error();
}

extern "C" int ur_unpack_conv_weight_grad_blocks(int Co, int Ci) {
    const int64_t total = (int64_t)Co * Ci;
    return (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
}

extern "C" int ur_unpack_conv_weight_grad_sumsq(const void* dwp, int64_t ld, float* out, int Co, int Ci, int Cpad, float* sumsq,
                                                int dtype, void* stream) {
    if (!dwp || !out || Co <= 0 || Ci <= 0 || Cpad < Ci || ld < 9 * (int64_t)Cpad) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = ur_unpack_conv_weight_grad_blocks(Co, Ci);
    UR_DISPATCH(dtype, hipLaunchKernelGGL((unpack_conv_weight_kernel<T>), dim3(grid), dim3(256), 0, s, (const T*)dwp, ld, out,
                                          Co, Ci, Cpad, sumsq));
    return last_error();
}

extern "C" int ur_unpack_conv_weight_grad(const void* dwp, int64_t ld, float* out, int Co, int Ci, int Cpad, int dtype,
                                          void* stream) {
    return ur_unpack_conv_weight_grad_sumsq(dwp, ld, out, Co, Ci, Cpad, nullptr, dtype, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// AdamW over up to UR_ADAMW_MAX_TENSORS tensors per launch.  One workgroup per 16384-element chunk; the chunk -> tensor
// map is a prefix-sum table in the kernel arguments (wave-uniform binary search).  HBM-bound: 16 B read + 12 B written
// per element.
// ---------------------------------------------------------------------------------------------------------------
struct AdamwArgs {
    ur_adamw_tensor t[UR_ADAMW_MAX_TENSORS];
    int chunk0[UR_ADAMW_MAX_TENSORS + 1];
    int n;
    float lr, beta1, beta2, eps, wd;
    const float *step, *grad_scale, *found_inf;
    const float* hyper;  // NULL, or device {lr, weight_decay}: read at run time instead of the by-value arguments
};
constexpr int ADAMW_CHUNK = 16384;

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float ginv, float decay, float b1, float b2,
                                          float step_size, float rbc2, float eps) {
    g /= ginv;  // ginv = grad_scale (1 when absent): a division like torch's fused kernel
    p *= decay;
    m = m + (1.0f - b1) * (g - m);
    v = b2 * v + (1.0f - b2) * g * g;
    p -= step_size * m / (sqrtf(v) * rbc2 + eps);
}

__global__ void __launch_bounds__(256) adamw_multi_kernel(const AdamwArgs a) {
    if (a.found_inf && *a.found_inf != 0.f) return;
    int lo = 0, hi = a.n;  // last tensor with chunk0 <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.chunk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const ur_adamw_tensor t = a.t[lo];
    const int64_t beg = (int64_t)((int)blockIdx.x - a.chunk0[lo]) * ADAMW_CHUNK;
    const int64_t end = beg + ADAMW_CHUNK < t.n ? beg + ADAMW_CHUNK : t.n;
    const float step = *a.step;
    // 1 - beta^step in double: beta2 = 0.999 at step 1 leaves 1e-3, which fp32 v_log / v_exp resolve to 1e-4 relative only
    const float bc1 = (float)(1.0 - pow((double)a.beta1, (double)step));
    const float bc2 = (float)(1.0 - pow((double)a.beta2, (double)step));
    // lr / weight decay from device memory when given: a captured graph then follows an lr scheduler without re-capture
    const float lr = a.hyper ? a.hyper[0] : a.lr, wd = a.hyper ? a.hyper[1] : a.wd;
    const float step_size = lr / bc1, rbc2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * wd;
    const float ginv = a.grad_scale ? *a.grad_scale : 1.0f;
    const bool vec = ((((uintptr_t)t.p) | ((uintptr_t)t.g) | ((uintptr_t)t.m) | ((uintptr_t)t.v)) & 15) == 0;
    if (vec) {
        const int64_t end4 = beg + ((end - beg) & ~(int64_t)3);
        // every byte is touched once per step and 48 GB pass before it is touched again: non-temporal both ways
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int64_t i = beg + 4 * threadIdx.x; i < end4; i += 1024) {
            f4 p = __builtin_nontemporal_load(reinterpret_cast<const f4*>(t.p + i));
            f4 m = __builtin_nontemporal_load(reinterpret_cast<const f4*>(t.m + i));
            f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(t.v + i));
            const f4 g = __builtin_nontemporal_load(reinterpret_cast<const f4*>(t.g + i));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adamw_one(pk, g[k], mk, vk, ginv, decay, a.beta1, a.beta2, step_size, rbc2, a.eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
            __builtin_nontemporal_store(p, reinterpret_cast<f4*>(t.p + i));
            __builtin_nontemporal_store(m, reinterpret_cast<f4*>(t.m + i));
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(t.v + i));
        }
        for (int64_t i = end4 + threadIdx.x; i < end; i += 256)
            adamw_one(t.p[i], t.g[i], t.m[i], t.v[i], ginv, decay, a.beta1, a.beta2, step_size, rbc2, a.eps);
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += 256)
            adamw_one(t.p[i], t.g[i], t.m[i], t.v[i], ginv, decay, a.beta1, a.beta2, step_size, rbc2, a.eps);
    }
}

extern "C" int ur_adamw_multi(const ur_adamw_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                              float weight_decay, const float* step, const float* grad_scale, const float* found_inf,
                              const float* hyper, void* stream) {
    if (!tensors || n_tensors <= 0 || n_tensors > UR_ADAMW_MAX_TENSORS || !step || !(beta1 >= 0.f && beta1 < 1.f) ||
        !(beta2 >= 0.f && beta2 < 1.f))  // the same range torch.optim.AdamW (and optim.FusedAdamW) accept
        return UR_E_BADARG;
    AdamwArgs a;
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].p || !tensors[i].g || !tensors[i].m || !tensors[i].v || tensors[i].n <= 0) return UR_E_BADARG;
        a.t[i] = tensors[i];
        a.chunk0[i] = (int)chunks;
        chunks += (tensors[i].n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
        if (chunks > 0x7fffffff) return UR_E_BADARG;
    }
    a.chunk0[n_tensors] = (int)chunks;
    a.n = n_tensors;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.step = step; a.grad_scale = grad_scale; a.found_inf = found_inf; a.hyper = hyper;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 <-> compute dtype casts of up to UR_CAST_MAX_TENSORS tensors per launch (chunk table in the kernel arguments)
// ---------------------------------------------------------------------------------------------------------------
struct CastArgs {
    ur_cast_tensor t[UR_CAST_MAX_TENSORS];
    int chunk0[UR_CAST_MAX_TENSORS + 1];
    int n;
    float* sumsq;  // to_f32 only: per-workgroup sum of squares of the values written (NULL: none)
};
constexpr int CAST_CHUNK = 8192;

template <typename T, bool TO_F32>
__global__ void __launch_bounds__(256) cast_multi_kernel(const CastArgs a) {
    int lo = 0, hi = a.n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.chunk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const ur_cast_tensor t = a.t[lo];
    const int64_t beg = (int64_t)((int)blockIdx.x - a.chunk0[lo]) * CAST_CHUNK;
    const int64_t end = beg + CAST_CHUNK < t.n ? beg + CAST_CHUNK : t.n;
    const bool vec = ((((uintptr_t)t.src) | ((uintptr_t)t.dst)) & 15) == 0;
    const int64_t end8 = vec ? beg + ((end - beg) & ~(int64_t)7) : beg;
    if constexpr (TO_F32) {
        const T* src = reinterpret_cast<const T*>(t.src);
        float* dst = reinterpret_cast<float*>(t.dst);
        float sq = 0.f;
        for (int64_t i = beg + 8 * threadIdx.x; i < end8; i += 2048) {
            float v[8];
            load8(src + i, v);
            *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
#pragma unroll
            for (int k = 0; k < 8; ++k) sq = fmaf(v[k], v[k], sq);
        }
        for (int64_t i = end8 + threadIdx.x; i < end; i += 256) {
            const float g = (float)src[i];
            dst[i] = g;
            sq = fmaf(g, g, sq);
        }
        if (a.sumsq) block_partial_store(sq, a.sumsq);
    } else {
        const float* src = reinterpret_cast<const float*>(t.src);
        T* dst = reinterpret_cast<T*>(t.dst);
        for (int64_t i = beg + 8 * threadIdx.x; i < end8; i += 2048) {
            const float4 x = *reinterpret_cast<const float4*>(src + i), y = *reinterpret_cast<const float4*>(src + i + 4);
            const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
            store8(dst + i, v);
        }
        for (int64_t i = end8 + threadIdx.x; i < end; i += 256) dst[i] = (T)src[i];
    }
}

extern "C" int ur_cast_multi_sumsq(const ur_cast_tensor* tensors, int n_tensors, int to_f32, int dtype, float* sumsq, void* stream);
extern "C" int ur_cast_multi(const ur_cast_tensor* tensors, int n_tensors, int to_f32, int dtype, void* stream) {
    return ur_cast_multi_sumsq(tensors, n_tensors, to_f32, dtype, nullptr, stream);
}

extern "C" int64_t ur_cast_multi_blocks(const ur_cast_tensor* tensors, int n_tensors) {
    int64_t chunks = 0;
    for (int i = 0; tensors && i < n_tensors; ++i) chunks += (tensors[i].n + CAST_CHUNK - 1) / CAST_CHUNK;
    return chunks;
}

extern "C" int ur_cast_multi_sumsq(const ur_cast_tensor* tensors, int n_tensors, int to_f32, int dtype, float* sumsq, void* stream) {
    if (!tensors || n_tensors <= 0 || n_tensors > UR_CAST_MAX_TENSORS || (sumsq && !to_f32)) return UR_E_BADARG;
    CastArgs a;
    a.sumsq = sumsq;
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].src || !tensors[i].dst || tensors[i].n <= 0) return UR_E_BADARG;
        a.t[i] = tensors[i];
        a.chunk0[i] = (int)chunks;
        chunks += (tensors[i].n + CAST_CHUNK - 1) / CAST_CHUNK;
        if (chunks > 0x7fffffff) return UR_E_BADARG;
    }
    a.chunk0[n_tensors] = (int)chunks;
    a.n = n_tensors;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (to_f32) {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((cast_multi_kernel<T, true>), dim3((unsigned)chunks), dim3(256), 0, s, a));
    } else {
        UR_DISPATCH(dtype, hipLaunchKernelGGL((cast_multi_kernel<T, false>), dim3((unsigned)chunks), dim3(256), 0, s, a));
    }
    return last_error();
}
