// GroupNorm(+SiLU) over NHWC (optionally over the concat of two sources) and LayerNorm, gfx950.
// Both are HBM-bound: 16-byte vector accesses, fp32 statistics, wave-shuffle / fixed-order LDS
// reductions (deterministic: no floating-point atomics anywhere).
#include "ur_common.h"
#include "../../include/ur_kernels.h"

#include <cstdlib>

// The one-launch GroupNorm puts the groups of one sample on ONE XCD (neighbouring groups share cache lines: 2.5x on large maps,
// profiles/r04_gnf_xcd_ab.txt) and the row-chunked kernels use the XCD-contiguous mapping; both were environment toggles
// during their A/B runs and are fixed now.
static constexpr int gnf_xcd() { return 1; }
// round 6: the one-launch GroupNorm keeps small strips in registers (gn_resident_kernel).  Bit 1 of ur_groupnorm_fused's `silu`
// argument (UR_GN_TWO_SWEEP) asks for the two-sweep kernel instead: A/B runs and the bitwise test of the two against each other.
static constexpr int norm_xcd() { return 1; }

namespace ur {

// ------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid = (nchunks, B); a workgroup sums rows [chunk*rpc, (chunk+1)*rpc) of
// sample b over all channels and writes partial[b][chunk][g] = (sum, sumsq).
// Thread mapping: a thread owns one 8-channel vector (fixed) and strides over rows, so consecutive
// threads read consecutive 16-B pieces of a row and, because NHWC rows are contiguous, of the next row.
// Eight row loads are always in flight per thread (rows past the chunk are clamped and weighted 0), a
// thread folds its 8 channels into the (at most two) groups they belong to, and every group is then
// summed by four threads over a few LDS entries -- all in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                       const lo_t<T>* __restrict__ x0_lo, const lo_t<T>* __restrict__ x1_lo, int c0,
                                                       int c1, int rows, int groups, int nchunks,
                                                       float* __restrict__ partial, int xcd) {
    __shared__ float4 tpart[256];  // per thread: (sum, sumsq) of its channels in group gA, and in group gA + 1
    __shared__ float2 chs[2048];   // cpg < 8 only (tiny test configurations): per-(row-subset, channel) sums
    const int C = c0 + c1, nvec = C >> 3, cpg = C / groups;
    // xcd: (sample, chunk) -> XCD so that every XCD owns ONE contiguous range of rows of the (stream-major stacked)
    // activation tensor, the same partition ur_igemm's tile remap gives its m-tiles: a consumer GEMM then finds the
    // rows it reads in the L2 of the XCD that wrote them (and this kernel the rows its producer wrote)
    const int lid_ = xcd ? xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y) : blockIdx.x + gridDim.x * blockIdx.y;
    const int b = lid_ / gridDim.x, chunk = lid_ - b * gridDim.x;
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int t = threadIdx.x;
    const int tpr = min(nvec, 256);
    const int rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    const int nv0 = c0 >> 3;
    float gs = 0.f, gss = 0.f;  // threads 4g .. 4g+3 accumulate group g over the passes
    for (int cvb = 0; cvb < nvec; cvb += tpr) {
        const int cv = cvb + cvl;
        float s[8], ss[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; }
        const bool active = rsub < rs && cv < nvec && rbeg < rend;
        if (active) {
            const T* base;
            const lo_t<T>* lbase;  // low part of a (hi, lo) residual-stream input, or null
            int64_t ld;
            int co;
            if (cv < nv0) { base = x0 + (int64_t)b * rows * c0; lbase = x0_lo ? x0_lo + (int64_t)b * rows * c0 : nullptr; ld = c0; co = cv * 8; }
            else { base = x1 + (int64_t)b * rows * c1; lbase = x1_lo ? x1_lo + (int64_t)b * rows * c1 : nullptr; ld = c1; co = (cv - nv0) * 8; }
            for (int r = rbeg + rsub; r < rend; r += 8 * rs) {  // 8 independent 16-byte loads in flight per thread
                typename Vec8<T>::type raw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    raw[u] = *reinterpret_cast<const typename Vec8<T>::type*>(base + (int64_t)min(r + u * rs, rend - 1) * ld + co);
                if (lbase) {
                    float rlo[8][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) load_lo<8>(lbase + (int64_t)min(r + u * rs, rend - 1) * ld + co, rlo[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float w = (r + u * rs < rend) ? 1.f : 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float v = ((float)raw[u][i] + rlo[u][i]) * w;
                            s[i] += v;
                            ss[i] += v * v;
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float w = (r + u * rs < rend) ? 1.f : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = (float)raw[u][i] * w;
                        s[i] += v;
                        ss[i] += v * v;
                    }
                }
            }
        }
        if (cpg < 8) {
            // narrow groups (a vector spans more than two): per-channel sums through LDS, one thread per group
            __syncthreads();
            if (rsub < rs) {
#pragma unroll
                for (int i = 0; i < 8; ++i) chs[rsub * (tpr * 8) + cvl * 8 + i] = active ? make_float2(s[i], ss[i]) : make_float2(0.f, 0.f);
            }
            __syncthreads();
            if ((t & 3) == 0 && (t >> 2) < groups) {
                const int g = t >> 2;
                const int pb = cvb * 8, pe = min(C, pb + tpr * 8);
                const int cb = max(pb, g * cpg), ce = min(pe, (g + 1) * cpg);
                for (int c = cb; c < ce; ++c)
                    for (int k = 0; k < rs; ++k) {
                        const float2 a = chs[k * (tpr * 8) + (c - pb)];
                        gs += a.x;
                        gss += a.y;
                    }
            }
            continue;
        }
        // fold the 8 channels into their (at most two: cpg >= 8) groups
        const int ca = cv * 8, ga = ca / cpg, split = (ga + 1) * cpg - ca;  // channels i < split belong to ga
        float4 tp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < split) { tp.x += s[i]; tp.y += ss[i]; }
            else { tp.z += s[i]; tp.w += ss[i]; }
        }
        __syncthreads();  // previous pass fully consumed
        tpart[t] = active ? tp : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        {   // group g = t / 4 (needs groups <= 64), quarter q = t & 3 of its contributing (vector, row-subset) entries
            const int g = t >> 2, q = t & 3;
            float a = 0.f, a2 = 0.f;
            if (g < groups) {
                // vectors of THIS pass touching group g: cv in [first, last]
                const int first = max(cvb, (g * cpg) >> 3), last = min(min(cvb + tpr, nvec) - 1, ((g + 1) * cpg - 1) >> 3);
                const int nv = last - first + 1;
                for (int e = q; e < nv * rs; e += 4) {
                    const int k = e / nv, cvv = first + (e - k * nv);
                    const float4 v = tpart[k * tpr + (cvv - cvb)];
                    const int gav = (cvv * 8) / cpg;
                    if (gav == g) { a += v.x; a2 += v.y; }
                    else if (gav + 1 == g) { a += v.z; a2 += v.w; }
                }
            }
            gs += a;
            gss += a2;
        }
    }
    // combine the four quarter sums of a group (lanes 4g .. 4g+3 of one wave), fixed order
    gs += __shfl_xor(gs, 1, 64);
    gss += __shfl_xor(gss, 1, 64);
    gs += __shfl_xor(gs, 2, 64);
    gss += __shfl_xor(gss, 2, 64);
    if ((t & 3) == 0 && (t >> 2) < groups) {
        float2* dst = reinterpret_cast<float2*>(partial) + ((int64_t)b * nchunks + chunk) * groups + (t >> 2);
        *dst = make_float2(gs, gss);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                       const lo_t<T>* __restrict__ x0_lo, const lo_t<T>* __restrict__ x1_lo, int c0,
                                                       int c1, int rows, int groups, int nstat, int nchunks,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu,
                                                       int bper, int pstride, T* __restrict__ out, int xcd) {
    __shared__ float2 stat[64];  // (mean, rstd) per group
    __shared__ float2 red[4][64];
    const int C = c0 + c1, nvec = C >> 3, cpg = C / groups;
    // xcd: (sample, chunk) -> XCD so that every XCD owns ONE contiguous range of rows of the (stream-major stacked)
    // activation tensor, the same partition ur_igemm's tile remap gives its m-tiles: a consumer GEMM then finds the
    // rows it reads in the L2 of the XCD that wrote them (and this kernel the rows its producer wrote)
    const int lid_ = xcd ? xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y) : blockIdx.x + gridDim.x * blockIdx.y;
    const int b = lid_ / gridDim.x, chunk = lid_ - b * gridDim.x;
    const int t = threadIdx.x;
    {   // reduce the stats partials of sample b: 4 thread-slices x groups, fixed order (deterministic)
        const int g = t & 63, j = t >> 6;
        float s = 0.f, ss = 0.f;
        if (g < groups) {
            const float2* src = reinterpret_cast<const float2*>(partial) + (int64_t)b * nstat * groups + g;
            // eight partials in flight per thread (round 6): the runtime-length load -> add loop was a chain of up to eight memory
            // round trips in the prologue of EVERY workgroup (nstat = 32), longer than the streaming part at the 64x64 level;
            // same summation order
            for (int k0 = j; k0 < nstat; k0 += 32) {
                float2 a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = src[(int64_t)min(k0 + 4 * u, nstat - 1) * groups];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 4 * u < nstat) { s += a[u].x; ss += a[u].y; }
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
    }
    __syncthreads();
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int tpr = min(nvec, 256);
    const int rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    const int nv0 = c0 >> 3;
    if (rsub >= rs || rbeg >= rend) return;
    const int poff = bper > 0 ? (b / bper) * pstride : 0;  // per-stream affine parameters (grouped execution)
    for (int cv = cvl; cv < nvec; cv += tpr) {
        float a[8], sh[8];
        {
            const float4* g4 = reinterpret_cast<const float4*>(gamma + poff + cv * 8);
            const float4* b4 = reinterpret_cast<const float4*>(beta + poff + cv * 8);
            const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const int ca = cv * 8, ga = ca / cpg, split = (ga + 1) * cpg - ca;
            const float2 sa = stat[ga], sb = stat[min(ga + 1, groups - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 st = (cpg < 8) ? stat[(ca + i) / cpg] : ((i < split) ? sa : sb);
                a[i] = st.y * gm[i];
                sh[i] = bt[i] - st.x * a[i];
            }
        }
        const T* base;
        const lo_t<T>* lbase;
        int64_t ld;
        int co;
        if (cv < nv0) { base = x0 + (int64_t)b * rows * c0; lbase = x0_lo ? x0_lo + (int64_t)b * rows * c0 : nullptr; ld = c0; co = cv * 8; }
        else { base = x1 + (int64_t)b * rows * c1; lbase = x1_lo ? x1_lo + (int64_t)b * rows * c1 : nullptr; ld = c1; co = (cv - nv0) * 8; }
        T* ob = out + (int64_t)b * rows * C + cv * 8;
        for (int r = rbeg + rsub; r < rend; r += 4 * rs) {  // 4 loads in flight (clamped rows), then normalise + store
            typename Vec8<T>::type raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                raw[u] = *reinterpret_cast<const typename Vec8<T>::type*>(base + (int64_t)min(r + u * rs, rend - 1) * ld + co);
            float rlo[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (lbase) load_lo<8>(lbase + (int64_t)min(r + u * rs, rend - 1) * ld + co, rlo[u]);
                else
#pragma unroll
                    for (int i = 0; i < 8; ++i) rlo[u][i] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * rs < rend) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float y = ((float)raw[u][i] + rlo[u][i]) * a[i] + sh[i];
                        v[i] = silu ? silu_f(y) : y;
                    }
                    store8(ob + (int64_t)(r + u * rs) * C, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave owns R consecutive rows, all held in registers (one read, exact two-pass variance).
// R rows per wave keep R independent 16-byte loads in flight per lane -- the kernel is latency-bound: with one
// row per wave a C = 320 row is a single 640-byte request per wave.
// ------------------------------------------------------------------------------------------
template <typename T, int MAXV, int R>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const lo_t<T>* __restrict__ x_lo, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int rows, int C,
                                                        int rows_per_set, int pstride, T* __restrict__ out, int xcd) {
    const int lane = threadIdx.x & 63;
    const int bid = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // contiguous row range per XCD (see gn_stats_kernel)
    const int row0 = (bid * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nvec = C >> 3;
    float v[R][MAXV][8];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
        const T* xr = x + (int64_t)min(row0 + r, rows - 1) * C;
        const lo_t<T>* xl = x_lo ? x_lo + (int64_t)min(row0 + r, rows - 1) * C : nullptr;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
                load8(xr + cv * 8, v[r][k]);
                if (xl) {
                    float l[8];
                    load_lo<8>(xl + cv * 8, l);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[r][k][i] += l[i];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < MAXV; ++k)
            if (lane + k * 64 < nvec) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[r] += v[r][k][i];
            }
    float mean[R], ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ss[r] = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k)
            if (lane + k * 64 < nvec) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = v[r][k][i] - mean[r];
                    ss[r] += d * d;
                }
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        const float rstd = rsqrtf(wave_sum(ss[r]) / (float)C + eps);
        if (row >= rows) continue;
        const int poff = rows_per_set > 0 ? (row / rows_per_set) * pstride : 0;  // per-stream affine parameters
        T* orow = out + (int64_t)row * C;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int cv = lane + k * 64;
            if (cv < nvec) {
                float o[8];
                const float4* g4 = reinterpret_cast<const float4*>(gamma + poff + cv * 8);
                const float4* b4 = reinterpret_cast<const float4*>(beta + poff + cv * 8);
                const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (v[r][k][i] - mean[r]) * rstd * g[i] + bb[i];
                store8(orow + cv * 8, o);
            }
        }
    }
}

// LayerNorm for the SD channel counts C = 320 / 640 / 1280 (C / 8 = 5 * L vectors, L = 8 / 16 / 32): L lanes share a
// row, every lane holds five 16-byte vectors, so all 64 lanes are busy (a C = 320 row only fills 40 lanes of the
// one-row-per-wave kernel), five loads are in flight per lane and the reductions run over L lanes only.
template <typename T, int L>
__global__ void __launch_bounds__(256) layernorm5_kernel(const T* __restrict__ x, const lo_t<T>* __restrict__ x_lo, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int rows, int C,
                                                         int rows_per_set, int pstride, T* __restrict__ out, int xcd) {
    constexpr int RPW = 64 / L;  // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % L;
    const int bid = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int row = (bid * 4 + (threadIdx.x >> 6)) * RPW + lane / L;
    const bool valid = row < rows;
    const T* xr = x + (int64_t)min(row, rows - 1) * C;
    float v[5][8];
#pragma unroll
    for (int k = 0; k < 5; ++k) load8(xr + (sub + k * L) * 8, v[k]);
    if (x_lo) {
        const lo_t<T>* xl = x_lo + (int64_t)min(row, rows - 1) * C;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float l[8];
            load_lo<8>(xl + (sub + k * L) * 8, l);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] += l[i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[k][i];
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = v[k][i] - mean;
            ss += d * d;
        }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss / (float)C + eps);
    if (!valid) return;
    const int poff = rows_per_set > 0 ? (row / rows_per_set) * pstride : 0;  // per-stream affine parameters
    T* orow = out + (int64_t)row * C;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int cv = sub + k * L;
        float o[8];
        const float4* g4 = reinterpret_cast<const float4*>(gamma + poff + cv * 8);
        const float4* b4 = reinterpret_cast<const float4*>(beta + poff + cv * 8);
        const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[k][i] - mean) * rstd * g[i] + bb[i];
        store8(orow + cv * 8, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-launch GroupNorm(+SiLU) for the small maps (deep levels): one workgroup per (sample, group) makes two sweeps
// over its [rows][cpg] strip -- sums over the hi parts, then normalise hi + lo -- the second sweep finds the strip in
// this XCD's L2.  Replaces stats + apply (two launches, a partials round trip) where those are launch-bound: 12.6 us
// for a 1.3 MB map against ~3 us of traffic.  A thread walks (row, piece) pairs, a piece = P channels (2 / 4 / 8
// by the divisibility of the group width); the fixed-order block reduction keeps the result deterministic.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int P>
__device__ __forceinline__ void load_piece(const T* p, float (&v)[P]) {
    if constexpr (P == 8) {
        load8(p, v);
    } else if constexpr (P == 4) {
        uint2 raw = *reinterpret_cast<const uint2*>(p);
        T h[4];
        __builtin_memcpy(h, &raw, 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = to_f(h[i]);
    } else {
        unsigned raw = *reinterpret_cast<const unsigned*>(p);
        T h[2];
        __builtin_memcpy(h, &raw, 4);
        v[0] = to_f(h[0]);
        v[1] = to_f(h[1]);
    }
}
template <typename T, int P>
__device__ __forceinline__ void store_piece(T* p, const float (&v)[P]) {
    if constexpr (P == 8) {
        store8(p, v);
    } else {
        T h[P];
#pragma unroll
        for (int i = 0; i < P; ++i) h[i] = from_f<T>(v[i]);
        if constexpr (P == 4) {
            uint2 raw;
            __builtin_memcpy(&raw, h, 8);
            *reinterpret_cast<uint2*>(p) = raw;
        } else {
            unsigned raw;
            __builtin_memcpy(&raw, h, 4);
            *reinterpret_cast<unsigned*>(p) = raw;
        }
    }
}

constexpr int GNF_THREADS = 1024;  // 16 waves: one round trip per sweep for the strips of the 16x16 / 8x8 levels

template <typename T, int P>
__global__ void __launch_bounds__(GNF_THREADS) gn_fused_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                       const lo_t<T>* __restrict__ x0_lo, const lo_t<T>* __restrict__ x1_lo, int c0,
                                                       int c1, int rows, int groups, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu, int bper,
                                                       int pstride, T* __restrict__ out, int xcd) {
    __shared__ float2 red[GNF_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float2 aff[128];  // per channel of the group: (gamma * rstd, beta - mean * gamma * rstd)
    // a group's strip is cpg * 2 bytes of every 128-byte line it touches: neighbouring groups share lines.  xcd: consecutive
    // logical ids (groups of one sample) on ONE XCD, so a line is fetched into one L2 instead of two or three
    const int lid_ = xcd ? xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y) : blockIdx.x + gridDim.x * blockIdx.y;
    const int t = threadIdx.x, g = lid_ % gridDim.x, b = lid_ / gridDim.x;
    const int C = c0 + c1, cpg = C / groups, ppr = cpg / P;  // pieces per row
    const int total = rows * ppr;
    const int dr = GNF_THREADS / ppr, dp = GNF_THREADS - dr * ppr;  // (row, piece) step of i += GNF_THREADS
    const int64_t row0 = (int64_t)b * rows;
    constexpr int U = 4;  // independent loads in flight per thread

    float s1 = 0.f, s2 = 0.f;
    {
        int r = t / ppr, pc = t - (t / ppr) * ppr;
        for (int i = t; i < total; i += GNF_THREADS * U) {
            float v[U][P];
            int rr = r, pp = pc;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = i + GNF_THREADS * u < total;
                const int rc = ok ? rr : 0;
                const int c = g * cpg + (ok ? pp : 0) * P;
                const T* src = c < c0 ? x0 + (row0 + rc) * c0 + c : x1 + (row0 + rc) * c1 + (c - c0);
                load_piece<T, P>(src, v[u]);
                if (!ok) {
#pragma unroll
                    for (int k = 0; k < P; ++k) v[u][k] = 0.f;
                }
                rr += dr; pp += dp;
                if (pp >= ppr) { pp -= ppr; rr += 1; }
            }
            r = rr; pc = pp;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < P; ++k) { s1 += v[u][k]; s2 += v[u][k] * v[u][k]; }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((t & 63) == 0) red[t >> 6] = make_float2(s1, s2);
    __syncthreads();
    const float n = (float)rows * (float)cpg;
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int w = 0; w < GNF_THREADS / 64; ++w) { sum += red[w].x; sq += red[w].y; }  // fixed order
    const float mean = sum / n;
    const float rstd = rsqrtf(fmaxf(sq / n - mean * mean, 0.f) + eps);
    const int poff = bper > 0 ? (b / bper) * pstride : 0;  // per-stream affine parameters (grouped execution)
    if (t < cpg) {
        const float a = gamma[poff + g * cpg + t] * rstd;
        aff[t] = make_float2(a, beta[poff + g * cpg + t] - mean * a);
    }
    __syncthreads();

    int r = t / ppr, pc = t - (t / ppr) * ppr;
    for (int i = t; i < total; i += GNF_THREADS * U) {
        float v[U][P];
        int cs[U], rs[U];
        bool oks[U];
        int rr = r, pp = pc;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = i + GNF_THREADS * u < total;
            const int rc = ok ? rr : 0;
            const int c = g * cpg + (ok ? pp : 0) * P;
            oks[u] = ok; cs[u] = c; rs[u] = rc;
            const bool first = c < c0;
            const int64_t off = first ? (row0 + rc) * c0 + c : (row0 + rc) * c1 + (c - c0);
            load_piece<T, P>((first ? x0 : x1) + off, v[u]);
            const lo_t<T>* lo = first ? x0_lo : x1_lo;
            if (lo) {
                float w[P];
                load_lo<P>(lo + off, w);
#pragma unroll
                for (int k = 0; k < P; ++k) v[u][k] += w[k];
            }
            rr += dr; pp += dp;
            if (pp >= ppr) { pp -= ppr; rr += 1; }
        }
        r = rr; pc = pp;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!oks[u]) continue;
            float y[P];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float2 ab = aff[cs[u] - g * cpg + k];
                float z = fmaf(v[u][k], ab.x, ab.y);
                if (silu) z = silu_f(z);
                y[k] = z;
            }
            store_piece<T, P>(out + (row0 + rs[u]) * C + cs[u], y);
        }
    }
}

// Register-resident variant of the one-launch GroupNorm (round 6): when a (sample, group) strip is at most NP pieces per thread,
// every piece (hi AND lo) is loaded ONCE, all loads issued back to back, kept as raw bits in registers across the block
// reduction, then normalised and stored -- one memory round trip instead of two (the second sweep of gn_fused_kernel re-read
// the strip from L2: on these 5-12 us launches a dependent round trip is 1-2 us).  Statistics over the hi parts only, in the
// same per-thread order and the same block reduction as gn_fused_kernel: results are bit-identical to it.
template <typename T, int P> struct RawPiece;
template <typename T> struct RawPiece<T, 8> { uint4 v; };
template <typename T> struct RawPiece<T, 4> { uint2 v; };
template <typename T> struct RawPiece<T, 2> { unsigned v; };
template <typename L, int P> struct RawLo;  // P low parts: e5m2 bytes (fp16 streams) or bf16
template <> struct RawLo<unsigned char, 8> { uint2 v; };
template <> struct RawLo<unsigned char, 4> { unsigned v; };
template <> struct RawLo<unsigned char, 2> { unsigned short v; };
template <> struct RawLo<bf16, 8> { uint4 v; };
template <> struct RawLo<bf16, 4> { uint2 v; };
template <> struct RawLo<bf16, 2> { unsigned v; };

__device__ __forceinline__ void opaque(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void opaque(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void opaque(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(unsigned short& v) { unsigned t = v; asm volatile("" : "+v"(t)); v = (unsigned short)t; }

template <typename T, int P>
__device__ __forceinline__ void unpack_piece(const RawPiece<T, P>& r, float (&v)[P]) {
    T h[P];
    __builtin_memcpy(h, &r.v, sizeof(T) * P);
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = to_f(h[i]);
}
template <typename L, int P>
__device__ __forceinline__ void unpack_lo(const RawLo<L, P>& r, float (&v)[P]) {
    L h[P];
    __builtin_memcpy(h, &r.v, sizeof(L) * P);
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = lo_to_f(h[i]);
}

template <typename T, int P, int NP>
__global__ void __launch_bounds__(GNF_THREADS) gn_resident_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                          const lo_t<T>* __restrict__ x0_lo, const lo_t<T>* __restrict__ x1_lo, int c0,
                                                          int c1, int rows, int groups, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int silu, int bper,
                                                          int pstride, T* __restrict__ out, int xcd) {
    __shared__ float2 red[GNF_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float2 aff[128];
    const int lid_ = xcd ? xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y) : blockIdx.x + gridDim.x * blockIdx.y;
    const int t = threadIdx.x, g = lid_ % gridDim.x, b = lid_ / gridDim.x;
    const int C = c0 + c1, cpg = C / groups, ppr = cpg / P;
    const int total = rows * ppr;
    const int64_t row0 = (int64_t)b * rows;
    const bool first = g * cpg < c0;  // a group lies entirely in one source (c0 is a multiple of the group width when c1 > 0)
    const T* src = first ? x0 : x1;
    const lo_t<T>* lsrc = first ? x0_lo : x1_lo;
    const int cs = first ? c0 : c1, cbase = first ? g * cpg : g * cpg - c0;

    RawPiece<T, P> hi[NP];
    RawLo<lo_t<T>, P> lo[NP];
    // element offset of piece i of the strip inside its source (recomputed where needed: an offset array would cost 2 * NP registers)
    auto piece_off = [&](int i) __attribute__((always_inline)) {
        const int r = i / ppr, pc = i - r * ppr;
        return (row0 + r) * cs + cbase + pc * P;
    };
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int i = min(t + u * GNF_THREADS, total - 1);  // clamped: out-of-range slots re-read the last piece, weighted 0 below
        hi[u].v = *reinterpret_cast<const decltype(hi[u].v)*>(src + piece_off(i));
    }
    if (lsrc) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
            lo[u].v = *reinterpret_cast<const decltype(lo[u].v)*>(lsrc + piece_off(min(t + u * GNF_THREADS, total - 1)));
    }
    // statistics over the hi parts, pieces in gn_fused_kernel's per-thread order (i = t, t + 1024, ...)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        if (t + u * GNF_THREADS < total) {
            float v[P];
            unpack_piece<T, P>(hi[u], v);
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
    for (int w = 0; w < GNF_THREADS / 64; ++w) { sum += red[w].x; sq += red[w].y; }  // fixed order
    const float mean = sum / n;
    const float rstd = rsqrtf(fmaxf(sq / n - mean * mean, 0.f) + eps);
    const int poff = bper > 0 ? (b / bper) * pstride : 0;
    if (t < cpg) {
        const float a = gamma[poff + g * cpg + t] * rstd;
        aff[t] = make_float2(a, beta[poff + g * cpg + t] - mean * a);
    }
    __syncthreads();
    // keep the strip as RAW bits across the reduction: without this the compiler holds the unpacked floats of the statistics
    // pass alive (2-4x the registers) and the larger instantiations spill
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        opaque(hi[u].v);
        opaque(lo[u].v);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int i = t + u * GNF_THREADS;
        if (i < total) {
            const int r = i / ppr, pc = i - r * ppr;
            float v[P], y[P];
            unpack_piece<T, P>(hi[u], v);
            if (lsrc) {
                float w[P];
                unpack_lo<lo_t<T>, P>(lo[u], w);
#pragma unroll
                for (int k = 0; k < P; ++k) v[k] += w[k];
            }
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float2 ab = aff[pc * P + k];
                float z = fmaf(v[k], ab.x, ab.y);
                if (silu) z = silu_f(z);
                y[k] = z;
            }
            store_piece<T, P>(out + (row0 + r) * C + g * cpg + pc * P, y);
        }
    }
}

}  // namespace ur

using namespace ur;

static int gn_check(const void* x0, const void* x1, int c0, int c1, int B, int rows, int groups, int nchunks) {
    if (!x0 || c0 <= 0 || (c0 & 7) || (c1 & 7) || c1 < 0 || (c1 > 0 && !x1)) return UR_E_BADARG;
    if (B <= 0 || rows <= 0 || groups <= 0 || groups > 64 || nchunks <= 0 || nchunks > 65535) return UR_E_BADARG;
    if ((c0 + c1) % groups) return UR_E_BADARG;
    return 0;
}

extern "C" int ur_groupnorm_stats(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1,
                                  int B, int rows, int groups, int nchunks, float* partial, int dtype, void* stream) {
    int rc = gn_check(x0, x1, c0, c1, B, rows, groups, nchunks);
    if (rc || !partial) return rc ? rc : UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(nchunks, B);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((gn_stats_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1,
                           (const lo_t<f16>*)x0_lo, (const lo_t<f16>*)x1_lo, c0, c1, rows, groups, nchunks, partial, norm_xcd());
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((gn_stats_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x0, (const bf16*)x1,
                           (const lo_t<bf16>*)x0_lo, (const lo_t<bf16>*)x1_lo, c0, c1, rows, groups, nchunks, partial, norm_xcd());
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_groupnorm_apply(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1,
                                  int B, int rows, int groups, int nstat, int nchunks, const float* partial, const float* gamma,
                                  const float* beta, float eps, int silu, int bper, int pstride, void* out,
                                  int dtype, void* stream) {
    int rc = gn_check(x0, x1, c0, c1, B, rows, groups, nchunks);
    if (rc || nstat <= 0 || !partial || !gamma || !beta || !out) return rc ? rc : UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(nchunks, B);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((gn_apply_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1,
                           (const lo_t<f16>*)x0_lo, (const lo_t<f16>*)x1_lo, c0, c1, rows, groups, nstat, nchunks, partial, gamma, beta,
                           eps, silu, bper, pstride, (f16*)out, norm_xcd());
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x0, (const bf16*)x1,
                           (const lo_t<bf16>*)x0_lo, (const lo_t<bf16>*)x1_lo, c0, c1, rows, groups, nstat, nchunks, partial, gamma,
                           beta, eps, silu, bper, pstride, (bf16*)out, norm_xcd());
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// One launch: statistics + normalisation by one workgroup per (sample, group) (csrc comment at gn_fused_kernel).
// Meant for maps up to a few tens of MB; larger ones are faster through ur_groupnorm_stats + ur_groupnorm_apply.
template <typename T>
static int launch_gn_fused(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1, int B,
                           int rows, int groups, const float* gamma, const float* beta, float eps, int silu, int bper,
                           int pstride, void* out, hipStream_t s) {
    const int cpg = (c0 + c1) / groups;
    dim3 grid(groups, B);
    if (cpg > 128) return UR_E_UNSUPPORTED;  // the group's affine pairs are staged in a 128-entry LDS table
    {
        // register-resident single sweep where the strip fits (<= 4 / 8 pieces per thread) and no group straddles the two sources
        const int P = (cpg % 8 == 0) ? 8 : (cpg % 4 == 0) ? 4 : (cpg % 2 == 0) ? 2 : 0;
        const int64_t per_thread = P ? ((int64_t)rows * (cpg / P) + GNF_THREADS - 1) / GNF_THREADS : 0;
        const bool whole = c1 == 0 || (c0 % cpg) == 0;
                // bounds = the instantiations that fit 128 VGPRs (1024 threads per workgroup) without spilling
        const int64_t most = P == 8 ? (sizeof(lo_t<T>) == 1 ? 8 : 4) : P == 4 ? 16 : 20;
        if (!(silu & 2) && P && whole && per_thread >= 1 && per_thread <= most) {
            const int np = per_thread <= 2 ? 2 : per_thread <= 4 ? 4 : per_thread <= 8 ? 8 : per_thread <= 16 ? 16 : 20;
#define UR_GNR(PP, NN)                                                                                                  \
    hipLaunchKernelGGL((gn_resident_kernel<T, PP, NN>), grid, dim3(GNF_THREADS), 0, s, (const T*)x0, (const T*)x1, (const lo_t<T>*)x0_lo, \
                       (const lo_t<T>*)x1_lo, c0, c1, rows, groups, gamma, beta, eps, silu & 1, bper, pstride, (T*)out, gnf_xcd())
#define UR_GNR_P(PP)                                                                \
    do {                                                                            \
        if (np == 2) UR_GNR(PP, 2); else if (np == 4) UR_GNR(PP, 4); else if (np == 8) UR_GNR(PP, 8); else UR_GNR(PP, 16); \
    } while (0)
            if (P == 8) { if (np == 2) UR_GNR(8, 2); else if (np == 4) UR_GNR(8, 4); else UR_GNR(8, 8); }
            else if (P == 4) UR_GNR_P(4);
            else if (np == 20) UR_GNR(2, 20);
            else UR_GNR_P(2);
#undef UR_GNR_P
#undef UR_GNR
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : -(int)e;
        }
    }
#define UR_GNF(PP)                                                                                                     \
    hipLaunchKernelGGL((gn_fused_kernel<T, PP>), grid, dim3(GNF_THREADS), 0, s, (const T*)x0, (const T*)x1, (const lo_t<T>*)x0_lo, \
                       (const lo_t<T>*)x1_lo, c0, c1, rows, groups, gamma, beta, eps, silu & 1, bper, pstride, (T*)out, gnf_xcd())
    if (cpg % 8 == 0) UR_GNF(8);
    else if (cpg % 4 == 0) UR_GNF(4);
    else if (cpg % 2 == 0) UR_GNF(2);
    else return UR_E_UNSUPPORTED;
#undef UR_GNF
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_groupnorm_fused(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1,
                                  int B, int rows, int groups, const float* gamma, const float* beta, float eps, int silu,
                                  int bper, int pstride, void* out, int dtype, void* stream) {
    int rc = gn_check(x0, x1, c0, c1, B, rows, groups, 1);
    if (rc || !gamma || !beta || !out) return rc ? rc : UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == UR_DT_F16)
        return launch_gn_fused<f16>(x0, x1, x0_lo, x1_lo, c0, c1, B, rows, groups, gamma, beta, eps, silu, bper, pstride, out, s);
    if (dtype == UR_DT_BF16)
        return launch_gn_fused<bf16>(x0, x1, x0_lo, x1_lo, c0, c1, B, rows, groups, gamma, beta, eps, silu, bper, pstride, out, s);
    return UR_E_BADARG;
}

template <typename T>
static void launch_ln(const void* x, const void* x_lo, const float* gamma, const float* beta, float eps, int rows, int C,
                      int rows_per_set, int pstride, void* out, hipStream_t s) {
    // rows per wave chosen so that a wave keeps >= 4 16-byte loads per lane in flight and the grid still fills the chip
#define UR_LN(MAXV, R)                                                                                              \
    hipLaunchKernelGGL((layernorm_kernel<T, MAXV, R>), dim3((rows + 4 * R - 1) / (4 * R)), dim3(256), 0, s,          \
                       (const T*)x, (const lo_t<T>*)x_lo, gamma, beta, eps, rows, C, rows_per_set, pstride, (T*)out, norm_xcd())
#define UR_LN5(LL)                                                                                                   \
    hipLaunchKernelGGL((layernorm5_kernel<T, LL>), dim3((rows + 4 * (64 / LL) - 1) / (4 * (64 / LL))), dim3(256), 0, s, \
                       (const T*)x, (const lo_t<T>*)x_lo, gamma, beta, eps, rows, C, rows_per_set, pstride, (T*)out, norm_xcd())
    if (C == 320) { UR_LN5(8); return; }
    if (C == 640) { UR_LN5(16); return; }
    if (C == 1280) { UR_LN5(32); return; }
#undef UR_LN5
    const bool big = rows >= 8192;  // enough rows to give every CU several workgroups even at R = 4
    if (C <= 512) { if (big) UR_LN(1, 4); else UR_LN(1, 1); }
    else if (C <= 1024) { if (big) UR_LN(2, 2); else UR_LN(2, 1); }
    else if (C <= 2048) { if (big) UR_LN(4, 2); else UR_LN(4, 1); }
    else UR_LN(8, 1);
#undef UR_LN
}

extern "C" int ur_layernorm(const void* x, const void* x_lo, const float* gamma, const float* beta, float eps, int rows,
                            int C, int rows_per_set, int pstride, void* out, int dtype, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0 || C <= 0 || (C & 7) || C > 4096) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == UR_DT_F16) launch_ln<f16>(x, x_lo, gamma, beta, eps, rows, C, rows_per_set, pstride, out, s);
    else if (dtype == UR_DT_BF16) launch_ln<bf16>(x, x_lo, gamma, beta, eps, rows, C, rows_per_set, pstride, out, s);
    else return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
