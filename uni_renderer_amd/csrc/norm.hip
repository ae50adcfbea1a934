// GroupNorm(+SiLU) over NHWC (optionally over the concat of two sources) and LayerNorm, gfx950.
// Both are HBM-bound: 16-byte vector accesses, fp32 statistics, wave-shuffle / fixed-order LDS
// reductions (deterministic: no floating-point atomics anywhere).
#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

// ------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid = (nchunks, B); a workgroup sums rows [chunk*rpc, (chunk+1)*rpc) of
// sample b over all channels and writes partial[b][chunk][g] = (sum, sumsq).
// Thread mapping: a thread owns one 8-channel vector (fixed) and strides over rows, so consecutive
// threads read consecutive 16-B pieces of a row and, because NHWC rows are contiguous, of the next row.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                       int c1, int rows, int groups, int nchunks,
                                                       float* __restrict__ partial) {
    __shared__ float2 chs[2048];  // per-(row-subset, channel) sums of the current pass
    __shared__ float2 gacc[64];
    const int C = c0 + c1, nvec = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int t = threadIdx.x;
    const int tpr = min(nvec, 256);
    const int rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    const int nv0 = c0 >> 3;
    float gs = 0.f, gss = 0.f;  // thread g < groups accumulates its group over the passes
    for (int cvb = 0; cvb < nvec; cvb += tpr) {
        const int cv = cvb + cvl;
        float s[8], ss[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; }
        if (rsub < rs && cv < nvec) {
            const T* base;
            int64_t ld;
            int co;
            if (cv < nv0) { base = x0 + (int64_t)b * rows * c0; ld = c0; co = cv * 8; }
            else { base = x1 + (int64_t)b * rows * c1; ld = c1; co = (cv - nv0) * 8; }
            int r = rbeg + rsub;
            for (; r + 7 * rs < rend; r += 8 * rs) {  // 8 independent 16-byte loads in flight per thread
                typename Vec8<T>::type raw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    raw[u] = *reinterpret_cast<const typename Vec8<T>::type*>(base + (int64_t)(r + u * rs) * ld + co);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = (float)raw[u][i];
                        s[i] += v;
                        ss[i] += v * v;
                    }
            }
            for (; r + 1 * rs < rend; r += 2 * rs) {
                float v0[8], v1[8];
                load8(base + (int64_t)r * ld + co, v0);
                load8(base + (int64_t)(r + rs) * ld + co, v1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s[i] += v0[i] + v1[i];
                    ss[i] += v0[i] * v0[i] + v1[i] * v1[i];
                }
            }
            for (; r < rend; r += rs) {
                float v[8];
                load8(base + (int64_t)r * ld + co, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) { s[i] += v[i]; ss[i] += v[i] * v[i]; }
            }
        }
        __syncthreads();  // previous pass fully consumed
        if (rsub < rs) {
#pragma unroll
            for (int i = 0; i < 8; ++i) chs[rsub * (tpr * 8) + cvl * 8 + i] = make_float2(s[i], ss[i]);
        }
        __syncthreads();
        if (t < groups) {
            // channels of this pass: [cvb*8, cvb*8 + tpr*8) intersected with the group's range
            const int pb = cvb * 8, pe = min(C, pb + tpr * 8);
            const int cb = max(pb, t * cpg), ce = min(pe, (t + 1) * cpg);
            for (int c = cb; c < ce; ++c)
                for (int k = 0; k < rs; ++k) {
                    float2 a = chs[k * (tpr * 8) + (c - pb)];
                    gs += a.x;
                    gss += a.y;
                }
        }
    }
    if (t < groups) {
        float2* dst = reinterpret_cast<float2*>(partial) + ((int64_t)b * nchunks + chunk) * groups + t;
        *dst = make_float2(gs, gss);
    }
    (void)gacc;
}

template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                       int c1, int rows, int groups, int nstat, int nchunks,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu,
                                                       int bper, int pstride, T* __restrict__ out) {
    __shared__ float2 stat[64];  // (mean, rstd) per group
    __shared__ float2 red[8][64];
    const int C = c0 + c1, nvec = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t = threadIdx.x;
    {   // reduce the stats partials of sample b: 8 thread-slices x groups, fixed order (deterministic)
        const int g = t & 63, j = t >> 6;  // 4 slices of up to 64 groups
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
    }
    __syncthreads();
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int rbeg = chunk * rpc, rend = min(rows, rbeg + rpc);
    const int tpr = min(nvec, 256);
    const int rs = 256 / tpr;
    const int rsub = t / tpr, cvl = t - rsub * tpr;
    const int nv0 = c0 >> 3;
    if (rsub >= rs) return;
    const int poff = bper > 0 ? (b / bper) * pstride : 0;  // per-stream affine parameters (grouped execution)
    for (int cv = cvl; cv < nvec; cv += tpr) {
        float a[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cv * 8 + i;
            const float2 st = stat[c / cpg];
            a[i] = st.y * gamma[poff + c];
            sh[i] = beta[poff + c] - st.x * a[i];
        }
        const T* base;
        int64_t ld;
        int co;
        if (cv < nv0) { base = x0 + (int64_t)b * rows * c0; ld = c0; co = cv * 8; }
        else { base = x1 + (int64_t)b * rows * c1; ld = c1; co = (cv - nv0) * 8; }
        T* ob = out + (int64_t)b * rows * C + cv * 8;
        int r = rbeg + rsub;
        for (; r + 3 * rs < rend; r += 4 * rs) {  // 4 loads in flight, then normalise + store
            typename Vec8<T>::type raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                raw[u] = *reinterpret_cast<const typename Vec8<T>::type*>(base + (int64_t)(r + u * rs) * ld + co);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float y = (float)raw[u][i] * a[i] + sh[i];
                    v[i] = silu ? silu_f(y) : y;
                }
                store8(ob + (int64_t)(r + u * rs) * C, v);
            }
        }
        for (; r < rend; r += rs) {
            float v[8];
            load8(base + (int64_t)r * ld + co, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = v[i] * a[i] + sh[i];
                v[i] = silu ? silu_f(y) : y;
            }
            store8(ob + (int64_t)r * C, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, values held in registers (one read, exact two-pass variance).
// ------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int rows, int C,
                                                        int rows_per_set, int pstride, T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int poff = rows_per_set > 0 ? (row / rows_per_set) * pstride : 0;  // per-stream affine parameters
    gamma += poff;
    beta += poff;
    const int nvec = C >> 3;
    const T* xr = x + (int64_t)row * C;
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int cv = lane + k * 64;
        if (cv < nvec) {
            load8(xr + cv * 8, v[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[k][i];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int cv = lane + k * 64;
        if (cv < nvec) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = v[k][i] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    T* orow = out + (int64_t)row * C;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int cv = lane + k * 64;
        if (cv < nvec) {
            float o[8];
            const float4* g4 = reinterpret_cast<const float4*>(gamma + cv * 8);
            const float4* b4 = reinterpret_cast<const float4*>(beta + cv * 8);
            const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[k][i] - mean) * rstd * g[i] + bb[i];
            store8(orow + cv * 8, o);
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

extern "C" int ur_groupnorm_stats(const void* x0, const void* x1, int c0, int c1, int B, int rows, int groups,
                                  int nchunks, float* partial, int dtype, void* stream) {
    int rc = gn_check(x0, x1, c0, c1, B, rows, groups, nchunks);
    if (rc || !partial) return rc ? rc : UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(nchunks, B);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((gn_stats_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1, c0, c1, rows,
                           groups, nchunks, partial);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((gn_stats_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x0, (const bf16*)x1, c0, c1,
                           rows, groups, nchunks, partial);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_groupnorm_apply(const void* x0, const void* x1, int c0, int c1, int B, int rows, int groups,
                                  int nstat, int nchunks, const float* partial, const float* gamma,
                                  const float* beta, float eps, int silu, int bper, int pstride, void* out,
                                  int dtype, void* stream) {
    int rc = gn_check(x0, x1, c0, c1, B, rows, groups, nchunks);
    if (rc || nstat <= 0 || !partial || !gamma || !beta || !out) return rc ? rc : UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(nchunks, B);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((gn_apply_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1, c0, c1, rows,
                           groups, nstat, nchunks, partial, gamma, beta, eps, silu, bper, pstride, (f16*)out);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x0, (const bf16*)x1, c0, c1,
                           rows, groups, nstat, nchunks, partial, gamma, beta, eps, silu, bper, pstride, (bf16*)out);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_layernorm(const void* x, const float* gamma, const float* beta, float eps, int rows, int C,
                            int rows_per_set, int pstride, void* out, int dtype, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0 || C <= 0 || (C & 7) || C > 4096) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4);
    const bool small = C <= 2048;
    if (dtype == UR_DT_F16) {
        if (small)
            hipLaunchKernelGGL((layernorm_kernel<f16, 4>), grid, dim3(256), 0, s, (const f16*)x, gamma, beta, eps, rows,
                               C, rows_per_set, pstride, (f16*)out);
        else
            hipLaunchKernelGGL((layernorm_kernel<f16, 8>), grid, dim3(256), 0, s, (const f16*)x, gamma, beta, eps, rows,
                               C, rows_per_set, pstride, (f16*)out);
    } else if (dtype == UR_DT_BF16) {
        if (small)
            hipLaunchKernelGGL((layernorm_kernel<bf16, 4>), grid, dim3(256), 0, s, (const bf16*)x, gamma, beta, eps,
                               rows, C, rows_per_set, pstride, (bf16*)out);
        else
            hipLaunchKernelGGL((layernorm_kernel<bf16, 8>), grid, dim3(256), 0, s, (const bf16*)x, gamma, beta, eps,
                               rows, C, rows_per_set, pstride, (bf16*)out);
    } else {
        return UR_E_BADARG;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
