// Small HBM-bound helpers: elementwise add (the enc->unet skip exchange), sinusoidal timestep
// embedding, and NCHW <-> NHWC layout glue at the module boundary.
#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

template <typename T>
__global__ void __launch_bounds__(256) add_kernel(const T* __restrict__ a, const T* __restrict__ b, float alpha,
                                                  T* __restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        load8(a + i * 8, x);
        load8(b + i * 8, y);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] += alpha * y[k];
        store8(out + i * 8, x);
    }
}

// diffusers get_timestep_embedding: emb = t * exp(-ln(1e4) * i / (half - shift)); [sin | cos], flipped
// to [cos | sin] when flip_sin_to_cos.
template <typename T>
__global__ void timestep_kernel(const float* __restrict__ t, int nt, int B, int dim, int flip, float shift,
                                T* __restrict__ out) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const float tv = t[nt == 1 ? 0 : b];
    const float freq = expf(-9.210340371976184f * (float)i / ((float)half - shift));
    const float arg = tv * freq;
    const float s = sinf(arg), c = cosf(arg);
    T* o = out + (int64_t)b * dim;
    if (flip) { o[i] = (T)c; o[half + i] = (T)s; }
    else { o[i] = (T)s; o[half + i] = (T)c; }
    if ((dim & 1) && i == 0) o[dim - 1] = (T)0.f;
}

template <typename S> __device__ __forceinline__ float ld_any(const S* p, int64_t i) { return (float)p[i]; }

template <typename S, typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const S* __restrict__ src, int B, int C, int H, int W,
                                                           T* __restrict__ dst, int Cpad) {
    const int64_t total = (int64_t)B * H * W * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const int64_t pix = i / Cpad;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int b = (int)(pix / ((int64_t)W * H));
        float v = 0.f;
        if (c < C) v = (float)src[(((int64_t)b * C + c) * H + y) * W + x];
        dst[i] = (T)v;
    }
}

template <typename T, typename S>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T* __restrict__ src, int B, int C, int H, int W,
                                                           S* __restrict__ dst) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int b = (int)(i / ((int64_t)W * H * C));
        dst[i] = (S)(float)src[(((int64_t)b * H + y) * W + x) * C + c];
    }
}

static inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}


// out[b][oy][ox][:] = in[b][sy(oy)][sx(ox)][:] with PyTorch's 'nearest' rule: s(o) = min(floor(o * (float)in / out), in - 1)
// (F.interpolate(size=...) of Upsample2D when the latent side is not a multiple of 8, controlnet.py:1129-1130).
template <typename T>
__global__ void __launch_bounds__(256) resize_nearest_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Hin,
                                                             int Win, int Hout, int Wout, int C) {
    const int nvec = C >> 3;
    const int64_t total = (int64_t)B * Hout * Wout * nvec;
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % nvec);
        const int64_t pix = i / nvec;
        const int ox = (int)(pix % Wout);
        const int oy = (int)((pix / Wout) % Hout);
        const int b = (int)(pix / ((int64_t)Wout * Hout));
        const int sy = min((int)floorf(oy * sh), Hin - 1), sx = min((int)floorf(ox * sw), Win - 1);
        typedef typename Vec8<T>::type vec8;
        *reinterpret_cast<vec8*>(out + pix * C + cv * 8) =
            *reinterpret_cast<const vec8*>(in + (((int64_t)b * Hin + sy) * Win + sx) * C + cv * 8);
    }
}

}  // namespace ur

using namespace ur;

extern "C" int ur_add(const void* a, const void* b, float alpha, void* out, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0 || (n & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t nvec = n / 8;
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((add_kernel<f16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const f16*)a, (const f16*)b, alpha,
                           (f16*)out, nvec);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((add_kernel<bf16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b,
                           alpha, (bf16*)out, nvec);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
__global__ void __launch_bounds__(256) add_hilo_kernel(const T* __restrict__ a, const lo_t<T>* __restrict__ a_lo,
                                                       const T* __restrict__ b, const lo_t<T>* __restrict__ b_lo, float alpha,
                                                       T* __restrict__ out, lo_t<T>* __restrict__ out_lo, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8], t[8];
        load8(a + i * 8, x);
        load8(b + i * 8, y);
        if (a_lo) {
            load_lo<8>(a_lo + i * 8, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] += t[k];
        }
        if (b_lo) {
            load_lo<8>(b_lo + i * 8, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] += t[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] + alpha * y[k];
        store8(out + i * 8, x);
        if (out_lo) store_lo8<T>(out_lo + i * 8, x);
    }
}

extern "C" int ur_add_hilo(const void* a, const void* a_lo, const void* b, const void* b_lo, float alpha, void* out,
                           void* out_lo, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0 || (n & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t nvec = n / 8;
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((add_hilo_kernel<f16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const f16*)a, (const lo_t<f16>*)a_lo,
                           (const f16*)b, (const lo_t<f16>*)b_lo, alpha, (f16*)out, (lo_t<f16>*)out_lo, nvec);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((add_hilo_kernel<bf16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)a,
                           (const lo_t<bf16>*)a_lo, (const bf16*)b, (const lo_t<bf16>*)b_lo, alpha, (bf16*)out, (lo_t<bf16>*)out_lo, nvec);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// ur_add_hilo over up to UR_ADD_MULTI_MAX tensor triples in ONE launch: the 13 exchange adds of a sampling step whose
// 1x1-conv operand is loop-invariant (hoist.py).  Workgroup -> (item, chunk of 1024 8-element vectors) through a prefix table
// in the kernel arguments; same per-element arithmetic as add_hilo_kernel with alpha = 1.
struct AddMultiArgs {
    ur_add_item t[UR_ADD_MULTI_MAX];
    int blk0[UR_ADD_MULTI_MAX + 1];
    int n;
};

template <typename T>
__global__ void __launch_bounds__(256) add_hilo_multi_kernel(const AddMultiArgs a) {
    int it = 0;
    while (it + 1 < a.n && (int)blockIdx.x >= a.blk0[it + 1]) ++it;
    const ur_add_item e = a.t[it];
    const T* pa = (const T*)e.a;
    const T* pb = (const T*)e.b;
    const lo_t<T>* la = (const lo_t<T>*)e.a_lo;
    const lo_t<T>* lb = (const lo_t<T>*)e.b_lo;
    T* po = (T*)e.out;
    lo_t<T>* lo = (lo_t<T>*)e.out_lo;
    const int64_t nvec = e.n / 8;
    const int64_t v0 = (int64_t)((int)blockIdx.x - a.blk0[it]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = v0 + u * 256 + threadIdx.x;
        if (i >= nvec) break;
        float x[8], y[8], t[8];
        load8(pa + i * 8, x);
        load8(pb + i * 8, y);
        if (la) {
            load_lo<8>(la + i * 8, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] += t[k];
        }
        if (lb) {
            load_lo<8>(lb + i * 8, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] += t[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] + 1.0f * y[k];
        store8(po + i * 8, x);
        if (lo) store_lo8<T>(lo + i * 8, x);
    }
}

extern "C" int ur_add_hilo_multi(const ur_add_item* items, int n, int dtype, void* stream) {
    if (!items || n <= 0 || n > UR_ADD_MULTI_MAX) return UR_E_BADARG;
    AddMultiArgs a;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].a || !items[i].b || !items[i].out || items[i].n <= 0 || (items[i].n & 7)) return UR_E_BADARG;
        a.t[i] = items[i];
        a.blk0[i] = blocks;
        const int64_t nb = (items[i].n / 8 + 1023) / 1024;
        if (nb + blocks > (1 << 30)) return UR_E_UNSUPPORTED;
        blocks += (int)nb;
    }
    a.blk0[n] = blocks;
    a.n = n;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == UR_DT_F16) hipLaunchKernelGGL((add_hilo_multi_kernel<f16>), dim3(blocks), dim3(256), 0, s, a);
    else if (dtype == UR_DT_BF16) hipLaunchKernelGGL((add_hilo_multi_kernel<bf16>), dim3(blocks), dim3(256), 0, s, a);
    else return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
extern "C" int ur_sizeof_add_item(void) { return (int)sizeof(ur_add_item); }

extern "C" int ur_timestep_embedding(const float* t, int nt, int B, int dim, int flip_sin_to_cos, float freq_shift,
                                     void* out, int dtype, void* stream) {
    if (!t || !out || B <= 0 || dim < 2 || (nt != 1 && nt != B)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n = B * (dim / 2);
    dim3 grid((n + 255) / 256);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((timestep_kernel<f16>), grid, dim3(256), 0, s, t, nt, B, dim, flip_sin_to_cos, freq_shift,
                           (f16*)out);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((timestep_kernel<bf16>), grid, dim3(256), 0, s, t, nt, B, dim, flip_sin_to_cos, freq_shift,
                           (bf16*)out);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename S>
static int to_nhwc_src(const void* src, int B, int C, int H, int W, void* dst, int Cpad, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W * Cpad;
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<S, f16>), dim3(grid_for(total)), dim3(256), 0, s, (const S*)src, B, C, H,
                           W, (f16*)dst, Cpad);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<S, bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const S*)src, B, C, H,
                           W, (bf16*)dst, Cpad);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_resize_nearest(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, int C, int dtype,
                                 void* stream) {
    if (!in || !out || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || (C & 7)) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * Hout * Wout * (C >> 3);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((resize_nearest_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, (const f16*)in, (f16*)out, B,
                           Hin, Win, Hout, Wout, C);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((resize_nearest_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in, (bf16*)out,
                           B, Hin, Win, Hout, Wout, C);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_nchw_to_nhwc(const void* src, int src_dtype, int B, int C, int H, int W, void* dst, int Cpad,
                               int dtype, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (src_dtype == 0) return to_nhwc_src<f16>(src, B, C, H, W, dst, Cpad, dtype, s);
    if (src_dtype == 1) return to_nhwc_src<bf16>(src, B, C, H, W, dst, Cpad, dtype, s);
    if (src_dtype == 2) return to_nhwc_src<float>(src, B, C, H, W, dst, Cpad, dtype, s);
    return UR_E_BADARG;
}

template <typename T>
static int to_nchw_dst(const void* src, int B, int C, int H, int W, void* dst, int dst_dtype, hipStream_t s) {
    const int64_t total = (int64_t)B * C * H * W;
    if (dst_dtype == 0)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, f16>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, B, C, H,
                           W, (f16*)dst);
    else if (dst_dtype == 1)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, B, C, H,
                           W, (bf16*)dst);
    else if (dst_dtype == 2)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, float>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, B, C, H,
                           W, (float*)dst);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_nhwc_to_nchw(const void* src, int dtype, int B, int C, int H, int W, void* dst, int dst_dtype,
                               void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == UR_DT_F16) return to_nchw_dst<f16>(src, B, C, H, W, dst, dst_dtype, s);
    if (dtype == UR_DT_BF16) return to_nchw_dst<bf16>(src, B, C, H, W, dst, dst_dtype, s);
    return UR_E_BADARG;
}

// ------------------------------------------------------------------------------------------
// On-device sampler update: the DDIM (eta = 0) step of models/pipeline.py:2691-2730 /
// 1645-1649 for a model that predicts x0 ("sample"), applied to C latent channels at once, with the per-step
// scalars read from a device table indexed by a device-side step counter -- so a whole sampling loop is graph
// replays with no host round trip.  Same operation order as the scheduler's fp32 torch expression (no FMA
// contraction), so the fused loop reproduces the unfused one bit for bit:
//     eps  = (x - sqrt(a_t) * x0) / sqrt(1 - a_t);   x' = sqrt(a_prev) * x0 + sqrt(1 - a_prev) * eps
// pred: NHWC [B][HW][pred_ld], channels pred_c0 .. pred_c0 + C.   lat: NCHW [B][C][HW] with batch stride lat_bs.
// master (optional, contiguous fp32 [B][C][HW]): the sampler's own copy of the latents -- the reference keeps them
// in the caller's dtype (fp32 in eval) between steps and only the network input is rounded to fp16/bf16.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ddim_update_kernel(const T* __restrict__ pred, int pred_ld, int pred_c0,
                                                          T* __restrict__ lat, int64_t lat_bs, int C, int B, int HW,
                                                          const float* __restrict__ coef, const int* __restrict__ step,
                                                          int nsteps, float* __restrict__ master, int round_master,
                                                          int cfg, float guidance, int cfg_channels) {
#pragma clang fp contract(off)
    const int st = min(*step, nsteps - 1);
    const float s_at = coef[4 * st + 0], s_1mat = coef[4 * st + 1], s_ap = coef[4 * st + 2], s_1map = coef[4 * st + 3];
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int64_t bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        T* xp = lat + (int64_t)b * lat_bs + (int64_t)c * HW + p;
        const float x = master ? master[i] : (float)*xp;
        T x0t = pred[((int64_t)b * HW + p) * pred_ld + pred_c0 + c];
        if (cfg && c < cfg_channels) {
            // classifier-free guidance as the reference's loop evaluates it in the prediction's dtype (every
            // operation rounded): pred = p_uncond + g * (p_cond - p_uncond), cond = samples [0, B), uncond = [B, 2B)
            const T pu = pred[((int64_t)(b + B) * HW + p) * pred_ld + pred_c0 + c];
            const T d = (T)((float)x0t - (float)pu);
            const T m = (T)((float)d * guidance);
            x0t = (T)((float)pu + (float)m);
        }
        const float x0 = (float)x0t;
        const float eps = (x - s_at * x0) / s_1mat;
        const float a = s_ap * x0;
        const float e = s_1map * eps;
        const float v = a + e;
        *xp = (T)v;
        if (cfg) xp[(int64_t)B * lat_bs] = (T)v;  // the uncond half of the next input is the same latent
        if (master) master[i] = round_master ? (float)(T)v : v;
    }
}

// UniPC (order <= 2, x0 / data prediction, B(h) = bh2) predictor-corrector update as ONE pass (schedulers.py
// UniPCMultistepScheduler.coefficient_table): with m_i the model's x0 prediction at step i,
//     L_i     = c0 L_{i-1} + c1 m_{i-1} + c2 m_{i-2} + c3 m_i      corrected sample ("last_sample"; step 0: L_0 = x_0)
//     x_{i+1} = p0 L_i     + p1 m_i     + p2 m_{i-1}               next model input
// coef row i = (c0, c1, c2, c3, p0, p1, p2, -).  `last` holds L, `xmaster` the evolving x (what the loop returns), `hist`
// the two previous predictions in a 2-slot ring indexed by the device step counter (m_{i-1} in slot (i+1)&1, m_{i-2}
// in slot i&1, which m_i then overwrites).  All fp32; with round_master the samples are rounded through the latent dtype
// where the reference's scheduler casts them (`x_t.to(x.dtype)`).
template <typename T>
__global__ void __launch_bounds__(256) unipc_update_kernel(const T* __restrict__ pred, int pred_ld, int pred_c0, T* __restrict__ lat,
                                                           int64_t lat_bs, int C, int B, int HW, const float* __restrict__ coef,
                                                           const int* __restrict__ step, int nsteps, float* __restrict__ last,
                                                           float* __restrict__ xmaster, float* __restrict__ hist, int round_master,
                                                           int cfg, float guidance, int cfg_channels) {
    const int st = min(*step, nsteps - 1);
    const float* cr = coef + (int64_t)st * 8;
    const float c0 = cr[0], c1 = cr[1], c2 = cr[2], c3 = cr[3], p0 = cr[4], p1 = cr[5], p2 = cr[6];
    const int64_t total = (int64_t)B * C * HW;
    float* h1 = hist + (int64_t)((st + 1) & 1) * total;  // m_{i-1}
    float* h2 = hist + (int64_t)(st & 1) * total;        // m_{i-2}, then m_i
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int64_t bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        T mt = pred[((int64_t)b * HW + p) * pred_ld + pred_c0 + c];
        if (cfg && c < cfg_channels) {  // as in ddim_update_kernel: the reference's guidance arithmetic in the prediction's dtype
            const T pu = pred[((int64_t)(b + B) * HW + p) * pred_ld + pred_c0 + c];
            const T d = (T)((float)mt - (float)pu);
            const T m = (T)((float)d * guidance);
            mt = (T)((float)pu + (float)m);
        }
        const float m = (float)mt, m1 = h1[i], m2 = h2[i];
        float L = c0 * last[i] + c1 * m1 + c2 * m2 + c3 * m;
        if (round_master) L = (float)(T)L;
        float x = p0 * L + p1 * m + p2 * m1;
        const T xt = (T)x;
        if (round_master) x = (float)xt;
        last[i] = L;
        xmaster[i] = x;
        h2[i] = m;
        T* xp = lat + (int64_t)b * lat_bs + (int64_t)c * HW + p;
        *xp = xt;
        if (cfg) xp[(int64_t)B * lat_bs] = xt;
    }
}

// step += 1; t_out[0..B) = tsteps[step] (the timestep the NEXT replay denoises at)
__global__ void sampler_advance_kernel(int* step, const float* __restrict__ tsteps, int nsteps, float* t_out, int B) {
    // ONE thread reads the counter and shares it through LDS: with every thread reading *step, waves 1-3 could see
    // the value thread 0 had already incremented and write tsteps[st + 1] for the samples beyond the first 64
    __shared__ int st_sh;
    if (threadIdx.x == 0) {
        st_sh = *step + 1;
        *step = st_sh;
    }
    __syncthreads();
    const int st = st_sh;
    if (t_out && (int)threadIdx.x < B) t_out[threadIdx.x] = tsteps[min(st, nsteps - 1)];
}

extern "C" int ur_ddim_update(const void* pred, int pred_ld, int pred_c0, void* lat, int64_t lat_bstride, int C, int B,
                              int HW, const float* coef, const int* step, int nsteps, float* master, int round_master,
                              int cfg, float guidance, int cfg_channels, int dtype, void* stream) {
    if (!pred || !lat || !coef || !step || C <= 0 || B <= 0 || HW <= 0 || nsteps <= 0 || pred_c0 < 0 ||
        pred_c0 + C > pred_ld)
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * C * HW;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((ddim_update_kernel<f16>), dim3(grid), dim3(256), 0, s, (const f16*)pred, pred_ld, pred_c0,
                           (f16*)lat, lat_bstride, C, B, HW, coef, step, nsteps, master, round_master, cfg, guidance,
                           cfg_channels);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((ddim_update_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)pred, pred_ld, pred_c0,
                           (bf16*)lat, lat_bstride, C, B, HW, coef, step, nsteps, master, round_master, cfg, guidance,
                           cfg_channels);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_unipc_update(const void* pred, int pred_ld, int pred_c0, void* lat, int64_t lat_bstride, int C, int B,
                               int HW, const float* coef, const int* step, int nsteps, float* last, float* xmaster,
                               float* hist, int round_master, int cfg, float guidance, int cfg_channels, int dtype,
                               void* stream) {
    if (!pred || !lat || !coef || !step || !last || !xmaster || !hist || C <= 0 || B <= 0 || HW <= 0 || nsteps <= 0 ||
        pred_c0 < 0 || pred_c0 + C > pred_ld)
        return UR_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * C * HW;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (dtype == UR_DT_F16)
        hipLaunchKernelGGL((unipc_update_kernel<f16>), dim3(grid), dim3(256), 0, s, (const f16*)pred, pred_ld, pred_c0,
                           (f16*)lat, lat_bstride, C, B, HW, coef, step, nsteps, last, xmaster, hist, round_master, cfg,
                           guidance, cfg_channels);
    else if (dtype == UR_DT_BF16)
        hipLaunchKernelGGL((unipc_update_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)pred, pred_ld, pred_c0,
                           (bf16*)lat, lat_bstride, C, B, HW, coef, step, nsteps, last, xmaster, hist, round_master, cfg,
                           guidance, cfg_channels);
    else
        return UR_E_BADARG;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_sampler_advance(int* step, const float* tsteps, int nsteps, float* t_out, int B, void* stream) {
    if (!step || !tsteps || nsteps <= 0 || B < 0 || B > 256) return UR_E_BADARG;
    hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), step, tsteps,
                       nsteps, t_out, B);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// dst[k][0..bytes_k) = src[k][*step][0..bytes_k) for up to 4 tables: the rows of the CURRENT step of per-step tables a sampling
// loop's prologue computed for all its steps (the time projections of every resnet: a function of the timestep only).
struct StepRows { const char* src[4]; char* dst[4]; int64_t bytes[4]; };
__global__ void __launch_bounds__(256) select_step_rows_kernel(StepRows a, int ntab, const int* __restrict__ step, int nsteps) {
    const int k = blockIdx.y;
    if (k >= ntab) return;
    const int st = min(max(*step, 0), nsteps - 1);
    const uint4* s = reinterpret_cast<const uint4*>(a.src[k] + (int64_t)st * a.bytes[k]);
    uint4* d = reinterpret_cast<uint4*>(a.dst[k]);
    const int64_t n = a.bytes[k] >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

extern "C" int ur_select_step_rows(const void* const* src, void* const* dst, const int64_t* bytes, int ntab, const int* step,
                                   int nsteps, void* stream) {
    if (!src || !dst || !bytes || !step || ntab <= 0 || ntab > 4 || nsteps <= 0) return UR_E_BADARG;
    StepRows a{};
    int64_t most = 0;
    for (int k = 0; k < ntab; ++k) {
        if (!src[k] || !dst[k] || bytes[k] <= 0 || (bytes[k] & 15) || ((uintptr_t)src[k] & 15) || ((uintptr_t)dst[k] & 15)) return UR_E_BADARG;
        a.src[k] = reinterpret_cast<const char*>(src[k]);
        a.dst[k] = reinterpret_cast<char*>(dst[k]);
        a.bytes[k] = bytes[k];
        most = bytes[k] > most ? bytes[k] : most;
    }
    int blocks = (int)((most / 16 + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
    hipLaunchKernelGGL(select_step_rows_kernel, dim3(blocks, ntab), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, ntab, step,
                       nsteps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// One 4-byte load per 128-byte line, grid-strided; the value is consumed by an empty asm so the load is not dropped.
__global__ void __launch_bounds__(256) prefetch_kernel(const char* __restrict__ p, int64_t lines) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += (int64_t)gridDim.x * blockDim.x)
        acc ^= __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p + i * 128));
    asm volatile("" ::"v"(acc));
}

extern "C" int ur_prefetch(const void* ptr, int64_t bytes, int wgs, void* stream) {
    if (!ptr || bytes < 0 || wgs <= 0) return UR_E_BADARG;
    const int64_t lines = bytes / 128;
    if (lines == 0) return 0;
    hipLaunchKernelGGL(prefetch_kernel, dim3(wgs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const char*>(ptr), lines);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ur_abi_version(void) { return UR_ABI_VERSION; }
#define UR_STR2(x) #x
#define UR_STR(x) UR_STR2(x)
extern "C" const char* ur_build_info(void) {
    return "liburhip gfx950 (hipcc, MFMA 16x16x32 + 32x32x16, LDS-DMA) abi " UR_STR(UR_ABI_VERSION);
}
extern "C" int ur_sizeof_igemm_desc(void) { return (int)sizeof(ur_igemm_desc); }
extern "C" int ur_sizeof_attn_desc(void) { return (int)sizeof(ur_attn_desc); }
extern "C" int ur_sizeof_attn_bwd_desc(void) { return (int)sizeof(ur_attn_bwd_desc); }
