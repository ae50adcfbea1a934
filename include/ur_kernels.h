/*
 * ur_kernels.h -- C ABI of liburhip.so: the MI355X (gfx950) device ops of Uni-Renderer's
 * dual-stream denoise step.
 *
 * Boundary contract:
 *   - plain pointers and sizes only, no torch types; every pointer is a DEVICE pointer owned by the
 *     caller (PyTorch allocates activations, packed weights and workspaces);
 *   - the library keeps no state: every call is re-entrant and asynchronous on `stream`
 *     (a hipStream_t passed as void*; 0 = the null stream); no host synchronisation inside;
 *   - return value: 0 on success, a negative code on a bad descriptor (UR_E_*), or the negated
 *     hipError_t of the launch; nothing throws across the ABI.
 *   - dtype: 0 = fp16, 1 = bf16 (the storage type of activations/weights; accumulation, norm
 *     statistics and softmax are always fp32).
 *   - activations are NHWC ("channels last"): a [B,H,W,C] image is also the [B, H*W, C] token matrix.
 *
 * What each entry point replaces in the reference (all of it is stock PyTorch/diffusers CUDA
 * dispatch there -- the reference has no custom kernel on this path):
 *   ur_igemm        nn.Conv2d 3x3 / 1x1 and nn.Linear inside ResnetBlock2D, Transformer2DModel,
 *                   Attention, FeedForward(GEGLU), Downsample2D, Upsample2D(+F.interpolate nearest),
 *                   torch.cat([hidden, skip]) in front of the up-path resnets
 *                   (models/unet_2d_blocks.py:1199-1217, 2546, 2575-2588, 2677, 2696-2701), the
 *                   zero-conv feature exchange `skip + conv1x1(other_stream_skip)`
 *                   (models/controlnet.py:1752-1769, 2446-2461, 2476-2477), conv_in/conv_out
 *                   (controlnet.py:1019, 1157, 1720, 2521) and the time-embedding MLP (916, 2407).
 *   ur_groupnorm_*  nn.GroupNorm(32)+SiLU in ResnetBlock2D / conv_norm_out (controlnet.py:1154-1156)
 *                   and the GroupNorm(eps 1e-6) at Transformer2DModel entry.
 *   ur_layernorm    the three nn.LayerNorm of BasicTransformerBlock.
 *   ur_attention    F.scaled_dot_product_attention of AttnProcessor2_0 (self and 77-key cross).
 *   ur_add          `down_block_res_sample + down_block_additional_residual`, `sample + mid residual`
 *                   (controlnet.py:1078-1087, 1114-1115).
 *   ur_timestep_embedding   diffusers Timesteps (controlnet.py:285, 909-914).
 *   ur_nchw_to_nhwc / ur_nhwc_to_nchw   layout glue at the module boundary (the reference is NCHW).
 *   ur_ddim_update / ur_unipc_update / ur_sampler_advance   the per-group scheduler `.step()` calls and the timestep bookkeeping between
 *                   two denoise steps of the sampling loops (models/pipeline.py:2691-2730, 1645-1649), on the device.
 */
#ifndef UR_KERNELS_H
#define UR_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_ABI_VERSION 12

#define UR_E_BADARG (-1001)   /* inconsistent descriptor (shape / alignment / null pointer)   */
#define UR_E_UNSUPPORTED (-1002) /* shape outside what the kernels are instantiated for       */

#define UR_DT_F16 0
#define UR_DT_BF16 1
#define UR_DT_F32 2   /* accepted only where a function says so (layout glue, ur_colsum) */

#define UR_ACT_NONE 0
#define UR_ACT_SILU 1
#define UR_ACT_GEGLU 2

/* Tile configurations of ur_igemm (rows x cols of the output tile computed by one workgroup). */
#define UR_TILE_AUTO 0
#define UR_TILE_128x128 1
#define UR_TILE_128x64 2      /* 3-deep LDS ring */
#define UR_TILE_64x64 3       /* 3-deep */
#define UR_TILE_128x128_S3 4  /* 3-deep (1 workgroup per CU) */
#define UR_TILE_128x64_S2 5   /* 2-deep */
#define UR_TILE_64x64_S4 6    /* 4-deep */
#define UR_TILE_64x64_S2 7    /* 2-deep */
#define UR_TILE_256x128 8     /* 8 waves, 2-deep */
#define UR_TILE_128x320 9     /* 10 waves, 2-deep: N = 320 layers without a ragged N tile */
#define UR_TILE_128x256 10    /* 8 waves, 2-deep */
#define UR_TILE_256x256 11    /* 16 waves, 2-deep */
#define UR_TILE_64x64_R 12    /* register-staged loader (global_load -> ds_write), 2 LDS buffers */
#define UR_TILE_128x64_R 13
#define UR_TILE_128x128_R 14
#define UR_TILE_128x320_R 15
#define UR_TILE_256x128_R 16
#define UR_TILE_64x64_W1 17     /* ONE wave per workgroup: the wave owns the whole 64x64 tile (0.5 KB LDS read / MFMA) */
#define UR_TILE_128x64_W2 18    /* two waves, 64x64 each */
#define UR_TILE_64x64_W1_S3 19
#define UR_TILE_64x128_W2 20
#define UR_TILE_64x64_W1_S4 21
/* the same tiles on v_mfma_f32_32x32x16 (csrc/igemm.hip: half the MFMA instructions per chunk) */
#define UR_TILE_128x320_M32 22
#define UR_TILE_128x128_M32 23
#define UR_TILE_128x64_M32 24    /* 2-deep */
#define UR_TILE_128x64_S3_M32 25 /* 3-deep */
#define UR_TILE_64x64_M32 26     /* 2 x 2 waves of 32 x 32, 2-deep */
#define UR_TILE_64x64_S3_M32 27
#define UR_TILE_256x256_M32 28
#define UR_TILE_256x128_M32 29
#define UR_TILE_128x256_M32 30
/* wave-specialised builds: <n> extra waves only issue the global -> LDS copies of the whole tile, the others only
 * multiply (csrc/igemm.hip NL) */
#define UR_TILE_128x320_L2 31
#define UR_TILE_128x320_L4 32
#define UR_TILE_128x128_L2 33
#define UR_TILE_128x128_S3_L2 34
#define UR_TILE_128x64_L1 35
#define UR_TILE_128x64_S3_L2 36
#define UR_TILE_64x64_S3_L1 37
#define UR_TILE_256x128_L2 38
#define UR_TILE_256x256_L0 39   /* reserved: not instantiated (UR_E_UNSUPPORTED) */
#define UR_TILE_128x256_L2 40
#define UR_TILE_128x256_S3 41   /* 8 waves, 3-deep ring (144 KB): does a deeper prefetch help a big tile? */
#define UR_TILE_128x320_W8_M32 42 /* 8 waves (4 x 2, 32 x 160 each) on the 32x32x16 MFMA: two waves per SIMD, balanced */
#define UR_TILE_256x320_W16_M32 43 /* 16 waves (8 x 2, 32 x 160 each) */
#define UR_TILE_128x160_M32 44    /* 4 waves (4 x 1, 32 x 160 each), 2-deep: N = 640 / 1280 layers in 4 / 8 column tiles without split-K */
#define UR_TILE_128x160_S3_M32 45 /* the same, 3-deep */
#define UR_TILE_64x320_M32 46     /* 4 waves (2 x 2, 32 x 160 each), 2-deep */
#define UR_TILE_WS320 47          /* weight-streaming 3x3 conv, 128 pixels x 320 channels per workgroup (csrc/wsconv.hip):\
                                     `w` is the stage-image stream of tchain.py wsconv_images, channels multiples of 320 */
#define UR_TILE_WS320_W8 48       /* the same with 8 waves per workgroup = 2 per SIMD (a wave: 32 pixels x 160 channels) */
/* 8-wave PING-PONG builds (csrc/igemm_pp.hip): two groups of four waves half a phase apart -- one multiplies while the
   other reads fragments and issues LDS-DMA copies; 32-deep stages in a 4- / 5-slot LDS ring, counted vmcnt, raw barriers;
   32x32x16 MFMA.  Same descriptor, operand layouts, split-K slabs and epilogue as the lock-step tiles. */
#define UR_TILE_PP_128x320 49     /* 4 x 2 waves of 32 x 160, 5-slot ring (140 KB) */
#define UR_TILE_PP_128x320_S4 50  /* the same, 4 slots (112 KB) */
#define UR_TILE_PP_256x128 51     /* 4 x 2 waves of 64 x 64, 5 slots (120 KB) */
#define UR_TILE_PP_128x256 52     /* 2 x 4 waves of 64 x 64, 5 slots (120 KB) */
#define UR_TILE_PP_256x256 53     /* 4 x 2 waves of 64 x 128, 4 slots (128 KB) */
#define UR_TILE_PP_128x128 54     /* 4 x 2 waves of 32 x 64, 5 slots (80 KB) */
#define UR_TILE_PP_256x320 55     /* 4 x 2 waves of 64 x 160, 4 slots (144 KB) */
/* round 6: FEW waves with BIG per-wave tiles on the lock-step loop (one wave per SIMD): a 64 x 160 wave tile reads 0.044 B of LDS
 * fragments per MAC against 0.0625 for the 64 x 64 wave tile of UR_TILE_128x320 -- the conv loop's LDS port (DMA writes +
 * fragment reads: ~1700 of ~1300 MFMA cycles per chunk) is what these go after */
#define UR_TILE_256x160_W4_M32 56 /* 4 waves (4 x 1, 64 x 160 each), 2-deep, 104 KB */
#define UR_TILE_256x320_W8_M32 57 /* 8 waves (4 x 2, 64 x 160 each), 2-deep, 144 KB */
#define UR_TILE_128x320_W4_M32 58 /* 4 waves (2 x 2, 64 x 160 each), 2-deep, 112 KB */
#define UR_TILE_256x128_W4_M32 59 /* 4 waves (4 x 1, 64 x 128 each), 2-deep, 96 KB */
#define UR_TILE_256x256_W8_M32 60 /* 8 waves (4 x 2, 64 x 128 each), 2-deep, 128 KB */
#define UR_TILE_256x320_W10 61    /* 10 waves (2 x 5, 128 x 64 each) on the 16x16x32 MFMA, 2-deep, 144 KB */
#define UR_TILE_COUNT 62

/*
 * Implicit GEMM:  out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )
 *
 *   X  (M rows)  is either a plain row-major matrix (taps == 1) or the im2col view of an NHWC image
 *      gathered on the fly (taps == 9: 3x3, pad 1, stride 1|2, optional nearest-2x upsample in
 *      front), optionally the channel-concatenation of two NHWC sources (x0 | x1).
 *      K = taps * (c0 + c1), k = tap * (c0 + c1) + c.  c0, c1 must be multiples of 64.
 *   W  (N rows)  is row-major [N][K] (K contiguous).
 *   epilogue: +bias[n] (fp32), +rowadd[m / rows_per_b][n], activation, +res[m][n], *out_scale.
 *      act == UR_ACT_GEGLU: packed columns come in groups of 8 = 4 value + 4 gate columns and the
 *      output has N/2 columns: out[m][(p/8)*4 + p%4] = value * gelu(gate).
 *   zbatch > 1: grid.z batches independent problems of one shape; every operand has its own per-z element
 *      stride (zx, zx1, zw, zbias, zrow, zres, zout).  Used for the transposed V projection
 *      Vt[b] = Wv . X[b]^T and for executing the two diffusion streams as ONE grouped launch.
 *   (hi, lo) residual stream: the tensors that carry `x + f(x)` through a network (block outputs) may be held as
 *      an fp16/bf16 PAIR whose fp32 sum is the value: `out` keeps the ordinary rounded value (what every GEMM operand
 *      consumer reads), `out_lo` the rounding remainder; residual adds (`res` + `res_lo`) and the norm kernels
 *      (`*_lo` arguments) consume the pair.  This removes the random walk of the storage rounding along the residual
 *      path (55 % of the fp16 error variance of the step).  Storage of a low part: for fp16
 *      streams ONE byte per element, e5m2 = the high byte of the fp16 encoding of the remainder, rounded to nearest
 *      even (|lo| <= ulp(hi)/2, so three significant bits put the pair at 2^-14 relative); for bf16 streams a bf16.
 *      Every `*_lo` pointer below is in that format, with the element strides of its hi tensor.
 *   splitk > 1: K is additionally split over grid.z; `partial` is a caller-provided fp32 workspace of
 *      zbatch * splitk * M * ldp floats and the epilogue runs in a second small kernel.
 */
typedef struct ur_igemm_desc {
    const void* x0;
    const void* x1;      /* second concat source or NULL                                   */
    const void* w;
    const float* bias;   /* [N] fp32 or NULL                                               */
    const void* rowadd;  /* [*, ld_rowadd] dtype or NULL; row index = m / rows_per_b        */
    const void* res;     /* [M][ldres] dtype or NULL                                       */
    void* out;           /* [M][ldc] dtype                                                 */
    float* partial;      /* split-K workspace or NULL                                      */
    const void* zero_page; /* >= 256 zero bytes, 16-byte aligned (read for padding rows)     */
    int64_t ldx0, ldx1;  /* element stride between consecutive pixels/rows of x0 / x1      */
    int64_t ldw;         /* element stride between rows of W                               */
    int64_t ldres, ldc;
    int64_t zx, zw, zout; /* per-z element strides (zbatch > 1) of x0, w, out               */
    int64_t zx1, zbias, zrow, zres; /* per-z element strides of x1, bias, rowadd, res (may be negative) */
    int64_t ldp;         /* row stride (floats) of `partial`, multiple of 64               */
    int32_t c0, c1;
    int32_t B, Hin, Win, Hout, Wout; /* conv geometry (taps == 9); ignored for taps == 1   */
    int32_t taps, stride, ups;
    int32_t M, N, K;
    int32_t n_store;     /* columns written per row (>= N writes zeros), <= ldc            */
    int32_t ld_rowadd, rows_per_b;
    int32_t act;
    float out_scale;
    int32_t zbatch, splitk;
    int32_t zx_div;      /* x0 of problem z starts at x0 + (z / zx_div) * zx (0/1 = every z)   */
    int32_t tile;        /* UR_TILE_*                                                      */
    int32_t dtype;
    /* (hi, lo) residual stream, see below: both optional, same leading dimension / z stride as res / out */
    const void* res_lo;  /* low part of the residual: the epilogue adds res + res_lo in fp32 */
    void* out_lo;        /* if set, receives dtype(v - float(dtype(v))) of every stored value v */
    /* K order of a 3x3 conv.  0: k = tap * (c0 + c1) + c (tap outer).  > 0 (multiple of 64 dividing c0; one source,
     * c1 == 0): channel blocks outer, k = (c / cblock) * 9 * cblock + tap * cblock + c % cblock -- a workgroup
     * re-reads an input line after cblock / 64 chunks instead of c0 / 64, which keeps the nine taps of a wide input
     * (c0 = 640 .. 2560) in the XCD's L2.  The weight matrix must be packed in the same order. */
    int32_t cblock;
    /* 1x1 tail of a 3x3 conv (taps == 9, stride 1, no upsampling): after the 9 * (c0 + c1) tap columns K continues with
     * ct0 channels of t0 and ct1 channels of t1, both read at the OUTPUT pixel (NHWC, Hout x Wout, leading dimensions
     * ldt0 / ldt1, per-z strides zt0 / zt1): out += Wtail . [t0 | t1].  This is how a ResnetBlock2D's 1x1 conv_shortcut
     * over its (possibly concatenated) input rides in the K loop of conv2 instead of being a launch of its own.
     * K = 9 * (c0 + c1) + ct0 + ct1; ct0, ct1 multiples of 64; ct0 == 0: no tail. */
    const void* t0;
    const void* t1;
    int64_t ldt0, ldt1, zt0, zt1;
    int32_t ct0, ct1;
    /* taps == 9: zero padding on the top / left edge, 1 or 0 (the bottom / right edge is whatever Hout / Wout imply:
     * taps beyond the image read zeros).  1 = the symmetric `padding=1` of every conv of the UNets; 0 with stride 2 =
     * the VAE encoder's Downsample2D, F.pad(x, (0, 1, 0, 1)) followed by a stride-2 conv with padding 0. */
    int32_t pad;
    /* Transposed side output (ABI 8): with out_vt set, the columns n >= vt_n0 are NOT written to `out`; they leave, after
     * +bias only (no activation, residual or out_scale), as
     *     out_vt[z * zvt + (m / vt_rows) * vt_bstride + (n - vt_n0) * ldvt + m % vt_rows]
     * i.e. as the matrix V^T[sample][channel][token] the attention kernels read, so that a self-attention's q | k | v
     * projection is ONE launch instead of a q | k GEMM plus a transposed V GEMM (models/unet_2d_blocks.py:1115-1126: the
     * Attention of every Transformer2DModel).  vt_n0 and N - vt_n0 are multiples of 16, n_store <= vt_n0, act == NONE and
     * res == NULL; the caller zero-fills token columns >= vt_rows of a padded V^T itself. */
    void* out_vt;
    int64_t ldvt, vt_bstride, zvt;
    int32_t vt_n0, vt_rows;
    /* size of the zero region behind zero_page in bytes (a power of two >= 256; 0 = 256).  Padding rows are read from it; with
     * a large region every (workgroup, wave) reads its own 128-byte line instead of all 256 CUs hammering one line of one L2
     * channel (round 4: the conv kernels read 2 - 8 padding rows per K step). */
    int32_t zero_page_bytes;
} ur_igemm_desc;

int ur_igemm(const ur_igemm_desc* d, void* stream);

/* 1 when ur_igemm runs this 3x3 conv (tile resolved, not UR_TILE_AUTO) on the kernel whose three dx taps share one staged
 * pixel block (csrc/igemm_dxs.hip: stride 1, pad 1, one source, W_out a power of two >= 8 dividing the tile height) -- only
 * in a library built with `make DXS=1` AND with UR_DXS=1 in the environment (measured slower in the step: an experiment,
 * not the product path); 0 when it stays on the lock-step kernel (always, in the product build). */
int ur_igemm_uses_dxs(const ur_igemm_desc* d);

/* A split-K 3x3 conv whose ONLY consumer is a GroupNorm (+ SiLU): main pass of `d` (splitk > 1, no residual / activation /
 * low part; bias and rowadd allowed), then ONE second pass per (z, sample, group) that sums the fp32 slabs, adds bias and the
 * per-sample row, rounds to the storage dtype, normalises over the group (eps, gamma / beta [N] fp32, + z * zgn for problem
 * z) and writes only the normalised tensor to d->out -- the conv output is never materialised.  The conv1 -> norm2 -> SiLU
 * hand-off of a ResnetBlock2D (models/unet_2d_blocks.py:1100-1111) at the 16x16 / 8x8 levels; replaces the split-K reduce
 * launch + ur_groupnorm_fused.  Needs (N / groups) % 4 == 0 and Hout * Wout * (N / groups) <= 16384, else UR_E_UNSUPPORTED. */
int ur_igemm_splitk_gn(const ur_igemm_desc* d, const float* gamma, const float* beta, int64_t zgn, float eps, int groups,
                       int silu, void* stream);

/* 1 when the library was built with the experimental weight-streaming conv tiles (UR_TILE_WS320*: `make WSCONV=1`); the
 * product build returns 0 and UR_E_UNSUPPORTED for those tile ids. */
int ur_has_wsconv(void);
int ur_has_pp(void);     /* 1: the ping-pong tiles (UR_TILE_PP_*) are built in (`make PP=1`); else they return UR_E_UNSUPPORTED */

/* Workspace (in floats) ur_igemm needs in `partial` for this descriptor (0 when splitk <= 1). */
int64_t ur_igemm_partial_floats(const ur_igemm_desc* d);

/*
 * GroupNorm over NHWC, optionally over the concatenation of two sources (x0 | x1), optional SiLU.
 *   stats:  partial[b][chunk][g][2] = (sum, sumsq) over the rows of that chunk, fp32.
 *   apply:  y = (x - mean) * rstd * gamma + beta  (-> SiLU), mean/rstd reduced (fixed order) from the
 *           `nstat` chunks of `partial` written by the stats pass.
 * rows = H*W per sample; nchunks = number of row chunks per sample (grid.x) of the call at hand.
 * bper > 0: sample b uses gamma/beta + (b / bper) * pstride (grouped execution of several streams).
 * x0_lo / x1_lo: NULL, or the low parts of (hi, lo) residual-stream inputs (same layout as x0 / x1): the value
 * normalised is x + x_lo in fp32.
 */
int ur_groupnorm_stats(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1, int B,
                       int rows, int groups, int nchunks, float* partial, int dtype, void* stream);
int ur_groupnorm_apply(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1, int B,
                       int rows, int groups, int nstat, int nchunks, const float* partial, const float* gamma,
                       const float* beta, float eps, int silu, int bper, int pstride, void* out, int dtype,
                       void* stream);

/* The same GroupNorm in ONE launch (one workgroup per (sample, group); statistics over the hi parts, normalisation of
 * hi + lo): for the maps of the deep levels, where stats + apply are launch-bound.  Group width (c0 + c1) / groups must be
 * even and <= 128 (UR_E_UNSUPPORTED otherwise).  Strips of at most 4 (16-byte pieces) / 8 (8- / 4-byte pieces) pieces per
 * thread are loaded once and stay in registers across the block reduction (round 6: one memory round trip); larger ones take
 * two sweeps, the second out of L2.  `silu`: bit 0 = SiLU after the affine map; bit 1 (UR_GN_TWO_SWEEP) = always the two-sweep
 * kernel (A/B runs, tests: the two kernels produce the same bits). */
#define UR_GN_TWO_SWEEP 2
int ur_groupnorm_fused(const void* x0, const void* x1, const void* x0_lo, const void* x1_lo, int c0, int c1, int B,
                       int rows, int groups, const float* gamma, const float* beta, float eps, int silu, int bper,
                       int pstride, void* out, int dtype, void* stream);

/* LayerNorm over the last dimension of x[rows][C] (C % 8 == 0, C <= 4096), fp32 statistics.
 * rows_per_set > 0: row r uses gamma/beta + (r / rows_per_set) * pstride.  x_lo: NULL or the low part of x. */
int ur_layernorm(const void* x, const void* x_lo, const float* gamma, const float* beta, float eps, int rows, int C,
                 int rows_per_set, int pstride, void* out, int dtype, void* stream);

/*
 * softmax(Q K^T * scale) V for all (batch, head) pairs.
 *   q  [B][Tq][ldq]   head h at columns q_off + h*d .. +d
 *   k  [B][Tk][ldk]   head h at columns k_off + h*d .. +d
 *   vt [B][H*d][ldvt] TRANSPOSED values: row h*d + j holds V[:, j] of head h over the keys;
 *                     ldvt is a multiple of 64 and columns >= Tk are zero; batch b starts at
 *                     vt + b * vt_bstride elements (lets several layers share one batched projection)
 *   o  [B][Tq][ldo]   head h at columns h*d .. +d
 * d in {32, 40, 64, 80, 128, 160}.
 * q_hstride / k_hstride (ABI 12; d <= 64 only, UR_E_UNSUPPORTED otherwise): 0 = the column-block image above.  > 0 = HEAD-MAJOR
 *   image [B][H][T][d]: head h of sample b starts at q + q_off + b * Tq * ldq + h * q_hstride (so ldq = H * d still sizes a
 *   sample and q_hstride = Tq * d) and its rows are d elements apart.  What ur_tchain's qk_heads mode writes: a head's keys
 *   are one contiguous run instead of 2d-byte slices of (2 H d)-byte token rows, which at H = 8, d = 40 cost 2.4 cache lines
 *   fetched per line used (profiles/r06_pmc_attn_l2_cfg5.json).
 * scale > 0: the usual softmax scale (the reference passes d^-1/2, AttnProcessor2_0).
 * scale <= 0: Q.K^T is ALREADY in log2 units, i.e. the caller folded scale*log2(e) into the q / k projections
 *             (ur_igemm out_scale, applied in fp32 before the single rounding); p = exp2(q.k - max).
 */
typedef struct ur_attn_desc {
    const void* q;
    const void* k;
    const void* vt;
    void* o;
    const void* zero_page;
    int64_t ldq, ldk, ldvt, ldo;
    int64_t vt_bstride;
    int64_t q_hstride, k_hstride;  /* 0, or the head stride of a head-major q / k image (see above) */
    int32_t q_off, k_off;
    int32_t B, H, Tq, Tk, d;
    float scale;  /* see above: <= 0 selects "scores already in log2 units" */
    int32_t dtype;
    float* lse;   /* NULL, or [B*H][Tq] fp32: the row log-sum-exp of the scaled scores in log2 units (what
                   * ur_attention_backward takes with has_lse = 1); needs scale > 0 */
} ur_attn_desc;

int ur_attention(const ur_attn_desc* d, void* stream);

/* out = a + b * alpha (elementwise, n % 8 == 0). */
int ur_add(const void* a, const void* b, float alpha, void* out, int64_t n, int dtype, void* stream);
/* the same over (hi, lo) pairs: out + out_lo = (a + a_lo) + alpha * (b + b_lo); any *_lo may be NULL */
int ur_add_hilo(const void* a, const void* a_lo, const void* b, const void* b_lo, float alpha, void* out, void* out_lo,
                int64_t n, int dtype, void* stream);

/* ur_add_hilo (alpha = 1) over up to UR_ADD_MULTI_MAX independent tensor triples in ONE launch (ABI 10).  What a sampling
 * loop with a loop-invariant exchange operand runs per step instead of 13 exchange GEMMs: in the inverse-rendering loop
 * (models/pipeline.py:2629-2690) the UNet's skips are constant, so `control_down_blocks[i](skip_unet[i])`
 * (models/controlnet.py:2446-2461, 2476-2477) is computed once per call and only the add remains per step; in the
 * rendering loop (pipeline.py:1587-1629) the same holds for the encoder's `controlnet_down_blocks[i](skip_enc[i])`
 * (controlnet.py:1752-1769, added at 1078-1087, 1114-1115). */
#define UR_ADD_MULTI_MAX 16
typedef struct ur_add_item {
    const void* a;
    const void* a_lo;   /* NULL: a has no low part */
    const void* b;
    const void* b_lo;
    void* out;
    void* out_lo;       /* NULL: the rounding remainder is dropped */
    int64_t n;          /* elements, multiple of 8 */
} ur_add_item;
int ur_add_hilo_multi(const ur_add_item* items, int n, int dtype, void* stream);
int ur_sizeof_add_item(void);

/* Sinusoidal timestep embedding of nt (1 or B) fp32 timesteps: out[b][:] = [cos | sin] (flip) or
 * [sin | cos]; fp32 math. */
int ur_timestep_embedding(const float* t, int nt, int B, int dim, int flip_sin_to_cos, float freq_shift,
                          void* out, int dtype, void* stream);

/* Layout glue.  src_dtype/dst dtype: 0 f16, 1 bf16, 2 f32.  Channels >= C of the padded NHWC output
 * are written as zeros. */
/* out[b][oy][ox][:] = in[b][sy][sx][:], PyTorch 'nearest': s = min(floor(o * (float)in / out), in - 1); NHWC, C % 8 == 0.
 * The general F.interpolate(size=...) of Upsample2D (controlnet.py:1129-1130: latent side not a multiple of 8); the
 * exact 2x case is fused into the conv gather (ur_igemm, ups = 1) instead. */
int ur_resize_nearest(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, int C, int dtype,
                      void* stream);
int ur_nchw_to_nhwc(const void* src, int src_dtype, int B, int C, int H, int W, void* dst, int Cpad, int dtype,
                    void* stream);
int ur_nhwc_to_nchw(const void* src, int dtype, int B, int C, int H, int W, void* dst, int dst_dtype,
                    void* stream);

/*
 * Sampler glue between two denoise steps, graph-capturable: no host value is baked in.
 * ur_ddim_update: DDIM (eta 0) update for an x0-predicting model over C latent channels,
 *     eps = (x - c[0]*x0) / c[1];  x <- c[2]*x0 + c[3]*eps,   c = coef + 4 * min(*step, nsteps-1)
 *     (c = sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)), evaluated in fp32 without FMA contraction.
 *     pred: NHWC [B][HW][pred_ld], channels pred_c0 .. pred_c0+C (a network output);
 *     lat:  NCHW [B][C][HW], sample b at lat + b*lat_bstride (a channel slice of the next step's input), in place.
 *     master: NULL, or the sampler's own contiguous fp32 [B][C][HW] copy of the latents: then x is read from it,
 *     the result is stored there (rounded through the storage dtype first if round_master) and its rounding in lat.
 *     cfg != 0: classifier-free guidance (pipeline.py:2695-2721, 1642-1644): pred and lat hold 2B samples (cond
 *     [0,B), uncond [B,2B)); channels c < cfg_channels use x0 = p_uncond + guidance * (p_cond - p_uncond) evaluated in
 *     the prediction's dtype like the reference, the others p_cond; the new latent is written to both halves of lat.
 * ur_sampler_advance: *step += 1; t_out[0..B) = tsteps[min(*step, nsteps-1)] (t_out may be NULL; B <= 256).
 */
int ur_ddim_update(const void* pred, int pred_ld, int pred_c0, void* lat, int64_t lat_bstride, int C, int B, int HW,
                   const float* coef, const int* step, int nsteps, float* master, int round_master, int cfg,
                   float guidance, int cfg_channels, int dtype, void* stream);
int ur_sampler_advance(int* step, const float* tsteps, int nsteps, float* t_out, int B, void* stream);
/*
 * ur_select_step_rows (round 6): dst[k][0 .. bytes[k]) = src[k] + clamp(*step, 0, nsteps-1) * bytes[k] for ntab <= 4 tables, one
 * launch.  The sampling loops (models/pipeline.py:2629-2730, 1587-1653) evaluate the time embedding and every resnet's
 * `time_emb_proj(SiLU(emb))` (unet_2d_blocks.py:1100-1111) on every step although they depend on the timestep only: the
 * hoisted loops compute them for ALL steps once per call and each step only picks its rows with the device-side step
 * counter.  Pointers and byte counts 16-byte aligned.
 */
int ur_select_step_rows(const void* const* src, void* const* dst, const int64_t* bytes, int ntab, const int* step, int nsteps,
                        void* stream);

/*
 * ur_unipc_update: the UniPCMultistepScheduler.step() calls of the live sampling loops (eval/test_real.py:485-492
 * attaches eight UniPC schedulers; models/pipeline.py:2725-2730, 1649) for all latent groups at once.  UniPC with data
 * (x0) prediction, B(h) = bh2, order <= 2 is a linear recurrence with data-independent scalars (host:
 * schedulers.UniPCMultistepScheduler.coefficient_table, row i = c0 c1 c2 c3 p0 p1 p2 -):
 *     L_i = c0 L_{i-1} + c1 m_{i-1} + c2 m_{i-2} + c3 m_i   (corrector; L_0 = the initial sample)
 *     x_{i+1} = p0 L_i + p1 m_i + p2 m_{i-1}                (predictor = next model input, written to `lat`)
 * with m_i = pred at step i = *step.  pred / lat / cfg / guidance / round_master as ur_ddim_update; last, xmaster:
 * fp32 [B][C][HW]; hist: fp32 [2][B][C][HW], zeroed by the caller before step 0.  The caller advances *step with
 * ur_sampler_advance.
 */
int ur_unipc_update(const void* pred, int pred_ld, int pred_c0, void* lat, int64_t lat_bstride, int C, int B, int HW,
                    const float* coef, const int* step, int nsteps, float* last, float* xmaster, float* hist,
                    int round_master, int cfg, float guidance, int cfg_channels, int dtype, void* stream);

/*
 * Weight prefetch into the 256 MB Infinity Cache (MI355X's memory-side last-level cache).  One denoise step reads
 * 3.5 GB of weights that are all cold (the cache holds < 8 % of them); the deep-level launches are then paced by HBM
 * latency, while HBM sits idle under the MFMA-bound 64x64-level launches.  This kernel touches `bytes` bytes at `ptr`
 * (one 4-byte non-temporal load per 128-byte line, results discarded) from `wgs` small workgroups, so that a caller can
 * run it on a second stream / graph branch one layer AHEAD of the layer that will read those weights.  Speed only:
 * no effect on results.  No reference counterpart (the reference has no custom kernels).
 */
int ur_prefetch(const void* ptr, int64_t bytes, int wgs, void* stream);

/*
 * Backward building blocks (device op 11; csrc/backward.hip).  The GEMM-shaped gradients run on
 * ur_igemm over transposed operands (dX = dY.W: x0 = dY, w = W^T; dW = dY^T.X: x0 = dY^T, w = X^T; conv dX = conv3x3
 * of dY with rotated weights; conv dW: x0 = dY^T, w = im2col(X)^T); these entry points are what that needs around
 * the GEMM plus the backward of the memory-bound ops.  Deterministic (fixed-order) reductions, fp32 sums.
 *
 * ur_transpose2d   dst[b][c][r] = src[b][r][c]; C, leading dims and batch strides multiples of 8; columns R .. ceil8(R)
 *                  of dst are written as zeros (ld_dst >= ceil8(R)).
 * ur_im2col3x3_t   out[(tap*C + c)][p] = x[pixel(p, tap)][c] of a 3x3 / pad 1 / stride 1|2 conv over NHWC x
 *                  (p = (b, oy, ox) row-major, ld_out >= P, columns P .. ld_out written as zeros).
 * ur_colsum        out[g][n] = sum of x[m][n] over the rows of group g (rows_per_group rows each; 0 = one group);
 *                  fp32 out; dtype may be UR_DT_F32.  Tall inputs are summed in row slices into ``workspace``
 *                  (ur_colsum_workspace_floats floats; may be null when that is 0) and folded in a fixed order.
 * ur_silu_backward dx = dy * d/dx(x * sigmoid(x)).
 * ur_geglu_forward / ur_geglu_backward   the reference's GEGLU layout h = [value | gate] (2D columns):
 *                  y = value * gelu(gate) (erf);  dh = [dy * gelu(gate) | dy * value * gelu'(gate)].
 * ur_groupnorm_backward   dx of y = act(GN(x) * gamma + beta) (act = SiLU if silu) from the forward statistics
 *                  partials (ur_groupnorm_stats, nstat chunks).  Three launches: per-channel partial (sum dz, sum dz*xhat)
 *                  over nred row chunks into chan_part[b][nred][C][2] (fp32 workspace), their fixed-order fold into
 *                  chan_sum[b][C][2] (fp32 output: summed over b these are dbeta and dgamma), and dx over nchunks
 *                  row chunks per sample.
 * ur_layernorm_backward   dx per row; part[wave][2][C] (fp32, waves = ceil(rows / rows_per_wave)) receives the
 *                  per-wave partial (dgamma, dbeta); C <= 2048.
 */
int ur_transpose2d(const void* src, int64_t ld_src, int64_t bs_src, void* dst, int64_t ld_dst, int64_t bs_dst, int R,
                   int C, int batch, int dtype, void* stream);
int ur_im2col3x3_t(const void* x, int B, int H, int W, int C, int stride, void* out, int64_t ld_out, int dtype,
                   void* stream);
/* out[k][c] = sum over b < B of in[b][c][k], k = 0, 1 (fp32): the per-sample (sum dz, sum dz * xhat) pairs of the GroupNorm
 * backward reduced over the batch straight into two contiguous parameter-gradient rows (dbeta = out[0], dgamma = out[1]). */
int ur_pairsum_rows(const float* in, int B, int C, float* out, void* stream);
int64_t ur_colsum_workspace_floats(int M, int N, int rows_per_group);
/* One-launch form (ABI 7): `counters` = ur_colsum_counters() zero-initialised 32-bit device counters, left at zero; the
 * last workgroup of a (group, column block) adds the slice sums in the order of the two-launch form (identical bits). */
int ur_colsum_counters(int M, int N, int rows_per_group);
int ur_colsum_fused(const void* x, int64_t ldx, int M, int N, int rows_per_group, float* out, float* workspace,
                    uint32_t* counters, int dtype, void* stream);
int ur_colsum(const void* x, int64_t ldx, int M, int N, int rows_per_group, float* out, float* workspace, int dtype,
              void* stream);
int ur_silu_backward(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);
int ur_geglu_forward(const void* h, void* y, int64_t M, int D, int dtype, void* stream);
int ur_geglu_backward(const void* h, const void* dy, void* dh, int64_t M, int D, int dtype, void* stream);
int ur_groupnorm_backward(const void* x, const void* dy, int C, int B, int rows, int groups, int nstat,
                          const float* partial, const float* gamma, const float* beta, float eps, int silu, int nred,
                          float* chan_part, float* chan_sum, int nchunks, void* dx, int dtype, void* stream);
/* The same gradient in ONE launch for small maps (one 1024-thread workgroup per (sample, group); no forward statistics, no
 * workspace): dx and chan_sum[b][C][2] as ur_groupnorm_backward leaves them.  Group width C / groups even and <= 128
 * (UR_E_UNSUPPORTED otherwise); meant for <= ~1024 pixels per sample (the 32x32 / 16x16 / 8x8 levels). */
int ur_groupnorm_backward_fused(const void* x, const void* dy, int C, int B, int rows, int groups, const float* gamma,
                                const float* beta, float eps, int silu, float* chan_sum, void* dx, int dtype, void* stream);
/* ... with `skip` (NULL or [rows][C] dtype): dx += skip -- the gradient of a residual connection around the norm (ABI 7) */
int ur_layernorm_backward_skip(const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                               int rows_per_wave, void* dx, float* part, const void* skip, int dtype, void* stream);
int ur_layernorm_backward(const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                          int rows_per_wave, void* dx, float* part, int dtype, void* stream);
/* Up to UR_TRANSPOSE_MAX independent batched transposes (each as ur_transpose2d: dst[b][c][r] = src[b][r][c], rows
 * R .. ceil8(R) -- or rows_out -- of the source read as zeros) in one launch; same dtype for all. */
#define UR_TRANSPOSE_MAX 32
typedef struct ur_transpose_desc {
    const void* src;
    void* dst;
    int64_t ld_src, bs_src, ld_dst, bs_dst;
    int32_t R, C, batch;
    int32_t rows_out;  /* 0: ceil8(R) destination columns are written (zeros past R); else a multiple of 8 in
                          [R, ceil64(R)]: the zero padding a following GEMM needs on its contraction dimension */
    /* optional fused column sums of the source (batch == 1): colsum[c] = sum_r src[r][c] in fp32, deterministic.
     * colsum_ws: ceil(R / 64) * C floats of scratch; colsum_cnt: ceil(C / 64) zero-initialised counters that the launch
     * leaves zero again (one buffer can serve every call on a stream). */
    float* colsum;
    float* colsum_ws;
    uint32_t* colsum_cnt;
} ur_transpose_desc;
int ur_transpose2d_multi(const ur_transpose_desc* descs, int n, int dtype, void* stream);

/*
 * Weight gradient of a linear layer or of a 3x3 conv WITHOUT transposed copies of its operands (ABI 9):
 *     dw[n][k] = sum over p < P of dy[p][n] * xcol[p][k],      db[n] = sum over p of dy[p][n]
 * where xcol[p][k] = x[p][k] (taps == 1) or, for taps == 9, the im2col row of output pixel p = (b, oy, ox):
 * k = tap * C + c -> x[b][oy * stride - pad + tap / 3][ox * stride - pad + tap % 3][c], zero outside the image -- the
 * packed weight layout [N][(ky, kx, c)] of ur_igemm.  Both operands are read as they lie in memory (rows = pixels / tokens,
 * the contraction index p is the SLOW dimension of both): the kernel stages [p][n] / [p][k] tiles with LDS-DMA and feeds the
 * MFMAs through the LDS transpose read (ds_read_b64_tr_b16).  Replaces ur_transpose2d_multi of dy and x (+ their zero
 * padding to 64), ur_im2col3x3_t and the ur_colsum of dy in the backward of train/train.py:1416 (autograd of F.linear /
 * F.conv2d, the weight / bias gradients).
 *   N % 8 == 0, K % 8 == 0 (taps == 9: C % 64 == 0, K = 9 * C, Hout and Wout powers of two); P arbitrary.
 *   splits > 1: the P range is cut into `splits` slices (each a multiple of 32 rows), fp32 slabs in `partial`
 *   (ur_wgrad_plan gives the slice count the library would choose and the floats needed), summed in slice order by a
 *   second launch -- deterministic.  db may be NULL.  dw is written in `dtype`.
 */
typedef struct ur_wgrad_desc {
    const void* dy;       /* [P][lddy]                                                              */
    const void* x;        /* taps == 1: [P][ldx]; taps == 9: NHWC [B][Hin][Win] pixels of ldx elements */
    void* dw;             /* [N][lddw] dtype                                                        */
    float* db;            /* [N] fp32 or NULL                                                       */
    float* partial;       /* splits > 1: ur_wgrad_plan's float count                                */
    const void* zero_page; /* zero region (rows >= P, taps outside the image), 16-byte aligned       */
    int64_t lddy, ldx, lddw;
    int32_t P, N, K;
    int32_t C, B, Hin, Win, Hout, Wout; /* taps == 9 only                                           */
    int32_t taps, stride, pad;
    int32_t splits;       /* >= 1                                                                   */
    int32_t tile;         /* 0: library's choice; (k) x (n) = 1: 128 x 128; 2: 128 x 64; 3: 64 x 64; 4: 256 x 256; 5: 256 x 128; 6: 128 x 256 */
    int32_t zero_page_bytes; /* as ur_igemm_desc.zero_page_bytes                                    */
    int32_t dtype;
} ur_wgrad_desc;
int ur_wgrad(const ur_wgrad_desc* d, void* stream);
/* n <= UR_WGRAD_GROUP_MAX problems of ONE shape (everything in `d` except dy / x / dw / db, which come from g[i]) in one launch
 * (+ one reduce launch when d->splits > 1): the weight gradients of equally shaped layers -- q | k | v, out and feed-forward
 * projections of the transformer blocks of a UNet level -- fill the chip together instead of one small GEMM after the other,
 * and need fewer (or no) slices.  `partial`: n times the floats of one problem (ur_wgrad_group_plan).  g[i].db may be NULL
 * per problem. */
#define UR_WGRAD_GROUP_MAX 64
typedef struct ur_wgrad_ptrs {
    const void* dy;
    const void* x;
    void* dw;
    float* db;
} ur_wgrad_ptrs;
int ur_wgrad_group(const ur_wgrad_desc* d, const ur_wgrad_ptrs* g, int n, void* stream);
int ur_wgrad_group_plan(const ur_wgrad_desc* d, const ur_wgrad_ptrs* g, int n, int32_t* splits, int64_t* partial_floats);
/* the slice count the library would use for this problem (d->splits ignored) and the floats `partial` needs for it */
int ur_wgrad_plan(const ur_wgrad_desc* d, int32_t* splits, int64_t* partial_floats);
/* floats `partial` needs for d->splits as given (0 when splits <= 1) */
int64_t ur_wgrad_partial_floats(const ur_wgrad_desc* d);

/* dst[i] = (dst type) src[i] for up to UR_CAST_MAX_TENSORS contiguous tensors in one launch: fp32 -> dtype (to_f32 = 0: the
 * master parameters into the compute dtype) or dtype -> fp32 (to_f32 = 1: their gradients back).  dtype: UR_DT_F16 / BF16. */
#define UR_CAST_MAX_TENSORS 128
typedef struct ur_cast_tensor {
    const void* src;
    void* dst;
    int64_t n;
} ur_cast_tensor;
int ur_cast_multi(const ur_cast_tensor* tensors, int n_tensors, int to_f32, int dtype, void* stream);
/* The same (to_f32 = 1 only) that also leaves, per workgroup, the sum of squares of the fp32 values it wrote in
 * sumsq[0 .. ur_cast_multi_blocks()) -- fixed order inside a workgroup, no atomics: the gradient norm of train/train.py:1422
 * (clip_grad_norm_) without a second sweep over the gradients (ABI 7). */
int64_t ur_cast_multi_blocks(const ur_cast_tensor* tensors, int n_tensors);
int ur_cast_multi_sumsq(const ur_cast_tensor* tensors, int n_tensors, int to_f32, int dtype, float* sumsq, void* stream);

/* AdamW over many parameter tensors per launch (the optimizer step of the training loop, train/train.py:1082-1100
 * torch.optim.AdamW, 1425 optimizer.step()).  Decoupled weight decay, bias correction, no amsgrad, fp32 everywhere; the
 * arithmetic of torch's fused kernel (param -= lr*wd*param; exp_avg = lerp(exp_avg, g, 1-beta1); exp_avg_sq = beta2*
 * exp_avg_sq + (1-beta2) g^2; param -= lr/bc1 * exp_avg / (sqrt(exp_avg_sq)/sqrt(bc2) + eps)).  Descriptors travel as
 * kernel arguments (no device table: pointers may change from step to step, and a captured graph keeps them by value).
 *   step        device scalar, the step count of THIS update (>= 1)
 *   grad_scale  device scalar or NULL: every gradient is divided by it (gradient clipping folded into the update)
 *   found_inf   device scalar or NULL: != 0 skips the update
 * Tensors whose four pointers are 16-byte aligned take the float4 path. */
#define UR_ADAMW_MAX_TENSORS 64
typedef struct ur_adamw_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} ur_adamw_tensor;
/* hyper (ABI 8): NULL, or a device pointer to {lr, weight_decay} read by the kernel INSTEAD of the by-value arguments, so
 * that a captured HIP graph of the update follows a learning-rate schedule (train.py --lr_scheduler) without re-capture. */
int ur_adamw_multi(const ur_adamw_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const float* step, const float* grad_scale, const float* found_inf,
                   const float* hyper, void* stream);

/* Flash backward of o = softmax(q k^T * scale) v (ur_attention_backward_supported: Tq % 64 == 0, d % 8 == 0, d <= 160).
 * Replaces the reference's autograd through F.scaled_dot_product_attention (diffusers AttnProcessor2_0 under
 * models/attention.py BasicTransformerBlock) in the training step, for the self-attention and the 77-key
 * cross-attention.  P is never materialised; two launches (dq, then dk / dv; a third that folds the query splits when
 * there are few keys), every sum in a fixed order.  All matrices are the reference's token matrices [B][T][ld] with head
 * h at columns h*d .. h*d+d-1 of the pointer handed in (q | k | v parts of a fused projection = column offsets):
 *   q, o, dout  [B][Tq][ld*]     k, v  [B][Tk][ld*]          dq | dk, dv: same layouts as q | k, v
 *   qt, dot, kt ABI <= 8: transposes of q, dout, k ([B][H*d][ld]).  ABI 9: NOT READ any more (may be NULL, ld* ignored): the
 *               kernels gather the transposed MFMA fragments from the row-major tiles with the LDS transpose read;
 *               ur_attention_backward_needs_transposes() is 0 (1 only in a library built with -DUR_ATTN_BWD_TRN=1)
 *   stats       [2][B*H][Tq] fp32 workspace (row log-sum-exp in log2 units | rowsum(dout * o)), written here;
 *               has_lse = 1: the first half already holds the forward's ur_attn_desc.lse
 * Per-head copies are the same interface with B = batch * heads, H = 1, d = the padded head dim (what the host side
 * does for d = 40: no bounds predicates in the kernels, slightly faster than reading 80-byte head rows in place).
 *   part        fp32 workspace of 2 * G * B*H * Tk64 * dp floats, G = ur_attention_backward_splits(B*H, Tq, Tk64, dp),
 *               Tk64 = Tk rounded up to 64, dp = d rounded up to 32; may be NULL when G == 1 */
typedef struct ur_attn_bwd_desc {
    const void *q, *k, *v, *o, *dout, *qt, *kt, *dot;
    int64_t ldq, ldk, ldv, ldo, lddo, ldqt, ldkt, lddot;
    float* stats;
    void *dq, *dk, *dv;
    int64_t lddq, lddk, lddv;
    float* part;
    int32_t B, H, d, Tq, Tk;
    int32_t Tk_rows;  /* rows per batch of k / v / dk / dv (0: Tk); > Tk when the caller padded the key rows */
    int32_t has_lse;
    float scale;
    int32_t dtype;
} ur_attn_bwd_desc;
int ur_attention_backward(const ur_attn_bwd_desc* d, void* stream);
int ur_attention_backward_supported(int Tq, int Tk, int d);
int ur_attention_backward_splits(int S, int Tq, int Tk64, int dp);
int ur_attention_backward_needs_transposes(void);

/* Attention backward helpers: P = softmax(Q K^T * scale) is recomputed and materialised per (batch, head); the five
 * GEMMs of the gradient (S, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q) run z-batched on ur_igemm.
 * ur_split_heads: x [B][T][ld], head h at columns off + h*d  ->  out [B*H][Tp][dp] (zero padded rows / columns);
 * ur_merge_heads: the inverse; ur_softmax_rows: in place over the first ncols columns of [rows][ld] (padding
 * columns become 0); ur_softmax_backward_rows: dp <- p * (dp - sum_k dp*p) * scale. */
int ur_split_heads(const void* x, int64_t ld, int off, int B, int T, int H, int d, void* out, int Tp, int dp, int dtype,
                   void* stream);
int ur_merge_heads(const void* g, int Tp, int dp, int B, int T, int H, int d, void* out, int64_t ld, int off, int dtype,
                   void* stream);
/* ur_split_heads / ur_merge_heads for up to UR_HEADS_MAX tensors of one (B, H, d, dp) in ONE launch (ABI 9): `tok` is the token
 * matrix [B][T][ld] (head h at columns off + h * d), `heads` the per-head copy [B * H][Tp][dp]; split reads tok and writes
 * heads (rows T .. Tp and columns d .. dp zero), merge the reverse. */
#define UR_HEADS_MAX 8
typedef struct ur_heads_desc {
    void* tok;
    void* heads;
    int64_t ld;
    int32_t off, T, Tp, reserved;
} ur_heads_desc;
int ur_split_heads_multi(const ur_heads_desc* descs, int n, int B, int H, int d, int dp, int dtype, void* stream);
int ur_merge_heads_multi(const ur_heads_desc* descs, int n, int B, int H, int d, int dp, int dtype, void* stream);
int ur_sizeof_heads_desc(void);
/* out[c] = sum over the M rows of in[r][c] for up to UR_COLSUM_MULTI_MAX fp32 matrices [M][N] in ONE launch, fixed-order sums
 * (ABI 9).  pair != 0: the columns are (channel, component) pairs [N / 2][2] and the result is planar, out[k * N / 2 + c] --
 * ur_pairsum_rows.  What the training step uses for the gamma / beta gradients of all LayerNorms and GroupNorms of a network
 * at the end of its backward (their per-wave / per-sample partial sums are summed in one launch instead of one or two per layer). */
#define UR_COLSUM_MULTI_MAX 96
typedef struct ur_colsum_item {
    const float* in;
    float* out;
    int32_t M, N;
    int32_t pair, reserved;
} ur_colsum_item;
int ur_colsum_multi(const ur_colsum_item* items, int n, void* stream);
int ur_sizeof_colsum_item(void);
int ur_softmax_rows(void* s, int64_t ld, int64_t rows, int ncols, int dtype, void* stream);
int ur_softmax_backward_rows(const void* p, void* dp, int64_t ld, int64_t rows, int ncols, float scale, int dtype,
                             void* stream);
/* Training-path glue: SiLU forward (unfused, the pre-activation is kept for the backward), and 2x resampling of NHWC
 * tensors -- mode 0 nearest upsample (forward of Upsample2D), 1 2x2 sum pooling (its backward), 2 zero insertion
 * (dgrad of the stride-2 Downsample2D conv = stride-1 conv of the zero-inserted dY with the rotated weights). */
int ur_silu_forward(const void* x, void* y, int64_t n, int dtype, void* stream);
int ur_resample2x(const void* in, void* out, int B, int Hout, int Wout, int C, int mode, int dtype, void* stream);

/* Training: the fp32 master weight of an nn.Conv2d 3x3 ([Co][Ci][3][3]) to the packed compute-dtype matrix
 * [Co][9 * Cpad] (k = tap * Cpad + c, channels Ci .. Cpad zero) that ur_igemm reads, and the packed weight gradient
 * (row stride ld >= 9 * Cpad) back to an fp32 [Co][Ci][3][3] gradient: cast + repack in one pass each way, instead of
 * the cast / permute / contiguous chain of torch kernels the autocast of train/train.py:1324-1354 amounts to. */
int ur_pack_conv_weight(const float* w, void* out, int Co, int Ci, int Cpad, int dtype, void* stream);
int ur_unpack_conv_weight_grad(const void* dwp, int64_t ld, float* out, int Co, int Ci, int Cpad, int dtype, void* stream);
/* ... with per-workgroup sums of squares of the gradient written, sumsq[0 .. ur_unpack_conv_weight_grad_blocks(Co, Ci)) (ABI 7) */
int ur_unpack_conv_weight_grad_blocks(int Co, int Ci);
int ur_unpack_conv_weight_grad_sumsq(const void* dwp, int64_t ld, float* out, int Co, int Ci, int Cpad, float* sumsq, int dtype,
                                     void* stream);

/*
 * Row-local transformer chains at C = 320 (csrc/tchain.hip): the GEMMs, LayerNorm, GEGLU and residual adds that follow an
 * attention inside one BasicTransformerBlock (diffusers 0.24, instantiated by models/unet_2d_blocks.py:1115-1126 of the
 * reference) as ONE launch, with the activations of a 128-row tile resident in registers and
 * only the weights streamed through LDS.
 *
 *   UR_TCHAIN_PRE y = a0 W0^T + b0                    (Transformer2DModel.proj_in of the GroupNorm output; y leaves as (hi, lo))
 *                 xn = LayerNorm(y);  out = xn Wq^T,  out2 = xn Wk^T  ([zbatch][M][320], no bias),
 *                 out3 = (xn Wv^T)^T per sample: [zbatch * M / rows_per_b][320][ld_vt] (the V^T layout ur_attention reads)
 *   UR_TCHAIN_Q   y = a0 W0^T + b0 + res              (attention out-projection + residual; y leaves as (hi, lo))
 *                 out = LayerNorm(y) Wq^T             (the query projection of the next attention; no bias)
 *   UR_TCHAIN_FF  y = a0 W0^T + b0 + res              (never stored)
 *                 y3 = y + b2 + GEGLU(LayerNorm(y) W1^T + b1) W2^T
 *                 out = y3 Wpo^T + bpo + blk          (Transformer2DModel.proj_out + the block input; (hi, lo))
 *
 * Operands: a0 / res / blk / y_out / out are [zbatch][M][320] row-major token matrices (row stride 320, z stride
 * M * 320); *_lo are the low parts of (hi, lo) residual-stream tensors (NULL: plain tensors / no low part written).
 * `wstream` = per z a sequence of 40960-byte LDS stage images in the order the kernel consumes them, `consts` = per z
 * the fp32 vectors [b0 | gamma | beta] (Q) or [b0 | gamma | beta | b1 value | b1 gate | b2 | bpo] (FF); both are built by
 * the host once per parameter version (uni_renderer_amd/tchain.py documents the image layout: rows x 64 k, 16-byte
 * chunks XOR-swizzled by (row >> 1) & 7, the k columns of every matrix whose operand comes out of an accumulator in the
 * order [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15] per 16).  ur_tchain_stream_bytes / ur_tchain_const_floats give the
 * per-z sizes.  Rows >= M of the last tile are computed and not stored.
 */
#define UR_TCHAIN_Q 0
#define UR_TCHAIN_FF 1
#define UR_TCHAIN_PRE 2
typedef struct ur_tchain_desc {
    const void* a0;
    const void* res;
    const void* res_lo;
    const void* blk;
    const void* blk_lo;
    void* y_out;
    void* y_out_lo;
    void* out;
    void* out_lo;
    void* out2;          /* PRE: k */
    void* out3;          /* PRE: V^T */
    const void* wstream;
    const float* consts;
    int64_t z_wstream;   /* bytes between the weight streams of two z (multiple of 16) */
    int64_t z_consts;    /* floats between the constant blocks of two z */
    int M, zbatch, mode, dtype, channels;
    int rows_per_b;      /* PRE: tokens per sample (multiple of 32, divides M) */
    int ld_vt;           /* PRE: row stride of V^T in elements (>= rows_per_b, multiple of 8) */
    int qk_heads;        /* ABI 12.  0: `out` (PRE, Q) and `out2` (PRE) are token matrices like every other operand.  8: they leave
                          * HEAD-MAJOR, [zbatch * M / rows_per_b][8][rows_per_b][40] -- the q / k images ur_attention reads with
                          * q_hstride = k_hstride = rows_per_b * 40 (needs rows_per_b in mode Q too); other values UR_E_UNSUPPORTED */
    float eps;           /* LayerNorm epsilon */
    void* profile;       /* diagnostics, normally NULL: int64 [workgroups][64] s_memtime stamps (0..15 phases, 16..63 stage starts) of each workgroup's wave 0 */
} ur_tchain_desc;
int ur_tchain(const ur_tchain_desc* d, void* stream);
int64_t ur_tchain_stream_bytes(int mode);
int ur_tchain_const_floats(int mode);

/* Library self-description. */
int ur_abi_version(void);
const char* ur_build_info(void);
/* sizeof() of the descriptor structs as compiled, so a binding can verify its mirror of the layout. */
int ur_sizeof_igemm_desc(void);
int ur_sizeof_attn_desc(void);
int ur_sizeof_attn_bwd_desc(void);
int ur_sizeof_tchain_desc(void);
int ur_sizeof_transpose_desc(void);
int ur_sizeof_wgrad_desc(void);

#ifdef __cplusplus
}
#endif
#endif /* UR_KERNELS_H */
