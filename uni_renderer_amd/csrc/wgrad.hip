// Weight gradients without transposed operand copies (include/ur_kernels.h: ur_wgrad; gfx950 only).
//
//     dw[n][k] = sum_p dy[p][n] * xcol[p][k]          db[n] = sum_p dy[p][n]
//
// The contraction index p (output pixel / token) is the SLOW dimension of both operands, i.e. both MFMA operands are
// "k-strided".  Rounds 1-3 therefore transposed dy and x (and materialised the transposed im2col of x for a conv) so that the
// forward GEMM kernel could be reused: 452 transpose launches, 104 im2col launches and ~480 column-sum launches per training
// step (profiles/r03_train_kernel_stats.csv).  Here the tiles are staged as they lie in memory, [p][n] and [p][k] row-major,
// by LDS-DMA, and the MFMA fragments are gathered with the LDS transpose read: for one 16-lane group, lane 4r + q supplies
// the 8-byte address of elements [row r][4q .. 4q+3] of a 4 x 16 block and lane i receives column i of it
// (ds_read_b64_tr_b16; cdna_hip_programming.md T10) -- two such reads give the 8 contraction values a lane feeds to
// v_mfma_f32_16x16x32.  The conv form reads the im2col rows implicitly: a lane's 16-byte piece belongs to one (tap, channel
// block) for the whole loop, only the pixel advances.
//
// LDS image of a tile of width TW elements (TW * 2 bytes per row, rows = p): 32-byte chunk c of row r is stored at chunk
// c ^ key(r), where key takes the row bits that differ between the 8 (row, group) pairs one half-wave reads at once
// (rows g * 8 + rr, rr = 0..3, g = 0, 1): key = rr | g0 << 2 for 256-byte rows and above; for 128-byte rows (two rows per
// 256-byte bank window) rr's low bit already separates the rows and key = rr1 | g0 << 1.  The swizzle lives in the SOURCE
// address of the LDS-DMA (lane-linear destination).
//
// Column sums of dy (the bias gradient) are taken from the dy fragments by the workgroups of k-tile 0: no separate launch.
// Reference: the autograd of F.linear / F.conv2d under train/train.py:1416 (accelerator.backward).
#include "ur_common.h"
#include "../../include/ur_kernels.h"

namespace ur {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <int TW>
__device__ __forceinline__ int wg_key(int r) {
    return TW * 2 == 128 ? (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) : ((r & 3) | (((r >> 3) & 1) << 2));
}

constexpr int WG_BKP = 32;  // contraction rows per stage

// kernel arguments: the shared description + the operand pointers of up to UR_WGRAD_GROUP_MAX equally shaped problems
// (ur_wgrad_group: the q / k / v / out / feed-forward weight gradients of all transformer blocks of a level in ONE launch)
struct WgradArgs {
    ur_wgrad_desc d;
    int n;
    ur_wgrad_ptrs g[UR_WGRAD_GROUP_MAX];
};

template <int TK, int TN, int NS>
constexpr int wgrad_lds_bytes() { return NS * WG_BKP * (TK + TN) * 2; }

// TK: tile extent along k (columns of xcol = MFMA rows), TN: along n (columns of dy = MFMA columns); WK x WN waves.
// The loop is bound by the global -> LDS delivery of the two tiles (ablations, profiles/r04_wgrad_ablate.txt: without the
// MFMAs the 128 x 128 kernel takes the same time), i.e. time ~ 1 / TK + 1 / TN: hence the 256-wide tiles with 8 / 16 waves.
// ABL (experiment builds only, `make WGRAD_ABL=1`; results are garbage): 1 no LDS-DMA copies, 2 no MFMAs, 3 no fragment reads,
// 4 plain ds_read_b64 at the same addresses instead of the transpose read
template <typename T, int TK, int TN, int WK, int WN, int NS, bool CONV, int ABL = 0>
__global__ void __launch_bounds__(WK * WN * 64, 4) wgrad_kernel(const WgradArgs a) {
    const ur_wgrad_desc& p = a.d;
    typedef typename Vec8<T>::type vec8;
    constexpr int BKP = WG_BKP;
    constexpr int NW = WK * WN;
    constexpr int KREP = TK / WK / 16, NREP = TN / WN / 16;
    constexpr int XT_BYTES = BKP * TK * 2, YT_BYTES = BKP * TN * 2, STAGE = XT_BYTES + YT_BYTES;
    constexpr int XSPR = TK / 8, YSPR = TN / 8;              // 16-byte slots per tile row
    constexpr int XI = BKP * XSPR / 64 / NW, YI = BKP * YSPR / 64 / NW;  // LDS-DMA instructions per wave per tile
    static_assert(XI >= 1 && YI >= 1 && XI * NW * 64 == BKP * XSPR && YI * NW * 64 == BKP * YSPR, "uniform copies per wave");
    constexpr int LOADS = XI + YI;
    constexpr int D = NS - 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / WN, wn = wave % WN;

    const int tiles_k = (p.K + TK - 1) / TK, tiles_n = (p.N + TN - 1) / TN;
    const int tiles = tiles_k * tiles_n;
    // slice-major logical ids: an XCD owns a contiguous run of them, i.e. a few P slices whose dy / x rows stay in its L2
    // while every (k, n) tile of the slice passes over them
    // problem-major, then slice-major
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int prob = lid / (tiles * p.splits);
    const int lip = lid - prob * tiles * p.splits;
    const int split = lip / tiles;
    const int t_ = lip - split * tiles;
    const ur_wgrad_ptrs gp = a.g[prob];
    // the dimension with fewer tiles runs fastest: the run of tiles an XCD works on at a time then spans a near-square patch
    // (few distinct dy AND few distinct x tiles in its L2) instead of one x tile against every dy tile
    const bool k_inner = tiles_k <= tiles_n;
    const int tile_n = k_inner ? t_ / tiles_k : t_ % tiles_n, tile_k = k_inner ? t_ % tiles_k : t_ / tiles_n;
    const int k0 = tile_k * TK, n0 = tile_n * TN;

    const int steps_total = (p.P + BKP - 1) / BKP;
    const int per = (steps_total + p.splits - 1) / p.splits;
    const int sbeg = split * per, send = min(steps_total, sbeg + per);
    const int nk = send - sbeg;

    // padding rows: this (workgroup, wave)'s own 128-byte line of the zero region (uniform base + one lane-offset register)
    const unsigned zbytes = p.zero_page_bytes >= 256 ? (unsigned)p.zero_page_bytes : 256u;
    const char* const zpb = reinterpret_cast<const char*>(p.zero_page) + ((((unsigned)lid * (unsigned)NW + (unsigned)wave) * 128u) & (zbytes - 128u));
    const int zoff = (lane & 7) * 16;

    // ---- loader bookkeeping: every lane owns XI + YI 16-byte slots (tile row r, source column block), each described by
    // one packed word (registers are what bounds the number of workgroups per CU): byte offset of the column in bits 0..23,
    // tile row in bits 24..28, valid column in bit 31; a conv adds (ky, kx) of the slot's tap in a second word ----
    const int PP = p.P;
    const char* const dyb = reinterpret_cast<const char*>(gp.dy);
    const char* const xb = reinterpret_cast<const char*>(gp.x);
    unsigned ymeta[YI], xmeta[XI];
    int xtap[XI];
    const unsigned ystep = (unsigned)p.lddy * (unsigned)sizeof(T);  // byte offsets fit 32 bits (wgrad_check)
    const unsigned xstep = (unsigned)p.ldx * (unsigned)sizeof(T);
#pragma unroll
    for (int it = 0; it < YI; ++it) {
        const int L = (it * NW + wave) * 64 + lane;
        const int r = L / YSPR, s = L % YSPR;
        const int n = n0 + ((s ^ (wg_key<TN>(r) << 1)) << 3);
        ymeta[it] = (unsigned)(n * (int)sizeof(T)) | ((unsigned)r << 24) | (n < p.N ? 0x80000000u : 0u);
    }
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int L = (it * NW + wave) * 64 + lane;
        const int r = L / XSPR, s = L % XSPR;
        const int k = k0 + ((s ^ (wg_key<TK>(r) << 1)) << 3);
        int col = k;
        xtap[it] = 0;
        if (CONV) {
            const int tap = k / p.C;
            col = k - tap * p.C;
            xtap[it] = ((tap / 3 - p.pad) & 0xffff) | ((tap % 3 - p.pad) << 16);
        }
        xmeta[it] = (unsigned)(col * (int)sizeof(T)) | ((unsigned)r << 24) | (k < p.K ? 0x80000000u : 0u);
    }
    const int wsh = CONV ? 31 - __builtin_clz((unsigned)p.Wout) : 0;
    const int hsh = CONV ? 31 - __builtin_clz((unsigned)p.Hout) : 0;
    const int pHin = p.Hin, pWin = p.Win, pstride = p.stride, pWo1 = p.Wout - 1, pHo1 = p.Hout - 1;

    int lstep = sbeg;  // next step the loader issues
    auto stage = [&](int buf) {
        char* xs = smem + buf * STAGE;
        char* ys = xs + XT_BYTES;
        const int pb = lstep * BKP;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int pr = pb + (int)((xmeta[it] >> 24) & 31u);
            const bool colok = (int)xmeta[it] < 0;
            const char* src = zpb + zoff;
            if (CONV) {
                const int ox = pr & pWo1, oy = (pr >> wsh) & pHo1, b = pr >> (wsh + hsh);
                const int iy = oy * pstride + (int)(short)(xtap[it] & 0xffff), ix = ox * pstride + (xtap[it] >> 16);
                const bool ok = pr < PP && colok && (unsigned)iy < (unsigned)pHin && (unsigned)ix < (unsigned)pWin;
                if (ok) src = xb + (size_t)((unsigned)((b * pHin + iy) * pWin + ix) * xstep + (xmeta[it] & 0xffffffu));
            } else {
                if (pr < PP && colok) src = xb + (size_t)((unsigned)pr * xstep + (xmeta[it] & 0xffffffu));
            }
            if (ABL != 1) glds16(src, xs + (it * NW + wave) * 1024);
        }
#pragma unroll
        for (int it = 0; it < YI; ++it) {
            const int pr = pb + (int)((ymeta[it] >> 24) & 31u);
            const char* src = (pr < PP && (int)ymeta[it] < 0) ? dyb + (size_t)((unsigned)pr * ystep + (ymeta[it] & 0xffffffu)) : zpb + zoff;
            if (ABL != 1) glds16(src, ys + (it * NW + wave) * 1024);
        }
        lstep += 1;
    };

    // ---- consumer bookkeeping ----
    const int g = lane >> 4, i16 = lane & 15, rr = i16 >> 2, q4 = i16 & 3;
    const int frow = g * 8 + rr;                       // tile row of the first transpose read (the second: + 4)
    int xoff[KREP], yoff[NREP];
#pragma unroll
    for (int f = 0; f < KREP; ++f)
        xoff[f] = frow * (TK * 2) + (((wk * KREP + f) ^ wg_key<TK>(frow)) << 5) + q4 * 8;
#pragma unroll
    for (int f = 0; f < NREP; ++f)
        yoff[f] = frow * (TN * 2) + (((wn * NREP + f) ^ wg_key<TN>(frow)) << 5) + q4 * 8;

    f32x4 acc[KREP][NREP];
#pragma unroll
    for (int a = 0; a < KREP; ++a)
#pragma unroll
        for (int b = 0; b < NREP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[NREP];
#pragma unroll
    for (int b = 0; b < NREP; ++b) dbacc[b] = 0.f;
    const bool want_db = gp.db != nullptr && tile_k == 0 && wk == 0;  // wave-uniform

    auto frag = [&](const char* base, int off, int row_bytes) __attribute__((always_inline)) {
        if constexpr (ABL == 3) {
            s16x8 c;
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = (short)(off + j);
            return __builtin_bit_cast(vec8, c);
        }
        if constexpr (ABL == 4) {
            const s16x4 lo = *reinterpret_cast<const s16x4*>(base + off), hi = *reinterpret_cast<const s16x4*>(base + off + 4 * row_bytes);
            const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(vec8, v);
        }
        const s16x4 lo = lds_tr16(base + off), hi = lds_tr16(base + off + 4 * row_bytes);
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(vec8, v);
    };
    auto compute = [&](int buf) {
        const char* xs = smem + buf * STAGE;
        const char* ys = xs + XT_BYTES;
        vec8 xf[KREP], yf[NREP];
#pragma unroll
        for (int f = 0; f < NREP; ++f) yf[f] = frag(ys, yoff[f], TN * 2);
#pragma unroll
        for (int f = 0; f < KREP; ++f) xf[f] = frag(xs, xoff[f], TK * 2);
        if (want_db) {
#pragma unroll
            for (int f = 0; f < NREP; ++f) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)yf[f][j];
                dbacc[f] += s;
            }
        }
        if constexpr (ABL == 2) {
#pragma unroll
            for (int a = 0; a < KREP; ++a)
#pragma unroll
                for (int b = 0; b < NREP; ++b) acc[a][b][0] += (float)xf[a][b & 7] + (float)yf[b][a & 7];
            return;
        }
#pragma unroll
        for (int a = 0; a < KREP; ++a)
#pragma unroll
            for (int b = 0; b < NREP; ++b) acc[a][b] = mfma16(xf[a], yf[b], acc[a][b]);
    };

    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nk) stage(s);
        int buf = 0, nbuf = D % NS;
        for (int t = 0; t < nk; ++t) {
            const int newer = min(D - 1, nk - 1 - t);  // stages issued after stage t that may stay in flight
            static_assert((D - 1) * LOADS <= 63 && D - 1 <= 6, "vmcnt is a 6-bit immediate");
            switch (newer) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS > 63 ? 63 : 3 * LOADS) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LOADS > 63 ? 63 : 4 * LOADS) : "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LOADS > 63 ? 63 : 5 * LOADS) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LOADS > 63 ? 63 : 6 * LOADS) : "memory"); break;
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + D < nk) stage(nbuf);
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1;
        }
    }

    // ---- epilogue: lane (g, i16) of block (a, b) holds dw[n = .. + i16][k = .. + 4 g + 0..3] ----
    const int ldp = (p.K + 4 + 3) & ~3;  // slab row: K gradient columns, then the bias-gradient column
    const int Np = (p.N + 7) & ~7;
    float* slab = p.splits > 1 ? p.partial + ((int64_t)prob * p.splits + split) * Np * ldp : nullptr;
#pragma unroll
    for (int b = 0; b < NREP; ++b) {
        const int n = n0 + (wn * NREP + b) * 16 + i16;
#pragma unroll
        for (int a = 0; a < KREP; ++a) {
            const int k = k0 + (wk * KREP + a) * 16 + 4 * g;
            if (n < p.N && k < p.K) {
                if (slab) {
                    *reinterpret_cast<float4*>(slab + (int64_t)n * ldp + k) =
                        make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
                } else {
                    T o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = from_f<T>(acc[a][b][r]);
                    uint2 bits;
                    __builtin_memcpy(&bits, o, 8);
                    *reinterpret_cast<uint2*>(reinterpret_cast<T*>(gp.dw) + (int64_t)n * p.lddw + k) = bits;
                }
            }
        }
        if (want_db) {
            float v = dbacc[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0 && n < p.N) {
                if (slab) slab[(int64_t)n * ldp + p.K] = v;
                else gp.db[n] = v;
            }
        }
    }
}

// second pass: sum the slabs in slice order; columns < K -> dw (dtype), column K -> db.  blockIdx.y = problem.
template <typename T>
__global__ void __launch_bounds__(256) wgrad_reduce(const WgradArgs a) {
    const ur_wgrad_desc& p = a.d;
    const ur_wgrad_ptrs gp = a.g[blockIdx.y];
    const int ldp = (p.K + 4 + 3) & ~3;
    const int Np = (p.N + 7) & ~7;
    const int groups = p.K / 4 + (gp.db ? 1 : 0);
    const int64_t total = (int64_t)p.N * groups;
    const int64_t slab = (int64_t)Np * ldp;
    const float* part = p.partial + (int64_t)blockIdx.y * p.splits * slab;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / groups), gq = (int)(idx - (int64_t)n * groups);
        const float* src = part + (int64_t)n * ldp + gq * 4;
        if (gq * 4 >= p.K) {
            float s = 0.f;
            for (int z = 0; z < p.splits; ++z) s += src[z * slab];
            gp.db[n] = s;
            continue;
        }
        float4 s = *reinterpret_cast<const float4*>(src);
        for (int z = 1; z < p.splits; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(src + z * slab);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        T o[4] = {from_f<T>(s.x), from_f<T>(s.y), from_f<T>(s.z), from_f<T>(s.w)};
        uint2 bits;
        __builtin_memcpy(&bits, o, 8);
        *reinterpret_cast<uint2*>(reinterpret_cast<T*>(gp.dw) + (int64_t)n * p.lddw + gq * 4) = bits;
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int wgrad_check(const ur_wgrad_desc& d) {
    if (!d.dy || !d.x || !d.dw || !d.zero_page) return UR_E_BADARG;
    if (d.P <= 0 || d.N <= 0 || d.K <= 0 || d.N % 8 || d.K % 8) return UR_E_BADARG;
    if (d.lddy % 8 || d.ldx % 8 || d.lddw % 4 || d.lddw < d.K || d.lddy < d.N) return UR_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(d.dy) | reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.zero_page)) & 15) return UR_E_BADARG;
    if (reinterpret_cast<uintptr_t>(d.dw) & 7) return UR_E_BADARG;
    if (d.dtype != UR_DT_F16 && d.dtype != UR_DT_BF16) return UR_E_BADARG;
    if (d.zero_page_bytes && (d.zero_page_bytes < 256 || (d.zero_page_bytes & (d.zero_page_bytes - 1)))) return UR_E_BADARG;
    if (d.taps == 9) {
        if (d.C <= 0 || d.C % 64 || d.K != 9 * d.C || d.ldx < d.C) return UR_E_BADARG;
        if (d.B <= 0 || d.Hin <= 0 || d.Win <= 0 || (int64_t)d.B * d.Hout * d.Wout != d.P) return UR_E_BADARG;
        if (d.stride < 1 || d.stride > 2 || d.pad < 0 || d.pad > 1) return UR_E_BADARG;
        if (!pow2(d.Hout) || !pow2(d.Wout)) return UR_E_UNSUPPORTED;
        if ((int64_t)d.B * d.Hin * d.Win * d.ldx * 2 >= (1ll << 32)) return UR_E_UNSUPPORTED;  // 32-bit byte offsets in the loader
    } else if (d.taps == 1) {
        if (d.ldx < d.K) return UR_E_BADARG;
        if ((int64_t)d.P * d.ldx * 2 >= (1ll << 32)) return UR_E_UNSUPPORTED;
    } else {
        return UR_E_BADARG;
    }
    if ((int64_t)d.N * 2 >= (1 << 24) || (int64_t)(d.taps == 9 ? d.C : d.K) * 2 >= (1 << 24)) return UR_E_UNSUPPORTED;
    if ((int64_t)d.P * d.lddy * 2 >= (1ll << 32)) return UR_E_UNSUPPORTED;
    if (d.tile < 0 || d.tile > 22 || (d.tile && ((d.tile & 7) == 0 || (d.tile & 7) == 7))) return UR_E_BADARG;
    return 0;
}

// Untuned problems (the Python layer looks a measured (tile, slices) pair up first: uni_renderer_amd/wgrad_tuning.json):
// 128 x 128, or 128 x 64 when a 128-wide tile would waste more than a fifth of the dy columns (N = 320), 64 x 64 for
// narrow problems.
static int wgrad_tile(const ur_wgrad_desc& d) {
    if (d.tile) return d.tile;
    if (d.N <= 64 || d.K <= 64) return 3;
    const int n128 = (d.N + 127) / 128 * 128;
    return (n128 - d.N) * 5 > n128 ? 2 : 1;
}

static void tile_dims(int tile, int& tk, int& tn) {
    static const int dims[8][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {256, 256}, {256, 128}, {128, 256}, {0, 0}};
    tk = dims[tile & 7][0];
    tn = dims[tile & 7][1];
}

// Slices: a power of two (the slices then map onto whole XCDs) that brings the launch closest to ~4 workgroups of 4 waves
// per CU, with at least 8 stages per slice (every slice costs a tile of fp32 slab traffic).
static int wgrad_auto_splits(const ur_wgrad_desc& d, int nprob) {
    int tk, tn;
    tile_dims(wgrad_tile(d), tk, tn);
    const int tiles = ((d.K + tk - 1) / tk) * ((d.N + tn - 1) / tn) * nprob;
    const int steps = (d.P + WG_BKP - 1) / WG_BKP;
    const int waves = (tk / 64) * (tn / 64) < 4 ? 4 : (tk / 64) * (tn / 64);
    const int target = 1024 * 4 / waves;
    int best = 1;
    for (int s = 2; s <= 64 && s * 8 <= steps; s *= 2) {
        // |log(tiles * s / target)| smaller than for `best`  <=>  tiles^2 * s * best closer to target^2 (both sides of it)
        const double a = (double)tiles * s / target, b = (double)tiles * best / target;
        const double da = a > 1 ? a : 1 / a, db = b > 1 ? b : 1 / b;
        if (da < db) best = s;
    }
    return best;
}

static int64_t wgrad_floats(const ur_wgrad_desc& d, int splits) {
    if (splits <= 1) return 0;
    const int64_t ldp = (d.K + 4 + 3) & ~3, Np = (d.N + 7) & ~7;
    return (int64_t)splits * Np * ldp;
}

#ifdef UR_WGRAD_ABL
template <typename T, int TK, int TN, int WK, int WN, int NS, int ABL>
static void wgrad_launch_abl(const WgradArgs& a, hipStream_t s, dim3 grid, int lds) {
    if (a.d.taps == 9) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, TK, TN, WK, WN, NS, true, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((wgrad_kernel<T, TK, TN, WK, WN, NS, true, ABL>), grid, dim3(WK * WN * 64), lds, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, TK, TN, WK, WN, NS, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((wgrad_kernel<T, TK, TN, WK, WN, NS, false, ABL>), grid, dim3(WK * WN * 64), lds, s, a);
    }
}
#endif

template <typename T, int TK, int TN, int WK, int WN, int NS>
static int wgrad_launch_cfg(const WgradArgs& a, hipStream_t s) {
    static std::atomic<uint64_t> done_c{0}, done_l{0};
    const ur_wgrad_desc& d = a.d;
    constexpr int lds = wgrad_lds_bytes<TK, TN, NS>();
    const int tiles = ((d.K + TK - 1) / TK) * ((d.N + TN - 1) / TN);
    const dim3 grid(tiles * d.splits * a.n);
#ifdef UR_WGRAD_ABL
    if (const char* e = getenv("UR_WGRAD_ABLATE")) {
        switch (atoi(e)) {
            case 1: wgrad_launch_abl<T, TK, TN, WK, WN, NS, 1>(a, s, grid, lds); return 0;
            case 2: wgrad_launch_abl<T, TK, TN, WK, WN, NS, 2>(a, s, grid, lds); return 0;
            case 3: wgrad_launch_abl<T, TK, TN, WK, WN, NS, 3>(a, s, grid, lds); return 0;
            case 4: wgrad_launch_abl<T, TK, TN, WK, WN, NS, 4>(a, s, grid, lds); return 0;
            default: break;
        }
    }
#endif
    if (d.taps == 9) {
        set_lds_limit_once(done_c, reinterpret_cast<const void*>(&wgrad_kernel<T, TK, TN, WK, WN, NS, true>), lds);
        hipLaunchKernelGGL((wgrad_kernel<T, TK, TN, WK, WN, NS, true>), grid, dim3(WK * WN * 64), lds, s, a);
    } else {
        set_lds_limit_once(done_l, reinterpret_cast<const void*>(&wgrad_kernel<T, TK, TN, WK, WN, NS, false>), lds);
        hipLaunchKernelGGL((wgrad_kernel<T, TK, TN, WK, WN, NS, false>), grid, dim3(WK * WN * 64), lds, s, a);
    }
    if (d.splits > 1) {
        const int64_t total = (int64_t)d.N * (d.K / 4 + 1);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL((wgrad_reduce<T>), dim3(blocks, a.n), dim3(256), 0, s, a);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
static int wgrad_launch(const WgradArgs& d, hipStream_t s) {
    // tile id = dims (1..6) + 8 * depth code; depth code 0 = the product's 2-stage ring (more workgroups per CU hide the
    // delivery latency better than a deeper ring does: profiles/r04_wgrad_ring.txt), 1 / 2 = 3 / 4 stages (experiment builds)
    switch (wgrad_tile(d.d)) {
        case 1: return wgrad_launch_cfg<T, 128, 128, 2, 2, 2>(d, s);
        case 2: return wgrad_launch_cfg<T, 128, 64, 2, 2, 2>(d, s);
        case 3: return wgrad_launch_cfg<T, 64, 64, 2, 2, 2>(d, s);
        case 4: return wgrad_launch_cfg<T, 256, 256, 4, 4, 2>(d, s);
        case 5: return wgrad_launch_cfg<T, 256, 128, 4, 2, 2>(d, s);
        case 6: return wgrad_launch_cfg<T, 128, 256, 2, 4, 2>(d, s);
#ifdef UR_WGRAD_ABL
        case 9: return wgrad_launch_cfg<T, 128, 128, 2, 2, 3>(d, s);
        case 10: return wgrad_launch_cfg<T, 128, 64, 2, 2, 3>(d, s);
        case 11: return wgrad_launch_cfg<T, 64, 64, 2, 2, 3>(d, s);
        case 12: return wgrad_launch_cfg<T, 256, 256, 4, 4, 3>(d, s);
        case 13: return wgrad_launch_cfg<T, 256, 128, 4, 2, 3>(d, s);
        case 14: return wgrad_launch_cfg<T, 128, 256, 2, 4, 3>(d, s);
        case 17: return wgrad_launch_cfg<T, 128, 128, 2, 2, 4>(d, s);
        case 19: return wgrad_launch_cfg<T, 64, 64, 2, 2, 4>(d, s);
        case 20: return wgrad_launch_cfg<T, 256, 256, 4, 4, 4>(d, s);
#endif
        default: return UR_E_UNSUPPORTED;
    }
}

}  // namespace ur

extern "C" int ur_sizeof_wgrad_desc(void) { return (int)sizeof(ur_wgrad_desc); }

static int wgrad_group_check(const ur_wgrad_desc& c, const ur_wgrad_ptrs* g, int n) {
    if (!g || n < 1 || n > UR_WGRAD_GROUP_MAX) return UR_E_BADARG;
    for (int i = 0; i < n; ++i) {
        ur_wgrad_desc d = c;
        d.dy = g[i].dy; d.x = g[i].x; d.dw = g[i].dw; d.db = g[i].db;
        const int rc = ur::wgrad_check(d);
        if (rc) return rc;
    }
    if (c.splits < 1 || (c.splits > 1 && !c.partial)) return UR_E_BADARG;
    if (c.splits > (c.P + ur::WG_BKP - 1) / ur::WG_BKP) return UR_E_BADARG;
    return 0;
}

extern "C" int ur_wgrad_group_plan(const ur_wgrad_desc* d, const ur_wgrad_ptrs* g, int n, int32_t* splits, int64_t* partial_floats) {
    if (!d || !splits || !partial_floats) return UR_E_BADARG;
    ur_wgrad_desc c = *d;
    c.splits = 1;
    const int rc = wgrad_group_check(c, g, n);
    if (rc) return rc;
    *splits = ur::wgrad_auto_splits(c, n);
    *partial_floats = ur::wgrad_floats(c, *splits) * n;
    return 0;
}

extern "C" int ur_wgrad_group(const ur_wgrad_desc* d, const ur_wgrad_ptrs* g, int n, void* stream) {
    if (!d) return UR_E_BADARG;
    const int rc = wgrad_group_check(*d, g, n);
    if (rc) return rc;
    ur::WgradArgs a;
    a.d = *d;
    a.n = n;
    for (int i = 0; i < n; ++i) a.g[i] = g[i];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == UR_DT_F16 ? ur::wgrad_launch<ur::f16>(a, s) : ur::wgrad_launch<ur::bf16>(a, s);
}

extern "C" int ur_wgrad_plan(const ur_wgrad_desc* d, int32_t* splits, int64_t* partial_floats) {
    if (!d) return UR_E_BADARG;
    const ur_wgrad_ptrs g = {d->dy, d->x, d->dw, d->db};
    return ur_wgrad_group_plan(d, &g, 1, splits, partial_floats);
}

extern "C" int64_t ur_wgrad_partial_floats(const ur_wgrad_desc* d) { return d ? ur::wgrad_floats(*d, d->splits) : 0; }

extern "C" int ur_wgrad(const ur_wgrad_desc* d, void* stream) {
    if (!d) return UR_E_BADARG;
    const ur_wgrad_ptrs g = {d->dy, d->x, d->dw, d->db};
    return ur_wgrad_group(d, &g, 1, stream);
}
