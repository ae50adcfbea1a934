// 3x3 conv as implicit GEMM where THE THREE dx TAPS SHARE ONE STAGED PIXEL BLOCK (round 4).
//
// Why.  The conv tiles are bound by what a CU can pull global -> LDS ("Round 4"); per 64-deep K chunk a
// 128x320 tile copies 16 KB of pixels and 40 KB of weights.  The pixel tile of tap (dy, dx) is the pixel tile of tap (dy, 1)
// shifted by one pixel -- with zeros exactly where the shift leaves the image row -- so when a tile covers WHOLE image rows
// the three dx taps can multiply out of ONE staged block that carries a one-pixel zero halo on both ends of every image
// row.  A timing-only build that simply skipped two of three pixel copies measured +3.9 % steps/s in the step
// (profiles/r04_dxshare_bound.txt): the upper bound this kernel goes after.
//
// What changes against igemm.hip's lock-step kernel (same descriptor, weight packing, split-K slabs, epilogue):
//   * K is walked as GROUPS (channel block, dy, 64-channel chunk) of three STEPS dx = 0, 1, 2 (then the 1x1 tail, one
//     single-step group per chunk).  The weight pointer of a step is a scalar offset into the packed row
//     (k = blk * 9 * cb + (3 dy + dx) * cb + 64 c), so no re-pack is needed.
//   * LDS: two pixel buffers (one per group parity) of BM + 2 * BM / W rows -- image row i of the tile at LDS rows
//     i * (W + 2) .. + W + 1, columns -1 .. W -- and two weight buffers (one per step parity).  The loader runs one step
//     ahead: every step copies the next step's weight tile, the last step of a group also the next group's pixel block.
//   * the B fragments of step dx are read at LDS row r + 2 * (r / W) + dx for tile pixel r (the XOR swizzle key follows the
//     shifted row); out-of-image reads hit the zero halo / zero rows, which IS the conv's padding.
//   * split-K slices are ranges of groups (any partition of K sums to the same slabs).
// Eligible: taps == 9, stride 1, pad 1, one source, W_out a power of two >= 8 dividing the tile height BM (all resnet convs of
// the UNets at 64x64 .. 8x8, including the nearest-2x fused upsample convs); everything else stays on igemm.hip.
#include "igemm_epi.h"

// -DUR_DXS_ABLATE=<bits>: timing-only builds (results are garbage; tools/experiments/r04_run13.sh):
//   1 = the pixel block is never copied, 2 = fragment reads at the unshifted rows (address math hoisted out of the loop),
//   4 = no loader bookkeeping (weight offset 0, pixel pointers never updated)
#ifndef UR_DXS_ABLATE
#define UR_DXS_ABLATE 0
#endif

namespace ur {

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) igemm_dxs_kernel(const ur_igemm_desc p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int NW = WM * WN;
    constexpr int MREP = BM / WM / 16;
    constexpr int NREP = BN / WN / 16;
    static_assert(NREP == 4, "a wave spans 64 output columns (epilogue layout)");
    constexpr int ARMAX = BM + BM / 4;            // pixel-block rows incl. halos for W_out >= 8
    constexpr int XP = ARMAX / 8, WP = BN / 8;    // LDS-DMA pieces (8 rows x 128 B)
    constexpr int XI = (XP + NW - 1) / NW, WI = (WP + NW - 1) / NW;
    constexpr int A_BYTES = ARMAX * 128, W_BYTES = BN * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // A0 | A1 | W0 | W1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
    const int zidx = lid / gridDim.x;
    const int tid_xy = lid - zidx * gridDim.x;
    const int tile_n = tid_xy % tiles_n;
    const int tile_m = tid_xy / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // K as groups: 3 * c0 / 64 main groups (three steps each), then (ct0 + ct1) / 64 tail groups (one step each)
    const int C = p.c0;
    const int cb = p.cblock > 0 ? p.cblock : C;  // channel block of the packed K order
    const int cpb = cb / BK;                     // 64-channel chunks per block
    const int gmain = 3 * (C / BK);
    const int gtail0 = p.ct0 / BK, gtail = (p.ct0 + p.ct1) / BK;
    const int gtotal = gmain + gtail;
    int gbeg = 0, gend = gtotal, zb = zidx;
    if (p.splitk > 1) {
        zb = zidx / p.splitk;
        const int ks = zidx - zb * p.splitk;
        const int per = (gtotal + p.splitk - 1) / p.splitk;
        gbeg = min(gtotal, ks * per);
        gend = min(gtotal, gbeg + per);
    }
    const char* x0 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.x0) + (int64_t)(zb / p.zx_div) * p.zx);
    const char* wp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.w) + (int64_t)zb * p.zw);
    const char* t0 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t0) + (int64_t)zb * p.zt0);
    const char* t1 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t1) + (int64_t)zb * p.zt1);
    const int jsw = (lane & 7) ^ (lane >> 3);  // swizzled source chunk of this lane (piece bases are multiples of 8 rows)
    // padding rows: this (workgroup, wave)'s own 128-byte line of the zero region (one hot line would be served to all CUs
    // by one L2 channel)
    const unsigned zbytes = p.zero_page_bytes >= 256 ? (unsigned)p.zero_page_bytes : 256u;
    const char* zp = reinterpret_cast<const char*>(p.zero_page) + ((((unsigned)lid * 16u + (unsigned)wave) * 128u) & (zbytes - 128u)) + (lane & 7) * 16;

    // ---- per-lane rows of the pixel block (fixed over the K loop): LDS row rho = image row ir of the tile, column col ----
    const int Wo = p.Wout, Ho = p.Hout, W2 = Wo + 2, ups = p.ups;
    const int Win = p.Win, HWin = p.Hin * p.Win;
    const int wsh = 31 - __builtin_clz(Wo);
    const int rows_tile = BM >> wsh;          // image rows per tile
    const int xp_used = (rows_tile * W2 + 7) >> 3;
    int a_base[XI], a_bout[XI], a_oy[XI], a_col[XI];  // base pixel of the sample in the input / output-sized image, output row, output column (or invalid)
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int rho = (it * NW + wave) * 8 + (lane >> 3);
        const int ir = rho / W2;
        const int col = rho - ir * W2 - 1;
        const int R = (m0 >> wsh) + ir;       // image row counted over the whole batch
        const int b = R / Ho;
        const bool ok = ir < rows_tile && col >= 0 && col < Wo && (m0 + (ir << wsh)) < p.M;
        a_base[it] = b * HWin;
        a_bout[it] = b * Ho * Wo;
        a_oy[it] = R - b * Ho;
        a_col[it] = ok ? col : -(1 << 20);
    }
    const char* wbase[WI];
    bool wok[WI];
#pragma unroll
    for (int it = 0; it < WI; ++it) {
        const int r = (it * NW + wave) * 8 + (lane >> 3);
        const int rho = r & 63;
        // LDS row (f, i) = f*16 + i holds semantic column (i>>2)*16 + f*4 + (i&3) of its 64-group (igemm.hip, MF = 16)
        const int sem = (r & ~63) | (((rho >> 2) & 3) << 4) | ((rho >> 4) << 2) | (rho & 3);
        const int n = n0 + sem;
        wok[it] = n < p.N;
        wbase[it] = wp + ((int64_t)n * p.ldw + jsw * 8) * (int64_t)sizeof(T);
    }

    // ---- loader state (wave-uniform).  Two cursors: the WEIGHT cursor is the next STEP to copy (one step ahead of the
    // consumer); the PIXEL cursor is the next GROUP to copy (one group ahead: a group's pixel block goes out at the FIRST
    // step of the group before it, so it has two to three steps to arrive -- its lines are first touches of another
    // kernel's output and miss L2, unlike the weight lines) ----
    const int64_t pldx0 = p.ldx0, pldt0 = p.ldt0, pldt1 = p.ldt1;
    struct Cur { int g, blk, dy, c; };  // group index; (channel block, dy, chunk) while g < gmain
    auto cur_at = [&](int g) __attribute__((always_inline)) {
        Cur k;
        k.g = g;
        const int gm = min(g, max(gmain - 1, 0));
        k.blk = gm / (3 * cpb);
        const int rem = gm - k.blk * 3 * cpb;
        k.dy = rem / cpb;
        k.c = rem - k.dy * cpb;
        return k;
    };
    auto cur_next = [&](Cur& k) __attribute__((always_inline)) {
        k.g += 1;
        k.c += 1;
        if (k.c == cpb) { k.c = 0; k.dy += 1; if (k.dy == 3) { k.dy = 0; k.blk += 1; } }
        k.g = __builtin_amdgcn_readfirstlane(k.g);
        k.c = __builtin_amdgcn_readfirstlane(k.c);
        k.dy = __builtin_amdgcn_readfirstlane(k.dy);
        k.blk = __builtin_amdgcn_readfirstlane(k.blk);
    };
    const char* xptr[XI];
    int xinc[XI];
    // pixel pointers of group `k`: a main group (blk, dy, c) reads x0 at rows oy + dy - 1 (nearest-2x: >> ups), channels
    // blk * cb + 64 c ..; a tail group reads t0 / t1 at the output pixel.  Inside a (blk, dy) run / a tail source the next
    // group is the next 64 channels: pointer += 128 bytes (`fresh` = false); only a new run recomputes.
    auto a_pointers = [&](const Cur& k, bool fresh) __attribute__((always_inline)) {
        const bool main = k.g < gmain;
        const int tg = k.g - gmain;
        const bool first = main ? k.c == 0 : (tg == 0 || tg == gtail0);
        if (!fresh && !first) {
#pragma unroll
            for (int it = 0; it < XI; ++it) xptr[it] += xinc[it];
            return;
        }
        const char* sb = x0;
        int64_t ld = pldx0;
        int dy = k.dy, coff = k.blk * cb + k.c * BK;
        if (!main) {
            dy = 1;
            if (tg < gtail0) { sb = t0; ld = pldt0; coff = tg * BK; }
            else { sb = t1; ld = pldt1; coff = (tg - gtail0) * BK; }
        }
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int iy = a_oy[it] + dy - 1;
            const bool ok = a_col[it] >= 0 && (unsigned)iy < (unsigned)Ho;
            const int pix = main ? a_base[it] + (iy >> ups) * Win + (a_col[it] >> ups) : a_bout[it] + iy * Wo + a_col[it];
            const int64_t off = ((int64_t)pix * ld + coff + jsw * 8) * (int64_t)sizeof(T);
            xptr[it] = ok ? sb + off : zp;
            xinc[it] = ok ? 128 : 0;
        }
    };
    // LDS-DMA instructions issue_A executes in this wave (a piece whose lanes are all halo / beyond the block is skipped)
    int na_wave = 0;
#pragma unroll
    for (int it = 0; it < XI; ++it)
        if (it * NW + wave < xp_used && __builtin_amdgcn_ballot_w64(a_col[it] >= 0) != 0) na_wave += 1;
    na_wave = __builtin_amdgcn_readfirstlane(na_wave);
    auto issue_A = [&](int par) __attribute__((always_inline)) {
        char* as = smem + par * A_BYTES;
#pragma unroll
        for (int it = 0; it < XI; ++it)
            // lanes of halo columns / rows beyond the block stay out: those LDS rows were zeroed once and never change
            if (!(UR_DXS_ABLATE & 1) && it * NW + wave < xp_used && a_col[it] >= 0) glds16(xptr[it], as + (it * NW + wave) * 1024);
    };
    // k offset (elements) into the packed weight row of step (k, dx)
    auto w_offset = [&](const Cur& k, int dx) __attribute__((always_inline)) {
        const int o = k.g < gmain ? k.blk * 9 * cb + (k.dy * 3 + dx) * cb + k.c * BK : 9 * C + (k.g - gmain) * BK;
        return __builtin_amdgcn_readfirstlane(o);
    };
    auto issue_W = [&](int par, int kofs) __attribute__((always_inline)) {
        char* ws = smem + 2 * A_BYTES + par * W_BYTES;
#pragma unroll
        for (int it = 0; it < WI; ++it)
            if (it * NW + wave < WP)
                glds16(wok[it] ? wbase[it] + (int64_t)kofs * (int64_t)sizeof(T) : zp, ws + (it * NW + wave) * 1024);
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, q = lane >> 4;
    int rho0[MREP];  // LDS row of this lane's pixel of fragment mf at dx = 0 (column -1 of its image row + pixel index)
#pragma unroll
    for (int mf = 0; mf < MREP; ++mf) {
        const int r = wm * (16 * MREP) + mf * 16 + l15;
        rho0[mf] = r + 2 * (r >> wsh);
    }
    auto compute = [&](int apar, int wpar, int dx) __attribute__((always_inline)) {
        const char* xs = smem + apar * A_BYTES;
        const char* ws = smem + 2 * A_BYTES + wpar * W_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = ((kk * 4 + q) ^ (l15 & 7)) << 4;
            vec8 wf[NREP], xf[MREP];
#pragma unroll
            for (int f = 0; f < NREP; ++f)
                wf[f] = *reinterpret_cast<const vec8*>(ws + (wn * 64 + f * 16 + l15) * 128 + c);
#pragma unroll
            for (int mf = 0; mf < MREP; ++mf) {
                const int rho = rho0[mf] + ((UR_DXS_ABLATE & 2) ? 1 : dx);
                xf[mf] = *reinterpret_cast<const vec8*>(xs + rho * 128 + (((kk * 4 + q) ^ (rho & 7)) << 4));
            }
#pragma unroll
            for (int mf = 0; mf < MREP; ++mf)
#pragma unroll
                for (int f = 0; f < NREP; ++f) acc[mf][f] = mfma16(wf[f], xf[mf], acc[mf][f]);
        }
    };

    // steps of this slice: three per main group, one per tail group
    const int nmain = max(0, min(gend, gmain) - gbeg);
    const int nsteps = 3 * nmain + (gend - gbeg - nmain);
    if (nsteps > 0) {
        // the halo columns (and the rows behind the block) of both pixel buffers are zero for the whole K loop: written here
        // once, never copied (no lane of issue_A touches them)
        for (int i = tid * 16; i < 2 * A_BYTES; i += NW * 64 * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // prologue: pixel block of the first group, weight tile of the first step
        Cur wc_ = cur_at(gbeg), ac_ = cur_at(gbeg);  // weight cursor (step), pixel cursor (group)
        int wdx = wc_.g < gmain ? 0 : 1;
        a_pointers(ac_, true);
        issue_A(0);
        issue_W(0, w_offset(wc_, wdx));
        // cursors -> the next step / the next group, their addresses computed ahead (in the shadow of the MFMAs below)
        auto w_advance = [&]() __attribute__((always_inline)) {
            if (wc_.g < gmain && wdx < 2) { wdx += 1; } else { cur_next(wc_); wdx = wc_.g < gmain ? 0 : 1; }
            wdx = __builtin_amdgcn_readfirstlane(wdx);
        };
        w_advance();
        int lkofs = (UR_DXS_ABLATE & 4) ? 0 : w_offset(wc_, wdx);
        cur_next(ac_);
        if (ac_.g < gend && !(UR_DXS_ABLATE & 4)) a_pointers(ac_, false);
        int cg = gbeg, cdx = cg < gmain ? 0 : 1, cgpar = 0;  // consumer: group, dx, pixel buffer
        int a_fly = 0;  // pixel-block copies of this wave issued in the previous iteration (may stay in flight one more step)
        for (int t = 0; t < nsteps; ++t) {
            // weights of step t were issued BEFORE the pixel block of the next group (same iteration): in the second step of
            // a main group that block may keep flying
            const int fly = (cdx == 1 && cg < gmain) ? a_fly : 0;
            if (fly == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (fly == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (fly == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (fly == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (fly == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool group_start = cg >= gmain || cdx == 0;
            const bool a_now = group_start && ac_.g < gend;  // ac_ == cg + 1
            if (t + 1 < nsteps) issue_W((t + 1) & 1, lkofs);
            if (a_now) issue_A(cgpar ^ 1);  // the buffer group cg - 1 was read from (before this barrier)
            a_fly = a_now ? na_wave : 0;
            __builtin_amdgcn_sched_barrier(0);
            compute(cgpar, t & 1, cdx);
            __builtin_amdgcn_sched_barrier(0);
            if (!(UR_DXS_ABLATE & 4)) {
                if (t + 2 < nsteps) {
                    w_advance();
                    lkofs = w_offset(wc_, wdx);
                }
                if (a_now) {
                    cur_next(ac_);
                    if (ac_.g < gend) a_pointers(ac_, false);
                }
            } else if (a_now) {
                cur_next(ac_);
            }
            if (cg < gmain && cdx < 2) { cdx += 1; } else { cg += 1; cgpar ^= 1; cdx = cg < gmain ? 0 : 1; }
            cg = __builtin_amdgcn_readfirstlane(cg);
            cdx = __builtin_amdgcn_readfirstlane(cdx);
            cgpar = __builtin_amdgcn_readfirstlane(cgpar);
        }
    }

    // ---- epilogue (igemm.hip's MF = 16 form) ----
    auto finish = [&](int m, int nc, float (&v)[16]) __attribute__((always_inline)) {
        if (p.splitk > 1) {
            if (m < p.M) {
                float4* pp = reinterpret_cast<float4*>(p.partial + ((int64_t)zidx * p.M + m) * p.ldp + nc);
#pragma unroll
                for (int i = 0; i < 4; ++i) pp[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            }
        } else {
            epilogue16<T>(p, reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout,
                          p.bias ? p.bias + (int64_t)zb * p.zbias : nullptr,
                          p.rowadd ? reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow : nullptr,
                          p.res ? reinterpret_cast<const T*>(p.res) + (int64_t)zb * p.zres : nullptr, m, nc, v,
                          HiLo<T>{p.res_lo ? reinterpret_cast<const lo_t<T>*>(p.res_lo) + (int64_t)zb * p.zres : nullptr,
                                  p.out_lo ? reinterpret_cast<lo_t<T>*>(p.out_lo) + (int64_t)zb * p.zout : nullptr},
                          p.out_vt ? reinterpret_cast<T*>(p.out_vt) + (int64_t)zb * p.zvt : nullptr);
        }
    };
    const int nc = n0 + wn * 64 + q * 16;
#pragma unroll
    for (int mf = 0; mf < MREP; ++mf) {
        const int m = m0 + wm * (16 * MREP) + mf * 16 + (lane & 15);
        float v[16];
#pragma unroll
        for (int f = 0; f < NREP; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[f * 4 + r] = acc[mf][f][r];
        finish(m, nc, v);
    }
}

// Shapes this kernel takes (the caller falls back to the lock-step kernel otherwise).
bool igemm_dxs_ok(const ur_igemm_desc& d, int bm) {
    if (d.taps != 9 || d.stride != 1 || d.pad != 1 || d.c1 != 0 || d.x1) return false;
    const int w = d.Wout;
    if (w < 8 || (w & (w - 1)) || (bm % w) || (d.M % w)) return false;
    if (d.ups && (d.Wout != 2 * d.Win || d.Hout != 2 * d.Hin)) return false;
    if (!d.ups && (d.Wout != d.Win || d.Hout != d.Hin)) return false;
    return true;
}

template <typename T, int BM, int BN, int WM, int WN>
static int dxs_launch_cfg(const ur_igemm_desc& d, hipStream_t s) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, d.zbatch * d.splitk);
    constexpr int lds = 2 * ((BM + BM / 4) * 128 + BN * 128);
    static_assert(lds <= 160 * 1024, "buffers do not fit the CU's LDS");
    static std::atomic<uint64_t> done{0};
    set_lds_limit_once(done, reinterpret_cast<const void*>(&igemm_dxs_kernel<T, BM, BN, WM, WN>), lds);
    hipLaunchKernelGGL((igemm_dxs_kernel<T, BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, s, d);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
static int dxs_launch_dtype(const ur_igemm_desc& d, hipStream_t s) {
    switch (d.tile) {
        case UR_TILE_128x320: return dxs_launch_cfg<T, 128, 320, 2, 5>(d, s);
        case UR_TILE_128x64_S2: case UR_TILE_128x64: return dxs_launch_cfg<T, 128, 64, 4, 1>(d, s);
        case UR_TILE_64x64_S2: case UR_TILE_64x64: case UR_TILE_64x64_S4: return dxs_launch_cfg<T, 64, 64, 4, 1>(d, s);
        case UR_TILE_128x128: case UR_TILE_128x128_S3: return dxs_launch_cfg<T, 128, 128, 2, 2>(d, s);
        case UR_TILE_256x128: return dxs_launch_cfg<T, 256, 128, 4, 2>(d, s);
    }
    return UR_E_UNSUPPORTED;
}

// tile height of the tiles this file instantiates (0: none)
int igemm_dxs_tile_bm(int tile) {
    switch (tile) {
        case UR_TILE_128x320: case UR_TILE_128x64_S2: case UR_TILE_128x64: case UR_TILE_128x128: case UR_TILE_128x128_S3: return 128;
        case UR_TILE_64x64_S2: case UR_TILE_64x64: case UR_TILE_64x64_S4: return 64;
        case UR_TILE_256x128: return 256;
    }
    return 0;
}

// main pass (the caller, igemm.hip, runs the shared split-K second pass)
int igemm_dxs_launch(const ur_igemm_desc& d, hipStream_t s) {
    if (d.dtype == UR_DT_F16) return dxs_launch_dtype<f16>(d, s);
    if (d.dtype == UR_DT_BF16) return dxs_launch_dtype<bf16>(d, s);
    return UR_E_BADARG;
}

}  // namespace ur
