// Implicit GEMM on MFMA for gfx950: conv3x3 (stride 1/2, fused nearest-2x upsample, fused channel
// concat of two sources), conv1x1 / Linear, and the batched transposed V projection, all with fused
// epilogues (bias, per-sample time-embedding add, SiLU, GEGLU, residual / feature-exchange add, scale).
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] ),  K chunked by 64 (one 128-byte line per row).
//
// Data path per workgroup (256 threads = 4 waves):
//   * both operand tiles are copied global -> LDS with `global_load_lds` (16 B per lane, no VGPR
//     round trip), double buffered, one barrier per K chunk;
//   * a tile row is 128 B; its eight 16-B chunks are XOR-swizzled by (row & 7) -- applied on the global
//     SOURCE address because the LDS-DMA destination is lane-linear -- so that the ds_read_b128 fragment
//     reads of the 16-lane groups hit 16 distinct bank slots;
//   * the image (im2col) gather is done by the loader: each lane owns fixed output pixels, and per
//     K chunk only adds the (dy,dx) tap offset; out-of-image taps, rows >= M and rows >= N read a
//     zero page;
//   * MFMA v_mfma_f32_16x16x32_{f16,bf16} with the WEIGHT tile as operand A and the PIXEL tile as
//     operand B, so a lane ends up with consecutive output channels of ONE pixel; the weight rows are
//     permuted while loading so that those channels are 16 consecutive ones -> 32-byte vector
//     stores/loads per lane and full 128-byte lines per pixel row in the epilogue.
#include <cstdlib>

#include "igemm_epi.h"

namespace ur {

// MF = 16: v_mfma_f32_16x16x32 (the original formulation, described above).
// MF = 32: v_mfma_f32_32x32x16: the same wave tiles with half the MFMA instructions: per 64-deep K chunk a wave issues
// (BM/WM/32) x (BN/WN/32) x 4 MFMAs of 32 cycles instead of twice as many of 16.  (Both shapes sustain the same
// 1.6-1.85 PFLOP/s on random operands, profiles/r03_mfma_rate.txt; round 2's "27-cycle 16x16x32" was a micro-benchmark
// artefact.  The M32 builds measure 4-13 % SLOWER on the heavy problems: the loop is not MFMA-issue bound.)  Same LDS image except for two details: a fragment
// read now spans 32 consecutive tile rows per half-wave, so the 16-byte chunk swizzle key is ((row >> 1) & 7) instead
// of (row & 7) (conflict-free for ds_read_b128's 16-lane service groups, tools/lds_bank_check.py), and the weight
// rows of a 32-row block are permuted so that the 16 accumulator registers of a lane (MFMA rows 8g + 4h + r) are 16
// CONSECUTIVE output channels 16h + 4g + r of one pixel -- the epilogue is shared.
// NL = 0: every wave copies its share of both tiles and multiplies (the original formulation).
// NL > 0: WAVE SPECIALISATION.  The K loop of the symmetric form costs MFMA time PLUS LDS-DMA issue time (ablation
// builds): all waves of a workgroup are phase-locked by the chunk barrier, so they all issue their
// 1-KiB DMA pieces (60-185 cycles of issue each) at the same moment, with every matrix pipe idle, and then all multiply.
// Here NL extra waves do NOTHING but the loader's bookkeeping and DMA issue for the whole tile, while the WM x WN
// consumer waves run an uninterrupted MFMA stream; the barrier protocol is unchanged (one s_barrier per chunk: the
// loader has waited for chunk t, the consumers have left the buffer chunk t+1 goes into).
template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, bool CONV, int MF, int NL>
__global__ void __launch_bounds__((WM * WN + NL) * 64) igemm_kernel(const ur_igemm_desc p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int NWC = WM * WN;              // consumer waves (own the output tile)
    constexpr int NW = NL > 0 ? NL : NWC;     // waves that copy: the NL loader waves, or everybody
    static_assert(NL == 0 || NSTAGE > 0, "loader waves use the LDS-DMA ring");
    constexpr int MREP = BM / WM / 16;
    constexpr int NREP = BN / WN / 16;
    static_assert(MF == 16 || MF == 32, "MFMA shape");
    static_assert(MF == 32 || NREP == 4, "a wave spans 64 output columns (epilogue layout)");
    static_assert(MF == 16 || ((BM / WM) % 32 == 0 && (BN / WN) % 32 == 0), "32x32 MFMA: wave tile in multiples of 32");
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;  // 32x32 blocks of a wave tile (MF == 32)
    constexpr int XT_BYTES = BM * 128;
    constexpr int WT_BYTES = BN * 128;
    constexpr int STAGE = XT_BYTES + WT_BYTES;
    constexpr bool REGSTAGE = NSTAGE < 0;  // register-staged loader with 2 LDS buffers
    constexpr int XI = (BM / 8 + NW - 1) / NW;  // LDS-DMA instructions per wave per X tile (8 rows each)
    constexpr int WI = (BN / 8 + NW - 1) / NW;
    static_assert(NSTAGE <= 2 || ((BM / 8) % NW == 0 && (BN / 8) % NW == 0), "counted vmcnt needs uniform loads per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = tid >> 6;
    const bool is_loader = NL == 0 || wave_id >= NWC;    // wave-uniform roles
    const bool is_consumer = NL == 0 || wave_id < NWC;
    const int wave = NL > 0 ? (is_loader ? wave_id - NWC : 0) : wave_id;  // index among the copying waves
    const int wm = (wave_id % NWC) / WN, wn = (wave_id % NWC) % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware mapping: logical ids are tile-major within one z slice, n fastest, so an XCD works on a
    // contiguous run of m-tiles x all n-tiles (activation rows fetched once per XCD, weights shared in its L2)
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
    const int zidx = lid / gridDim.x;
    const int tid_xy = lid - zidx * gridDim.x;
    const int tile_n = tid_xy % tiles_n;
    const int tile_m = tid_xy / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kt_total = p.K / BK;
    int kbeg = 0, kend = kt_total, zb = zidx;  // grid.z = zbatch * splitk, split index fastest
    if (p.splitk > 1) {
        zb = zidx / p.splitk;
        const int ks = zidx - zb * p.splitk;
        const int per = (kt_total + p.splitk - 1) / p.splitk;
        kbeg = ks * per;
        kend = min(kt_total, kbeg + per);
    }
    const char* x0 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.x0) + (int64_t)(zb / p.zx_div) * p.zx);
    const char* x1 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.x1) + (int64_t)zb * p.zx1);
    const char* wp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.w) + (int64_t)zb * p.zw);
    const char* t0 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t0) + (int64_t)zb * p.zt0);  // 1x1 tail sources
    const char* t1 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t1) + (int64_t)zb * p.zt1);
    // swizzled source chunk of this lane's 16 bytes in LDS-DMA piece `piece` (8 tile rows, row = 8 * piece + lane / 8):
    // key = row & 7 (MF 16) or (row >> 1) & 7 (MF 32)
    auto jsw = [&](int piece) __attribute__((always_inline)) {
        return MF == 16 ? ((lane & 7) ^ (lane >> 3)) : ((lane & 7) ^ ((4 * piece + (lane >> 4)) & 7));
    };
    // padding rows: this (workgroup, wave)'s own 128-byte line of the zero region (one hot line would be served to all CUs
    // by one L2 channel)
    const unsigned zbytes = p.zero_page_bytes >= 256 ? (unsigned)p.zero_page_bytes : 256u;
    const char* zp = reinterpret_cast<const char*>(p.zero_page) + ((((unsigned)lid * 16u + (unsigned)wave_id) * 128u) & (zbytes - 128u)) + (lane & 7) * 16;

    // ---- per-lane row bookkeeping (fixed over the K loop) ----
    // The K axis is walked as segments (tap, source) of c_src/64 chunks.  Per segment every lane
    // recomputes one 64-bit pointer per owned row (pixel + tap offset, or the zero page when the tap
    // falls outside the image / the row is >= M); inside a segment a chunk step is `ptr += inc`
    // (inc = 128 bytes, or 0 for zero-page rows).
    int xa[XI], xy[XI], xx[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int r = (it * NW + wave) * 8 + (lane >> 3);
        const int m = m0 + r;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            xa[it] = b * p.Hin * p.Win;
            xy[it] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
            xx[it] = ox * p.stride - p.pad;
        } else {
            xa[it] = (m < p.M) ? m : -1;
            xy[it] = 0;
            xx[it] = 0;
        }
    }
    const char* xptr[XI];
    int xinc[XI];
    const char* wptr[WI];
    int winc[WI];
#pragma unroll
    for (int it = 0; it < WI; ++it) {
        const int r = (it * NW + wave) * 8 + (lane >> 3);  // LDS row of the tile
        int sem;
        if (MF == 16) {
            const int rho = r & 63;
            // LDS row (f, i) = f*16 + i holds semantic column (i>>2)*16 + f*4 + (i&3) of its 64-group
            sem = (r & ~63) | (((rho >> 2) & 3) << 4) | ((rho >> 4) << 2) | (rho & 3);
        } else {
            const int rho = r & 31;
            // LDS row 8g + 4h + rr of a 32-block holds semantic column 16h + 4g + rr
            sem = (r & ~31) | (((rho >> 2) & 1) << 4) | ((rho >> 3) << 2) | (rho & 3);
        }
        const int n = n0 + sem;
        const bool ok = n < p.N;
        const int64_t off = ((int64_t)n * p.ldw + (int64_t)kbeg * BK + jsw(it * NW + wave) * 8) * (int64_t)sizeof(T);
        wptr[it] = ok ? wp + off : zp;
        winc[it] = ok ? 128 : 0;
    }

    // segment state of the loader (all wave-uniform).  Tap-outer order (cblock == 0): segments (tap, source) of
    // c_src/64 chunks.  Block-outer order (cblock > 0, one source): segments (channel block, tap) of cblock/64 chunks.
    // Then, if ct0 > 0, the 1x1 tail: one segment per tail source (seg_src = 2, 3), read at the centre tap.
    // (every descriptor field the lambdas below choose between is copied into a scalar first: a runtime select between
    // p.fieldA and p.fieldB makes the compiler materialise the whole kernel-argument struct in scratch)
    const int pc0 = p.c0, pc1 = p.c1, pct0 = p.ct0, pct1 = p.ct1;
    const int64_t pldx0 = p.ldx0, pldx1 = p.ldx1, pldt0 = p.ldt0, pldt1 = p.ldt1;
    const int pHin = p.Hin << p.ups, pWin = p.Win << p.ups, pups = p.ups, pW = p.Win;
    const int segs_per_tap = (pc1 > 0) ? 2 : 1;
    const int cblk = p.cblock;
    int seg_tap, seg_src, seg_left, seg_coff = 0;  // seg_coff: first channel of the current block
    auto uniform_i64 = [](int64_t v) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto uniform_ptr = [&](const char* q) __attribute__((always_inline)) {
        return reinterpret_cast<const char*>(uniform_i64(reinterpret_cast<int64_t>(q)));
    };
    // base pointer / leading dimension of the current source: loop-carried state assigned where seg_src changes (a
    // four-way choice by seg_src would be lowered to a lookup table in scratch)
    const char* seg_base = x0;
    int64_t seg_ld = pldx0;
    // pointers of the segment (seg_base, seg_tap, seg_coff), advanced by cc channels
    auto set_pointers = [&](int cc) __attribute__((always_inline)) {
        const char* sb = seg_base;
        const int64_t ld = seg_ld;
        const int dy = seg_tap / 3, dx = seg_tap - dy * 3;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            int pix;
            bool ok;
            if (CONV) {
                const int iy = xy[it] + dy, ix = xx[it] + dx;
                ok = ((unsigned)iy < (unsigned)pHin) && ((unsigned)ix < (unsigned)pWin);
                pix = xa[it] + (iy >> pups) * pW + (ix >> pups);
            } else {
                ok = xa[it] >= 0;
                pix = xa[it];
            }
            const int64_t off = ((int64_t)pix * ld + seg_coff + cc + jsw(it * NW + wave) * 8) * (int64_t)sizeof(T);
            xptr[it] = ok ? sb + off : zp;
            xinc[it] = ok ? 128 : 0;
        }
    };
    {
        const int Cin = pc0 + pc1;
        const int kglob = kbeg * BK;
        int cc;
        if (CONV && kglob >= 9 * Cin) {  // a split-K slice that starts inside the tail
            cc = kglob - 9 * Cin;
            seg_src = 2; seg_base = t0; seg_ld = pldt0;
            seg_left = (pct0 - cc) / BK;
            if (cc >= pct0) { seg_src = 3; cc -= pct0; seg_base = t1; seg_ld = pldt1; seg_left = (pct1 - cc) / BK; }
            seg_tap = 4;
        } else if (cblk > 0) {
            const int blk = kglob / (9 * cblk);
            const int rem = kglob - blk * 9 * cblk;
            seg_tap = rem / cblk;
            cc = rem - seg_tap * cblk;
            seg_src = 0;
            seg_coff = blk * cblk;
            seg_left = (cblk - cc) / BK;
        } else {
            seg_tap = kglob / Cin;
            cc = kglob - seg_tap * Cin;
            seg_src = (cc >= pc0) ? 1 : 0;
            if (seg_src) { cc -= pc0; seg_base = x1; seg_ld = pldx1; }
            seg_left = ((seg_src ? pc1 : pc0) - cc) / BK;
        }
        set_pointers(cc);
    }

    auto next_segment = [&]() __attribute__((always_inline)) {
        bool to_tail = false;
        if (seg_src >= 2) {
            seg_src = 3;  // tail source 0 -> 1 (or past the end of K: never loaded)
            seg_base = t1; seg_ld = pldt1; seg_left = pct1 / BK;
        } else if (cblk > 0) {
            seg_tap += 1;
            if (seg_tap == 9) { seg_tap = 0; seg_coff += cblk; }
            seg_left = cblk / BK;
            to_tail = CONV && seg_coff >= pc0;
        } else {
            seg_src += 1;
            if (seg_src >= segs_per_tap) { seg_src = 0; seg_tap += 1; }
            seg_base = seg_src ? x1 : x0;
            seg_ld = seg_src ? pldx1 : pldx0;
            seg_left = (seg_src ? pc1 : pc0) / BK;
            to_tail = CONV && seg_tap == 9;
        }
        if (to_tail) {  // centre tap of the tail sources (stride 1, no upsampling: output pixel = input pixel)
            seg_src = 2; seg_base = t0; seg_ld = pldt0; seg_left = pct0 / BK;
            seg_tap = 4; seg_coff = 0;
        }
        // the segment state is wave-uniform by construction; say so, or it lives in VGPRs (and spills)
        seg_tap = __builtin_amdgcn_readfirstlane(seg_tap);
        seg_src = __builtin_amdgcn_readfirstlane(seg_src);
        seg_left = __builtin_amdgcn_readfirstlane(seg_left);
        seg_coff = __builtin_amdgcn_readfirstlane(seg_coff);
        seg_base = uniform_ptr(seg_base);
        seg_ld = uniform_i64(seg_ld);
        set_pointers(0);
    };

    // issue the LDS-DMA copies of the loader's current chunk into `buf`, then advance by one chunk
    // (-DUR_ABLATE=1 builds a kernel without the copies, =2 one without the MFMAs: the two ablations behind the
    // "MFMA time + loader time" model of DESIGN.md section 4; never defined in the product build)
    auto stage = [&](int buf) {
#if defined(UR_ABLATE) && UR_ABLATE == 1
        if (buf >= 0) { seg_left -= 1; if (seg_left == 0) next_segment(); return; }
#endif
        char* xs = smem + buf * STAGE;
        char* ws = xs + XT_BYTES;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
#if defined(UR_ABLATE) && UR_ABLATE == 3
            // timing-only upper bound of "the three dx taps share one staged pixel block" (DESIGN.md section 4, round 4): the
            // pixel tile is copied for one tap in three, the other two multiply stale LDS contents (results are garbage)
            if (it * NW + wave < BM / 8 && !(CONV && seg_src < 2 && (seg_tap % 3) != 0)) glds16(xptr[it], xs + (it * NW + wave) * 1024);
#else
            if (it * NW + wave < BM / 8) glds16(xptr[it], xs + (it * NW + wave) * 1024);
#endif
            xptr[it] += xinc[it];
        }
#pragma unroll
        for (int it = 0; it < WI; ++it) {
            if (it * NW + wave < BN / 8) glds16(wptr[it], ws + (it * NW + wave) * 1024);
            wptr[it] += winc[it];
        }
        seg_left -= 1;
        if (seg_left == 0) next_segment();
    };

    // Register-staged alternative (NSTAGE < 0): plain 16-byte global loads into VGPRs, issued BEFORE the MFMAs of
    // the current chunk and written to the other LDS buffer AFTER them (ds_write_b128).  Same LDS image as the DMA
    // path (the swizzle lives in the source address), so the MFMA side is unchanged.  A global_load costs a few
    // issue cycles against 60-185 for an LDS-DMA piece (MI355X_MICROARCH.md), which is what bounds small tiles.
    u32x4 xr[XI], wr[WI];
    auto gload = [&]() {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            if (it * NW + wave < BM / 8) xr[it] = *reinterpret_cast<const u32x4*>(xptr[it]);
            xptr[it] += xinc[it];
        }
#pragma unroll
        for (int it = 0; it < WI; ++it) {
            if (it * NW + wave < BN / 8) wr[it] = *reinterpret_cast<const u32x4*>(wptr[it]);
            wptr[it] += winc[it];
        }
        seg_left -= 1;
        if (seg_left == 0) next_segment();
    };
    auto lstore = [&](int buf) {
        char* xs = smem + buf * STAGE + lane * 16;
        char* ws = xs + XT_BYTES;
#pragma unroll
        for (int it = 0; it < XI; ++it)
            if (it * NW + wave < BM / 8) *reinterpret_cast<u32x4*>(xs + (it * NW + wave) * 1024) = xr[it];
#pragma unroll
        for (int it = 0; it < WI; ++it)
            if (it * NW + wave < BN / 8) *reinterpret_cast<u32x4*>(ws + (it * NW + wave) * 1024) = wr[it];
    };

    f32x4 acc[MF == 16 ? MREP : 1][MF == 16 ? NREP : 1];
    f32x16 acc32[MF == 32 ? MI : 1][MF == 32 ? NI : 1];
    if constexpr (MF == 16) {
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc32[i][j][v] = 0.f;
    }

    const int l15 = lane & 15, q = lane >> 4;
    const int l31 = lane & 31, hh = lane >> 5;
    auto compute = [&](int buf) {
#if defined(UR_ABLATE) && UR_ABLATE == 2
        if (buf >= 0) return;
#endif
        const char* xs = smem + buf * STAGE;
        const char* ws = xs + XT_BYTES;
        if constexpr (MF == 32) {
            // four k16 steps per chunk; lane (row l31 of a 32-block, half hh) reads chunk 2 s + hh of its row, swizzled
            // by ((row >> 1) & 7) = (l31 >> 1) & 7 (block bases are multiples of 32)
            const int key = (l31 >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = ((2 * s + hh) ^ key) << 4;
                vec8 wf[NI], xf[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[ni] = *reinterpret_cast<const vec8*>(ws + (wn * (32 * NI) + ni * 32 + l31) * 128 + c);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    xf[mi] = *reinterpret_cast<const vec8*>(xs + (wm * (32 * MI) + mi * 32 + l31) * 128 + c);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc32[mi][ni] = mfma32(wf[ni], xf[mi], acc32[mi][ni]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = ((kk * 4 + q) ^ (l15 & 7)) << 4;  // swizzled 16-B chunk of this lane
                vec8 wf[NREP], xf[MREP];
#pragma unroll
                for (int f = 0; f < NREP; ++f)
                    wf[f] = *reinterpret_cast<const vec8*>(ws + (wn * 64 + f * 16 + l15) * 128 + c);
#pragma unroll
                for (int mf = 0; mf < MREP; ++mf)
                    xf[mf] = *reinterpret_cast<const vec8*>(xs + (wm * (16 * MREP) + mf * 16 + l15) * 128 + c);
#pragma unroll
                for (int mf = 0; mf < MREP; ++mf)
#pragma unroll
                    for (int f = 0; f < NREP; ++f) acc[mf][f] = mfma16(wf[f], xf[mf], acc[mf][f]);
            }
        }
    };

    const int nk = kend - kbeg;
    if constexpr (REGSTAGE) {
        if (nk > 0) {
            gload();
            lstore(0);
            __syncthreads();
            for (int t = 0; t < nk; ++t) {
                if (t + 1 < nk) gload();           // chunk t+1 in flight while chunk t is multiplied
                compute(t & 1);
                if (t + 1 < nk) lstore((t + 1) & 1);  // the buffer every wave left before the last barrier
                __syncthreads();
            }
        }
    } else if (nk > 0) {
        // NSTAGE-deep LDS ring, prefetch distance D = NSTAGE-1 chunks, ONE barrier per chunk:
        //   wait (counted vmcnt: only chunk t must have landed, newer ones stay in flight) -> s_barrier
        //   -> issue chunk t+D into the buffer every wave finished reading before that barrier -> MFMAs.
        // Raw s_barrier + inline-asm vmcnt(N): __syncthreads() would drain the LDS-DMA queue (vmcnt(0)).
        constexpr int D = (NSTAGE > 0 ? NSTAGE : 2) - 1;
        constexpr int LOADS = XI + WI;  // LDS-DMA instructions per wave per chunk
        if constexpr (NL > 0) {
            // Two role-specific loops with the SAME barrier sequence (nk s_barriers each): written apart so that the
            // loader's pointer arrays and the consumers' accumulators are never live at the same program point (one
            // merged loop made the register allocator hold both: 168+ VGPRs and scratch spills).
            if (is_loader) {
#pragma unroll
                for (int s = 0; s < D; ++s)
                    if (s < nk) stage(s);
                int nbuf = D % NSTAGE;
                for (int t = 0; t < nk; ++t) {
                    const int newer = min(D - 1, nk - 1 - t);
                    if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
                    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (t + D < nk) stage(nbuf);
                    nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
                }
                return;  // loader waves are done (no barrier follows the K loop)
            }
            int buf = 0;
            for (int t = 0; t < nk; ++t) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                compute(buf);
                buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
            }
        } else {
#pragma unroll
            for (int s = 0; s < D; ++s)
                if (s < nk) stage(s);
            int buf = 0, nbuf = D % NSTAGE;
            for (int t = 0; t < nk; ++t) {
                const int newer = min(D - 1, nk - 1 - t);  // chunks issued after chunk t that may stay in flight
                if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
                else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t + D < nk) stage(nbuf);
                compute(buf);
                buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
                nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
            }
        }
    }

    if (!is_consumer) return;  // loader waves are done (no barrier follows the K loop)
    // ---- epilogue: a lane holds 16 consecutive output channels of one pixel ----
    //   MF 16: lane (j = lane&15, q = lane>>4): pixel row j of a 16-row block, channels q*16 .. q*16+15 of the wave's 64
    //   MF 32: lane (j = lane&31, h = lane>>5): pixel row j of a 32-row block, channels h*16 .. h*16+15 of a 32-block
    auto finish = [&](int m, int nc, float (&v)[16]) __attribute__((always_inline)) {
        if (p.splitk > 1) {
            if (m < p.M) {
                float4* pp = reinterpret_cast<float4*>(p.partial + ((int64_t)zidx * p.M + m) * p.ldp + nc);
#pragma unroll
                for (int i = 0; i < 4; ++i) pp[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            }
        } else {
            epilogue16<T, CONV>(p, reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout,
                                p.bias ? p.bias + (int64_t)zb * p.zbias : nullptr,
                                p.rowadd ? reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow : nullptr,
                                p.res ? reinterpret_cast<const T*>(p.res) + (int64_t)zb * p.zres : nullptr, m, nc, v,
                                HiLo<T>{p.res_lo ? reinterpret_cast<const lo_t<T>*>(p.res_lo) + (int64_t)zb * p.zres : nullptr,
                                        p.out_lo ? reinterpret_cast<lo_t<T>*>(p.out_lo) + (int64_t)zb * p.zout : nullptr},
                                (!CONV && p.out_vt) ? reinterpret_cast<T*>(p.out_vt) + (int64_t)zb * p.zvt : nullptr);
        }
    };
    if constexpr (MF == 32) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int m = m0 + wm * (32 * MI) + mi * 32 + l31;
                const int nc = n0 + wn * (32 * NI) + ni * 32 + hh * 16;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc32[mi][ni][r];
                finish(m, nc, v);
            }
    } else {
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
}

// Second pass of split-K: sum the fp32 slabs and run the epilogue.  One thread per (row, 16 columns).  Round 6: the epilogue's
// operand loads are issued BEFORE the slab loads and the slabs four at a time (igemm_epi.h, epi_preload / epi_apply): the kernel has
// ~2.5 waves per SIMD on the small maps and its run time was the length of its chain of dependent loads.  Same sums in the same
// order: bit-identical results.
// PIPE = true only for SMALL second passes (< REDUCE_PIPE_MAX_THREADS threads: the 8x8 level, everything at cfg 2): there the launch is
// under one wave per SIMD and pure latency; from the 16x16 level up the kernel is L2 / HBM-bound and the extra loads in flight made it
// 5 % SLOWER in the step (profiles/r06_splitk_reduce_ab.txt, r06_reduce_prof.txt), so those keep the slab-after-slab loop.
constexpr int64_t REDUCE_PIPE_MAX_THREADS = 65536;
template <typename T, bool PIPE>
__global__ void __launch_bounds__(256) igemm_splitk_reduce(const ur_igemm_desc p) {
    const int groups = (int)(p.ldp / 16);
    const int64_t total = (int64_t)p.M * groups;
    const int zb = blockIdx.y;
    T* outz = reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout;
    const float* biasz = p.bias ? p.bias + (int64_t)zb * p.zbias : nullptr;
    const T* rowz = p.rowadd ? reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow : nullptr;
    const T* resz = p.res ? reinterpret_cast<const T*>(p.res) + (int64_t)zb * p.zres : nullptr;
    const lo_t<T>* rloz = p.res_lo ? reinterpret_cast<const lo_t<T>*>(p.res_lo) + (int64_t)zb * p.zres : nullptr;
    lo_t<T>* oloz = p.out_lo ? reinterpret_cast<lo_t<T>*>(p.out_lo) + (int64_t)zb * p.zout : nullptr;
    T* vtz = p.out_vt ? reinterpret_cast<T*>(p.out_vt) + (int64_t)zb * p.zvt : nullptr;
    const int64_t sstride = (int64_t)p.M * p.ldp;  // floats between two slabs of one problem
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / groups);
        const int nc = (int)(idx - (int64_t)m * groups) * 16;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
        const float* base = p.partial + ((int64_t)zb * p.splitk * p.M + m) * p.ldp + nc;
        if constexpr (!PIPE) {
            for (int z = 0; z < p.splitk; ++z) {
                const float4* pp = reinterpret_cast<const float4*>(base + (int64_t)z * sstride);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 a = pp[i];
                    v[4 * i] += a.x; v[4 * i + 1] += a.y; v[4 * i + 2] += a.z; v[4 * i + 3] += a.w;
                }
            }
            if (nc < p.n_store || nc < p.N) epilogue16<T>(p, outz, biasz, rowz, resz, m, nc, v, HiLo<T>{rloz, oloz}, vtz);
            continue;
        }
        const bool split = epi_split_ok<T>(p, nc, vtz);
        EpiOperands<T> op;
        if (split) epi_preload<T>(p, op, biasz, rowz, resz, rloz, m, nc, p.zero_page);
        for (int z0 = 0; z0 < p.splitk; z0 += 4) {  // four slabs (16 loads) in flight, summed in slab order
            float4 a[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4* pp = reinterpret_cast<const float4*>(base + (int64_t)min(z0 + u, p.splitk - 1) * sstride);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[u][i] = pp[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (z0 + u < p.splitk) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[4 * i] += a[u][i].x; v[4 * i + 1] += a[u][i].y; v[4 * i + 2] += a[u][i].z; v[4 * i + 3] += a[u][i].w;
                    }
                }
        }
        if (split)
            epi_apply<T>(p, op, outz, biasz != nullptr, rowz != nullptr, resz != nullptr, rloz != nullptr, oloz, m, nc, v);
        else if (nc < p.n_store || nc < p.N)
            epilogue16<T>(p, outz, biasz, rowz, resz, m, nc, v, HiLo<T>{rloz, oloz}, vtz);
    }
}


// Second pass of split-K FUSED with the GroupNorm (+ SiLU) that follows the conv: one workgroup per (problem z, sample,
// group) sums the fp32 slabs of its [rows][cpg] strip, adds bias and the per-sample time-embedding row, rounds to the
// storage dtype (the value the two-kernel path would have stored and re-read), takes the group statistics in fp32 (fixed
// order), normalises and writes ONLY the normalised tensor: the conv output itself is never materialised.  This is the
// conv1 -> norm2 -> SiLU hand-off inside a ResnetBlock2D at the 16x16 / 8x8 levels (models/unet_2d_blocks.py:1100-1111),
// where conv1 runs split-K and its output has no other consumer.  Replaces igemm_splitk_reduce + gn_fused_kernel.
constexpr int RGN_THREADS = 1024, RGN_MAXQ = 4;  // <= 4 channel quads per thread: rows * cpg <= 16384
template <typename T>
__global__ void __launch_bounds__(RGN_THREADS) igemm_splitk_reduce_gn(const ur_igemm_desc p, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, int64_t zgn, float eps,
                                                                       int groups, int silu, int rows) {
    __shared__ float red[2][RGN_THREADS / 64];
    // the groups of one sample on ONE XCD: neighbouring groups share the 128-byte lines of the slabs (a strip is a 4 * cpg-byte
    // run of every row), which would otherwise be fetched into two or three L2s
    const int lid_ = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int g = lid_ % gridDim.x, b = (lid_ / gridDim.x) % gridDim.y, zb = lid_ / (gridDim.x * gridDim.y);
    const int cpg = p.N / groups, qpr = cpg >> 2;  // channel quads per row of the strip
    const int nq = rows * qpr;
    const int c0 = g * cpg;
    const float* biasz = p.bias ? p.bias + (int64_t)zb * p.zbias : nullptr;
    const T* rowz = p.rowadd ? reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow + (int64_t)b * p.ld_rowadd : nullptr;
    float v[RGN_MAXQ][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < RGN_MAXQ; ++i) {
        const int e = threadIdx.x + i * RGN_THREADS;
        if (e < nq) {
            const int r = e / qpr, c = c0 + (e - r * qpr) * 4;
            const int m = b * rows + r;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ks = 0; ks < p.splitk; ++ks) {
                const float4 t = *reinterpret_cast<const float4*>(p.partial + (((int64_t)zb * p.splitk + ks) * p.M + m) * p.ldp + c);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (biasz) x[k] += biasz[c + k];
                if (rowz) x[k] += to_f(rowz[c + k]);
                x[k] = to_f(from_f<T>(x[k] * p.out_scale));  // what the unfused path stores and GroupNorm re-reads
                v[i][k] = x[k];
                s1 += x[k];
                s2 = fmaf(x[k], x[k], s2);
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < RGN_THREADS / 64; ++w) { t1 += red[0][w]; t2 += red[1][w]; }
    const float n = (float)rows * (float)cpg;
    const float mean = t1 / n;
    const float rstd = rsqrtf(fmaxf(t2 / n - mean * mean, 0.f) + eps);
    const float* gz = gamma + (int64_t)zb * zgn;
    const float* bz = beta + (int64_t)zb * zgn;
    T* outz = reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout;
#pragma unroll
    for (int i = 0; i < RGN_MAXQ; ++i) {
        const int e = threadIdx.x + i * RGN_THREADS;
        if (e < nq) {
            const int r = e / qpr, c = c0 + (e - r * qpr) * 4;
            const int m = b * rows + r;
            T y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = rstd * gz[c + k];
                const float t = (v[i][k] - mean) * a + bz[c + k];
                y[k] = from_f<T>(silu ? silu_f(t) : t);
            }
            *reinterpret_cast<uint2*>(outz + (int64_t)m * p.ldc + c) = *reinterpret_cast<const uint2*>(y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct TileCfg { int bm, bn, stages; };
// index = UR_TILE_* (include/ur_kernels.h)
static const TileCfg kTiles[UR_TILE_COUNT] = {{0, 0, 0},      {128, 128, 2}, {128, 64, 3}, {64, 64, 3},
                                              {128, 128, 3}, {128, 64, 2},  {64, 64, 4},  {64, 64, 2},
                                              {256, 128, 2}, {128, 320, 2}, {128, 256, 2}, {256, 256, 2},
                                              {64, 64, -2},  {128, 64, -2}, {128, 128, -2}, {128, 320, -2},
                                              {256, 128, -2}, {64, 64, 2},   {128, 64, 2},   {64, 64, 3},
                                              {64, 128, 2},  {64, 64, 4},
                                              // 32x32x16-MFMA builds (UR_TILE_*_M32)
                                              {128, 320, 2}, {128, 128, 2}, {128, 64, 2}, {128, 64, 3}, {64, 64, 2},
                                              {64, 64, 3},   {256, 256, 2}, {256, 128, 2}, {128, 256, 2},
                                              // wave-specialised builds: dedicated loader waves (UR_TILE_*_L<n>)
                                              {128, 320, 2}, {128, 320, 2}, {128, 128, 2}, {128, 128, 3}, {128, 64, 2},
                                              {128, 64, 3},  {64, 64, 3},   {256, 128, 2}, {256, 256, 2}, {128, 256, 2},
                                              {128, 256, 3}, {128, 320, 2}, {256, 320, 2}, {128, 160, 2},
                                              {128, 160, 3}, {64, 320, 2},
                                              // weight-streaming conv (wsconv.hip): 4 and 8 waves
                                              {128, 320, 2}, {128, 320, 2},
                                              // 8-wave ping-pong builds (igemm_pp.hip)
                                              {128, 320, 5}, {128, 320, 4}, {256, 128, 5}, {128, 256, 5}, {256, 256, 4},
                                              {128, 128, 5}, {256, 320, 4},
                                              // round 6: few waves, big per-wave tiles (UR_TILE_*_W4_M32 / _W8_M32)
                                              {256, 160, 2}, {256, 320, 2}, {128, 320, 2}, {256, 128, 2}, {256, 256, 2}, {256, 320, 2}};

static int pick_tile(const ur_igemm_desc& d) {
    // Cost model: the busiest CU runs ceil(workgroups / 256) tiles; bigger tiles have a better
    // MFMA : LDS-read ratio.  The Python host normally passes an explicit tile from its tuning table.
    const double eff[4] = {0, 1.0, 0.85, 0.6};
    int best = UR_TILE_64x64;
    double best_cost = 1e30;
    for (int t = 1; t <= 3; ++t) {
        const int64_t tm = (d.M + kTiles[t].bm - 1) / kTiles[t].bm;
        const int64_t tn = (d.N + kTiles[t].bn - 1) / kTiles[t].bn;
        const int64_t wgs = tm * tn * (d.zbatch > 1 ? d.zbatch : 1) * (d.splitk > 1 ? d.splitk : 1);
        const double cost = (double)((wgs + 255) / 256) * kTiles[t].bm * kTiles[t].bn / eff[t];
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, bool CONV, int MF, int NL>
static void ensure_lds_limit(int lds) {
    static std::atomic<uint64_t> done{0};  // per (instantiation, device), see set_lds_limit_once
    set_lds_limit_once(done, reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, WM, WN, NSTAGE, CONV, MF, NL>), lds);
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, int MF = 16, int NL = 0>
static int launch_cfg(const ur_igemm_desc& d, hipStream_t s, bool reduce) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, d.zbatch * d.splitk);
    const size_t lds = (NSTAGE > 0 ? NSTAGE : 2) * (BM + BN) * 128;
    hipError_t e;
    if (d.taps == 9) {
        ensure_lds_limit<T, BM, BN, WM, WN, NSTAGE, true, MF, NL>((int)lds);
        hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, NSTAGE, true, MF, NL>), grid, dim3((WM * WN + NL) * 64), lds, s, d);
    } else {
        ensure_lds_limit<T, BM, BN, WM, WN, NSTAGE, false, MF, NL>((int)lds);
        hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, NSTAGE, false, MF, NL>), grid, dim3((WM * WN + NL) * 64), lds, s, d);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    if (d.splitk > 1) {
        const int64_t total = (int64_t)d.M * (d.ldp / 16);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (total * d.zbatch < REDUCE_PIPE_MAX_THREADS) hipLaunchKernelGGL((igemm_splitk_reduce<T, true>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((igemm_splitk_reduce<T, false>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        e = hipGetLastError();
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

#ifdef UR_WITH_WSCONV
int wsconv_launch(const ur_igemm_desc& d, hipStream_t s);  // wsconv.hip (make WSCONV=1)
#else
static int wsconv_launch(const ur_igemm_desc&, hipStream_t) { return UR_E_UNSUPPORTED; }  // not in the product build
#endif
#ifdef UR_WITH_PP
int igemm_pp_launch(const ur_igemm_desc& d, hipStream_t s);  // igemm_pp.hip (make PP=1)
#else
static int igemm_pp_launch(const ur_igemm_desc&, hipStream_t) { return UR_E_UNSUPPORTED; }  // not in the product build
#endif
// igemm_dxs.hip: the 3x3 conv with the three dx taps sharing one staged pixel block.  Parity-green and 4-8 % SLOWER in the step
// than the lock-step kernels, so an opt-in build like wsconv.hip: `make DXS=1`, then UR_DXS=1.
#ifdef UR_WITH_DXS
int igemm_dxs_launch(const ur_igemm_desc& d, hipStream_t s);
bool igemm_dxs_ok(const ur_igemm_desc& d, int bm);
int igemm_dxs_tile_bm(int tile);
static int dxs_enabled() {
    static const int v = [] { const char* e = std::getenv("UR_DXS"); return (e && e[0] == '1') ? 1 : 0; }();
    return v;
}
#else
static int igemm_dxs_launch(const ur_igemm_desc&, hipStream_t) { return UR_E_UNSUPPORTED; }
static bool igemm_dxs_ok(const ur_igemm_desc&, int) { return false; }
static int igemm_dxs_tile_bm(int) { return 0; }
static int dxs_enabled() { return 0; }
#endif

// ping-pong main pass + the shared split-K second pass
template <typename T>
static int launch_pp(const ur_igemm_desc& d, hipStream_t s, bool reduce) {
    const int rc = igemm_pp_launch(d, s);
    if (rc) return rc;
    if (d.splitk > 1 && reduce) {
        const int64_t total = (int64_t)d.M * (d.ldp / 16);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (total * d.zbatch < REDUCE_PIPE_MAX_THREADS) hipLaunchKernelGGL((igemm_splitk_reduce<T, true>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((igemm_splitk_reduce<T, false>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

// weight-streaming conv main pass + the shared split-K second pass
template <typename T>
static int launch_ws(const ur_igemm_desc& d, hipStream_t s, bool reduce) {
    const int rc = wsconv_launch(d, s);
    if (rc) return rc;
    if (d.splitk > 1 && reduce) {
        const int64_t total = (int64_t)d.M * (d.ldp / 16);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (total * d.zbatch < REDUCE_PIPE_MAX_THREADS) hipLaunchKernelGGL((igemm_splitk_reduce<T, true>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((igemm_splitk_reduce<T, false>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

template <typename T>
// reduce = false: main pass only (the caller runs its own second pass over the fp32 slabs: ur_igemm_splitk_gn)
static int launch_dtype(ur_igemm_desc& d, hipStream_t s, bool reduce) {
    if (d.taps == 9 && dxs_enabled()) {
        const int bm = igemm_dxs_tile_bm(d.tile);
        if (bm && igemm_dxs_ok(d, bm)) {
            const int rc = igemm_dxs_launch(d, s);
            if (rc) return rc;
            if (d.splitk > 1 && reduce) {
                const int64_t total = (int64_t)d.M * (d.ldp / 16);
                int blocks = (int)((total + 255) / 256);
                if (blocks > 4096) blocks = 4096;
                if (total * d.zbatch < REDUCE_PIPE_MAX_THREADS) hipLaunchKernelGGL((igemm_splitk_reduce<T, true>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((igemm_splitk_reduce<T, false>), dim3(blocks, d.zbatch), dim3(256), 0, s, d);
                const hipError_t e = hipGetLastError();
                if (e != hipSuccess) return -(int)e;
            }
            return 0;
        }
    }
    switch (d.tile) {
        case UR_TILE_128x128: return launch_cfg<T, 128, 128, 2, 2, 2>(d, s, reduce);
        case UR_TILE_128x64: return launch_cfg<T, 128, 64, 4, 1, 3>(d, s, reduce);
        case UR_TILE_64x64: return launch_cfg<T, 64, 64, 4, 1, 3>(d, s, reduce);
        case UR_TILE_128x128_S3: return launch_cfg<T, 128, 128, 2, 2, 3>(d, s, reduce);
        case UR_TILE_128x64_S2: return launch_cfg<T, 128, 64, 4, 1, 2>(d, s, reduce);
        case UR_TILE_64x64_S4: return launch_cfg<T, 64, 64, 4, 1, 4>(d, s, reduce);
        case UR_TILE_64x64_S2: return launch_cfg<T, 64, 64, 4, 1, 2>(d, s, reduce);
        case UR_TILE_256x128: return launch_cfg<T, 256, 128, 4, 2, 2>(d, s, reduce);
        case UR_TILE_128x320: return launch_cfg<T, 128, 320, 2, 5, 2>(d, s, reduce);
        case UR_TILE_128x256: return launch_cfg<T, 128, 256, 2, 4, 2>(d, s, reduce);
        case UR_TILE_256x256: return launch_cfg<T, 256, 256, 4, 4, 2>(d, s, reduce);
        case UR_TILE_64x64_R: return launch_cfg<T, 64, 64, 4, 1, -2>(d, s, reduce);
        case UR_TILE_128x64_R: return launch_cfg<T, 128, 64, 4, 1, -2>(d, s, reduce);
        case UR_TILE_128x128_R: return launch_cfg<T, 128, 128, 2, 2, -2>(d, s, reduce);
        case UR_TILE_128x320_R: return launch_cfg<T, 128, 320, 2, 5, -2>(d, s, reduce);
        case UR_TILE_256x128_R: return launch_cfg<T, 256, 128, 4, 2, -2>(d, s, reduce);
        case UR_TILE_64x64_W1: return launch_cfg<T, 64, 64, 1, 1, 2>(d, s, reduce);
        case UR_TILE_128x64_W2: return launch_cfg<T, 128, 64, 2, 1, 2>(d, s, reduce);
        case UR_TILE_64x64_W1_S3: return launch_cfg<T, 64, 64, 1, 1, 3>(d, s, reduce);
        case UR_TILE_64x128_W2: return launch_cfg<T, 64, 128, 1, 2, 2>(d, s, reduce);
        case UR_TILE_64x64_W1_S4: return launch_cfg<T, 64, 64, 1, 1, 4>(d, s, reduce);
        case UR_TILE_128x320_M32: return launch_cfg<T, 128, 320, 2, 5, 2, 32>(d, s, reduce);
        case UR_TILE_128x128_M32: return launch_cfg<T, 128, 128, 2, 2, 2, 32>(d, s, reduce);
        case UR_TILE_128x64_M32: return launch_cfg<T, 128, 64, 4, 1, 2, 32>(d, s, reduce);
        case UR_TILE_128x64_S3_M32: return launch_cfg<T, 128, 64, 4, 1, 3, 32>(d, s, reduce);
        case UR_TILE_64x64_M32: return launch_cfg<T, 64, 64, 2, 2, 2, 32>(d, s, reduce);
        case UR_TILE_64x64_S3_M32: return launch_cfg<T, 64, 64, 2, 2, 3, 32>(d, s, reduce);
        case UR_TILE_256x256_M32: return launch_cfg<T, 256, 256, 4, 4, 2, 32>(d, s, reduce);
        case UR_TILE_256x128_M32: return launch_cfg<T, 256, 128, 4, 2, 2, 32>(d, s, reduce);
        case UR_TILE_128x256_M32: return launch_cfg<T, 128, 256, 2, 4, 2, 32>(d, s, reduce);
        case UR_TILE_128x320_L2: return launch_cfg<T, 128, 320, 2, 5, 2, 16, 2>(d, s, reduce);
        case UR_TILE_128x320_L4: return launch_cfg<T, 128, 320, 2, 5, 2, 16, 4>(d, s, reduce);
        case UR_TILE_128x128_L2: return launch_cfg<T, 128, 128, 2, 2, 2, 16, 2>(d, s, reduce);
        case UR_TILE_128x128_S3_L2: return launch_cfg<T, 128, 128, 2, 2, 3, 16, 2>(d, s, reduce);
        case UR_TILE_128x64_L1: return launch_cfg<T, 128, 64, 4, 1, 2, 16, 1>(d, s, reduce);
        case UR_TILE_128x64_S3_L2: return launch_cfg<T, 128, 64, 4, 1, 3, 16, 2>(d, s, reduce);
        case UR_TILE_64x64_S3_L1: return launch_cfg<T, 64, 64, 4, 1, 3, 16, 1>(d, s, reduce);
        case UR_TILE_256x128_L2: return launch_cfg<T, 256, 128, 4, 2, 2, 16, 2>(d, s, reduce);
        case UR_TILE_256x256_L0: return UR_E_UNSUPPORTED;  /* 16 consumer waves already fill the 1024-thread limit */
        case UR_TILE_128x256_L2: return launch_cfg<T, 128, 256, 2, 4, 2, 16, 2>(d, s, reduce);
        case UR_TILE_128x256_S3: return launch_cfg<T, 128, 256, 2, 4, 3>(d, s, reduce);
        case UR_TILE_128x320_W8_M32: return launch_cfg<T, 128, 320, 4, 2, 2, 32>(d, s, reduce);
        case UR_TILE_256x320_W16_M32: return launch_cfg<T, 256, 320, 8, 2, 2, 32>(d, s, reduce);
        case UR_TILE_128x160_M32: return launch_cfg<T, 128, 160, 4, 1, 2, 32>(d, s, reduce);
        case UR_TILE_128x160_S3_M32: return launch_cfg<T, 128, 160, 4, 1, 3, 32>(d, s, reduce);
        case UR_TILE_64x320_M32: return launch_cfg<T, 64, 320, 2, 2, 2, 32>(d, s, reduce);
        case UR_TILE_256x160_W4_M32: return launch_cfg<T, 256, 160, 4, 1, 2, 32>(d, s, reduce);
        case UR_TILE_256x320_W8_M32: return launch_cfg<T, 256, 320, 4, 2, 2, 32>(d, s, reduce);
        case UR_TILE_128x320_W4_M32: return launch_cfg<T, 128, 320, 2, 2, 2, 32>(d, s, reduce);
        case UR_TILE_256x128_W4_M32: return launch_cfg<T, 256, 128, 4, 1, 2, 32>(d, s, reduce);
        case UR_TILE_256x256_W8_M32: return launch_cfg<T, 256, 256, 4, 2, 2, 32>(d, s, reduce);
        case UR_TILE_256x320_W10: return launch_cfg<T, 256, 320, 2, 5, 2>(d, s, reduce);
        case UR_TILE_WS320: return launch_ws<T>(d, s, reduce);
        case UR_TILE_WS320_W8: return launch_ws<T>(d, s, reduce);
        case UR_TILE_PP_128x320: case UR_TILE_PP_128x320_S4: case UR_TILE_PP_256x128: case UR_TILE_PP_128x256:
        case UR_TILE_PP_256x256: case UR_TILE_PP_128x128: case UR_TILE_PP_256x320: return launch_pp<T>(d, s, reduce);
    }
    return UR_E_BADARG;
}

static int64_t padded_ldp(const ur_igemm_desc& d, int tile) {
    const int bn = kTiles[tile].bn;
    return (int64_t)((d.N + bn - 1) / bn) * bn;
}

}  // namespace ur

extern "C" int ur_has_pp(void) {
#ifdef UR_WITH_PP
    return 1;
#else
    return 0;
#endif
}

extern "C" int ur_has_wsconv(void) {
#ifdef UR_WITH_WSCONV
    return 1;
#else
    return 0;
#endif
}

extern "C" int64_t ur_igemm_partial_floats(const ur_igemm_desc* d) {
    if (!d || d->splitk <= 1) return 0;
    // worst case over the tile configurations: N padded to a multiple of the tile width
    int64_t ldp = 0;
    for (int t = 1; t < UR_TILE_COUNT; ++t) {
        const int64_t bn = ur::kTiles[t].bn;
        const int64_t v = (d->N + bn - 1) / bn * bn;
        if (v > ldp) ldp = v;
    }
    return (int64_t)d->splitk * (d->zbatch > 1 ? d->zbatch : 1) * d->M * ldp;
}

// validates and completes the descriptor, then launches (main pass + the standard split-K second pass if `reduce`)
static int igemm_run(ur_igemm_desc& d, void* stream, bool reduce = true) {
    using namespace ur;
    if (!d.x0 || !d.w || !d.out || !d.zero_page) return UR_E_BADARG;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return UR_E_BADARG;
    if (d.taps != 1 && d.taps != 9) return UR_E_BADARG;
    if ((d.c0 % BK) || (d.c1 % BK) || (d.c1 > 0 && !d.x1)) return UR_E_BADARG;
    if (d.ct0 < 0 || d.ct1 < 0 || (d.ct0 == 0 && d.ct1 != 0)) return UR_E_BADARG;
    if (d.ct0 > 0 && (d.taps != 9 || d.stride != 1 || d.ups || !d.t0 || (d.ct1 > 0 && !d.t1) || (d.ct0 % BK) || (d.ct1 % BK) ||
                      (d.ldt0 % 8) || (d.ct1 > 0 && (d.ldt1 % 8))))
        return UR_E_BADARG;
    if (d.K != d.taps * (d.c0 + d.c1) + d.ct0 + d.ct1) return UR_E_BADARG;
    if ((d.ldx0 % 8) || (d.c1 > 0 && (d.ldx1 % 8)) || (d.ldw % 8)) return UR_E_BADARG;
    if (d.taps == 9) {
        if (d.act == UR_ACT_GEGLU || d.out_vt) return UR_E_BADARG;  // the conv instantiations carry the lean epilogue (igemm_epi.h)
        if (d.B <= 0 || d.Hin <= 0 || d.Win <= 0 || d.Hout <= 0 || d.Wout <= 0) return UR_E_BADARG;
        if (d.stride != 1 && d.stride != 2) return UR_E_BADARG;
        if (d.M != d.B * d.Hout * d.Wout) return UR_E_BADARG;
        if (d.ups && d.stride != 1) return UR_E_BADARG;
        if (d.pad != 0 && d.pad != 1) return UR_E_BADARG;
    }
    if (d.cblock < 0 || (d.cblock > 0 && (d.taps != 9 || (d.cblock % BK) || d.c1 != 0 || (d.c0 % d.cblock)))) return UR_E_BADARG;
    if (d.zbatch < 1) d.zbatch = 1;
    if (d.zx_div < 1) d.zx_div = 1;
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > d.K / BK) d.splitk = d.K / BK;
    if (d.splitk > 1 && !d.partial) return UR_E_BADARG;
    if (d.rowadd && d.rows_per_b <= 0) return UR_E_BADARG;
    if (d.act == UR_ACT_GEGLU && (d.N % 16)) return UR_E_BADARG;
    if (d.out_vt) {  // transposed side output for the columns n >= vt_n0
        if (d.vt_n0 <= 0 || d.vt_n0 >= d.N || (d.vt_n0 % 16) || ((d.N - d.vt_n0) % 16) || d.vt_rows <= 0 || d.ldvt <= 0) return UR_E_BADARG;
        if (d.act != UR_ACT_NONE || d.res || d.rowadd) return UR_E_BADARG;
        if (d.n_store <= 0) d.n_store = d.vt_n0;
        if (d.n_store > d.vt_n0) return UR_E_BADARG;
    }
    if (d.n_store <= 0) d.n_store = (d.act == UR_ACT_GEGLU) ? d.N / 2 : d.N;
    if (d.zero_page_bytes != 0 && (d.zero_page_bytes < 256 || (d.zero_page_bytes & (d.zero_page_bytes - 1)))) return UR_E_BADARG;
    if (d.tile == UR_TILE_AUTO) d.tile = pick_tile(d);
    if (d.tile < 1 || d.tile >= UR_TILE_COUNT) return UR_E_BADARG;
    d.ldp = padded_ldp(d, d.tile);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.dtype == UR_DT_F16) return launch_dtype<f16>(d, s, reduce);
    if (d.dtype == UR_DT_BF16) return launch_dtype<bf16>(d, s, reduce);
    return UR_E_BADARG;
}

// 1 when ur_igemm would run this conv on the dx-tap-sharing kernel (igemm_dxs.hip), 0 otherwise -- for tests and tools
extern "C" int ur_igemm_uses_dxs(const ur_igemm_desc* d) {
    using namespace ur;
    if (!d || d->taps != 9 || !dxs_enabled()) return 0;
    const int bm = igemm_dxs_tile_bm(d->tile);
    return bm && igemm_dxs_ok(*d, bm) ? 1 : 0;
}

extern "C" int ur_igemm(const ur_igemm_desc* din, void* stream) {
    if (!din) return UR_E_BADARG;
    ur_igemm_desc d = *din;
    return igemm_run(d, stream);
}

extern "C" int ur_igemm_splitk_gn(const ur_igemm_desc* din, const float* gamma, const float* beta, int64_t zgn, float eps,
                                  int groups, int silu, void* stream) {
    using namespace ur;
    if (!din || !gamma || !beta || groups <= 0) return UR_E_BADARG;
    ur_igemm_desc d = *din;
    if (d.taps != 9 || d.splitk <= 1 || d.res || d.out_lo || d.out_vt || d.act != UR_ACT_NONE) return UR_E_BADARG;
    const int rows = d.Hout * d.Wout;
    if (d.N % groups || ((d.N / groups) & 3) || (d.ldc & 3) || (int64_t)rows * (d.N / groups) > (int64_t)RGN_THREADS * RGN_MAXQ * 4)
        return UR_E_UNSUPPORTED;
    if (d.rowadd && d.rows_per_b != rows) return UR_E_BADARG;
    if (d.n_store > 0 && d.n_store != d.N) return UR_E_BADARG;
    const int rc = igemm_run(d, stream, /*reduce=*/false);
    if (rc) return rc;
    if (d.splitk <= 1) return UR_E_BADARG;  // (clamped away: K too short for a split)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(groups, d.B, d.zbatch);
    if (d.dtype == UR_DT_F16)
        hipLaunchKernelGGL((igemm_splitk_reduce_gn<f16>), grid, dim3(RGN_THREADS), 0, s, d, gamma, beta, zgn, eps, groups, silu, rows);
    else
        hipLaunchKernelGGL((igemm_splitk_reduce_gn<bf16>), grid, dim3(RGN_THREADS), 0, s, d, gamma, beta, zgn, eps, groups, silu, rows);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
