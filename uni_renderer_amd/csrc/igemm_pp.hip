// 8-wave PING-PONG implicit GEMM for gfx950 (UR_TILE_PP_*): the same problem, descriptor, operand layouts and epilogue as
// igemm.hip (conv3x3 s1 / s2 / nearest-2x / 2-source concat / 1x1 tail, 1x1 conv / Linear, split-K), a different main loop.
//
// Why.  The lock-step tiles of igemm.hip run every wave of a workgroup through the same phase at the same time: all
// issue their LDS-DMA pieces (matrix pipes idle), all read fragments, all multiply, all wait at the chunk barrier.  Counters
// of the dominant conv (profiles/r03_pmc_conv_sq.json): 44 % of wave-cycles parked, MFMA-busy 33 %.  Per 64-deep K chunk a
// 128x320 tile needs 1280 matrix-pipe cycles per SIMD AND 56 one-KiB LDS-DMA pieces through the CU's 64 B/clk
// texture-address path (~900 cycles): in lock step the two ADD, here they OVERLAP.
//
// How (cdna_hip_programming.md section 5, "8-phase" template; MI355X_MICROARCH.md "Two waves per SIMD").  512 threads =
// 8 waves = 2 per SIMD, in two groups of four (waves 0-3 / 4-7: one wave of each group on every SIMD).  Every wave runs
//
//     READ(s): ds_read the fragments of stage s, issue its LDS-DMA pieces of stage s + NS - 2   | s_barrier |
//     MFMA(s): the stage's MFMAs at raised priority                                              | s_barrier |
//
// and group 1 executes ONE extra s_barrier first, so the groups are half a phase apart: between two barriers one group
// multiplies while the other reads and copies.  A stage is 32 K elements (64-byte LDS rows) so that an NS-deep ring (4 or
// 5 slots) fits beside nothing else: prefetch distance NS - 2 stages, counted vmcnt (never 0 in the steady state), raw
// s_barrier.  Hazards, by global barrier number b (group 0: READ(s) in (2s, 2s+1), MFMA(s) in (2s+1, 2s+2); group 1 one
// later):
//   RAW  a wave waits for ITS pieces of stage s+1 before barrier 2(s+1); every READ(s+1) is after that barrier.
//   WAR  stage s+NS-2 lands in the slot of stage s-2; its last reader (group 1, READ(s-2) in (2s-3, 2s-2), data back
//        before its MFMAs in (2s-2, 2s-1)) is two barriers behind the earliest issue (group 0, after barrier 2s).
//
// LDS image: row = one pixel / one output channel, 64 bytes = 4 chunks of 16; chunk c of row r sits at position
// c ^ ((r >> 2) & 3) (applied on the SOURCE address, the LDS-DMA destination is lane-linear): the 32x32x16 fragment reads
// (32 consecutive rows per half-wave) are bank-conflict free (tools/lds_bank_check.py model).  v_mfma_f32_32x32x16 with the
// weight tile as operand A; weight rows permuted inside 32-blocks so that a lane ends with 16 consecutive channels of one
// pixel -- igemm.hip's MF = 32 layout, epilogue shared (igemm_epi.h).
#include "igemm_epi.h"

namespace ur {

constexpr int BKS = 32;  // K elements per stage (64 bytes per row)

// -DUR_PP_ABLATE=<bits> builds measurement-only variants (tools/experiments/r04_run3.sh; never in the product build):
//   1 = no LDS-DMA copies (pointer bookkeeping kept), 2 = no MFMAs, 4 = no fragment reads, 8 = no barriers / waits
#ifndef UR_PP_ABLATE
#define UR_PP_ABLATE 0
#endif
// Schedule variants (compile-time, measured against each other in tools/experiments/r04_run5.sh):
//   bit 0: no s_setprio around the MFMA block
//   bit 1: the LDS-DMA pieces are issued from INSIDE the MFMA block (one piece after every few MFMAs, in the shadow of
//          the matrix pipe) instead of in the READ block, prefetch distance NS - 1 stages
#ifndef UR_PP_VARIANT
#define UR_PP_VARIANT 1
#endif

template <int N>
__device__ __forceinline__ void pp_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int NS, bool CONV>
__global__ void __launch_bounds__(512) igemm_pp_kernel(const ur_igemm_desc p) {
    typedef typename Vec8<T>::type vec8;
    static_assert(WM * WN == 8, "two groups of four waves");
    static_assert((BM / WM) % 32 == 0 && (BN / WN) % 32 == 0, "wave tile in 32x32 MFMA blocks");
    static_assert(NS == 4 || NS == 5, "prefetch distance NS - 2 = 2 or 3 stages");
    static_assert(BM % 128 == 0, "every wave copies the same number of pixel pieces");
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int XP = BM / 16, WP = BN / 16;          // LDS-DMA pieces (16 rows x 64 B) per stage
    constexpr int XI = (XP + 7) / 8, WI = (WP + 7) / 8;
    constexpr int XT_BYTES = BM * 64, WT_BYTES = BN * 64, STAGE = XT_BYTES + WT_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD (cyclic placement): one of each group per SIMD
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
    const int zidx = lid / gridDim.x;
    const int tid_xy = lid - zidx * gridDim.x;
    const int tile_n = tid_xy % tiles_n;
    const int tile_m = tid_xy / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kt_total = p.K / BK;  // split-K slices are counted in 64-element chunks, as in igemm.hip (same slabs)
    int kbeg = 0, kend = kt_total, zb = zidx;
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
    const char* t0 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t0) + (int64_t)zb * p.zt0);
    const char* t1 = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.t1) + (int64_t)zb * p.zt1);
    // lane l of a piece copies LDS position (row 16 * piece + l / 4, chunk l & 3) <- source chunk (l & 3) ^ key(row),
    // key(row) = (row >> 2) & 3 = (l >> 4) & 3 (piece bases are multiples of 16 rows)
    const int jsrc = (lane & 3) ^ ((lane >> 4) & 3);
    // padding rows: this (workgroup, wave)'s own 128-byte line of the zero region (one hot line would be served to all CUs
    // by one L2 channel)
    const unsigned zbytes = p.zero_page_bytes >= 256 ? (unsigned)p.zero_page_bytes : 256u;
    const char* zp = reinterpret_cast<const char*>(p.zero_page) + ((((unsigned)lid * 16u + (unsigned)wave) * 128u) & (zbytes - 128u)) + (lane & 3) * 16;

    // ---- per-lane row bookkeeping (fixed over the K loop): igemm.hip's, with 16-row pieces ----
    int xa[XI], xy[XI], xx[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int r = (it * 8 + wave) * 16 + (lane >> 2);
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
        const int r = (it * 8 + wave) * 16 + (lane >> 2);  // LDS row of the weight tile
        const int rho = r & 31;
        // LDS row 8g + 4h + rr of a 32-block holds semantic column 16h + 4g + rr (a lane's 16 accumulator registers of a
        // 32x32 block are then 16 consecutive output channels)
        const int sem = (r & ~31) | (((rho >> 2) & 1) << 4) | ((rho >> 3) << 2) | (rho & 3);
        const int n = n0 + sem;
        const bool ok = n < p.N;
        const int64_t off = ((int64_t)n * p.ldw + (int64_t)kbeg * BK + jsrc * 8) * (int64_t)sizeof(T);
        wptr[it] = ok ? wp + off : zp;
        winc[it] = ok ? 64 : 0;
    }

    // segment state of the loader (wave-uniform), in stages of 32 channels; see igemm.hip for the K orders
    const int pc0 = p.c0, pc1 = p.c1, pct0 = p.ct0, pct1 = p.ct1;
    const int64_t pldx0 = p.ldx0, pldx1 = p.ldx1, pldt0 = p.ldt0, pldt1 = p.ldt1;
    const int pHin = p.Hin << p.ups, pWin = p.Win << p.ups, pups = p.ups, pW = p.Win;
    const int segs_per_tap = (pc1 > 0) ? 2 : 1;
    const int cblk = p.cblock;
    int seg_tap, seg_src, seg_left, seg_coff = 0;
    auto uniform_i64 = [](int64_t v) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto uniform_ptr = [&](const char* q) __attribute__((always_inline)) {
        return reinterpret_cast<const char*>(uniform_i64(reinterpret_cast<int64_t>(q)));
    };
    const char* seg_base = x0;
    int64_t seg_ld = pldx0;
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
            const int64_t off = ((int64_t)pix * ld + seg_coff + cc + jsrc * 8) * (int64_t)sizeof(T);
            xptr[it] = ok ? sb + off : zp;
            xinc[it] = ok ? 64 : 0;
        }
    };
    {
        const int Cin = pc0 + pc1;
        const int kglob = kbeg * BK;
        int cc;
        if (CONV && kglob >= 9 * Cin) {  // a split-K slice that starts inside the 1x1 tail
            cc = kglob - 9 * Cin;
            seg_src = 2; seg_base = t0; seg_ld = pldt0;
            seg_left = (pct0 - cc) / BKS;
            if (cc >= pct0) { seg_src = 3; cc -= pct0; seg_base = t1; seg_ld = pldt1; seg_left = (pct1 - cc) / BKS; }
            seg_tap = 4;
        } else if (cblk > 0) {
            const int blk = kglob / (9 * cblk);
            const int rem = kglob - blk * 9 * cblk;
            seg_tap = rem / cblk;
            cc = rem - seg_tap * cblk;
            seg_src = 0;
            seg_coff = blk * cblk;
            seg_left = (cblk - cc) / BKS;
        } else {
            seg_tap = kglob / Cin;
            cc = kglob - seg_tap * Cin;
            seg_src = (cc >= pc0) ? 1 : 0;
            if (seg_src) { cc -= pc0; seg_base = x1; seg_ld = pldx1; }
            seg_left = ((seg_src ? pc1 : pc0) - cc) / BKS;
        }
        set_pointers(cc);
    }
    auto next_segment = [&]() __attribute__((always_inline)) {
        bool to_tail = false;
        if (seg_src >= 2) {
            seg_src = 3;
            seg_base = t1; seg_ld = pldt1; seg_left = pct1 / BKS;
        } else if (cblk > 0) {
            seg_tap += 1;
            if (seg_tap == 9) { seg_tap = 0; seg_coff += cblk; }
            seg_left = cblk / BKS;
            to_tail = CONV && seg_coff >= pc0;
        } else {
            seg_src += 1;
            if (seg_src >= segs_per_tap) { seg_src = 0; seg_tap += 1; }
            seg_base = seg_src ? x1 : x0;
            seg_ld = seg_src ? pldx1 : pldx0;
            seg_left = (seg_src ? pc1 : pc0) / BKS;
            to_tail = CONV && seg_tap == 9;
        }
        if (to_tail) {
            seg_src = 2; seg_base = t0; seg_ld = pldt0; seg_left = pct0 / BKS;
            seg_tap = 4; seg_coff = 0;
        }
        seg_tap = __builtin_amdgcn_readfirstlane(seg_tap);
        seg_src = __builtin_amdgcn_readfirstlane(seg_src);
        seg_left = __builtin_amdgcn_readfirstlane(seg_left);
        seg_coff = __builtin_amdgcn_readfirstlane(seg_coff);
        seg_base = uniform_ptr(seg_base);
        seg_ld = uniform_i64(seg_ld);
        set_pointers(0);
    };
    // piece j (0 .. XI-1: pixel pieces, XI .. XI+WI-1: weight pieces) of the loader's current stage into ring slot `slot`
    auto issue_piece = [&](int j, int slot) __attribute__((always_inline)) {
        char* xs = smem + slot * STAGE;
        char* ws = xs + XT_BYTES;
        if (UR_PP_ABLATE & 1) return;
        if (j < XI) {
            glds16(xptr[j], xs + (j * 8 + wave) * 1024);  // XP is a multiple of 8
        } else {
            const int it = j - XI;
            if ((it + 1) * 8 <= WP || it * 8 + wave < WP) glds16(wptr[it], ws + (it * 8 + wave) * 1024);
        }
    };
    // advance the loader by one stage (pointer steps; a new (tap, source) segment recomputes the pixel pointers)
    auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XI; ++it) xptr[it] += xinc[it];
#pragma unroll
        for (int it = 0; it < WI; ++it) wptr[it] += winc[it];
        seg_left -= 1;
        if (seg_left == 0) next_segment();
    };
    // issue this wave's LDS-DMA pieces of the loader's current stage into ring slot `slot`, advance by one stage
    auto stage = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < XI + WI; ++j) issue_piece(j, slot);
        advance();
    };
    // LDS-DMA instructions a wave issues per stage: L0, or L0 + 1 for the first WP % 8 waves -- the unit of the counted
    // vmcnt waits.  wait_newer(k): at most k STAGES of this wave's copies still in flight (k <= NS - 2).
    constexpr int L0 = XP / 8 + WP / 8;
    const bool extra = wave < (WP & 7);
    auto wait_newer = [&](int k) __attribute__((always_inline)) {
        if (UR_PP_ABLATE & 8) return;
        if (NS == 5 && k >= 3) {
            if (extra) pp_vmcnt<3 * (L0 + 1)>(); else pp_vmcnt<3 * L0>();
        } else if (k >= 2) {
            if (extra) pp_vmcnt<2 * (L0 + 1)>(); else pp_vmcnt<2 * L0>();
        } else if (k >= 1) {
            if (extra) pp_vmcnt<L0 + 1>(); else pp_vmcnt<L0>();
        } else {
            pp_vmcnt<0>();
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // fragment addresses inside a slot: lane (row l31 of a 32-block, half hh) reads, at k16 step s, source chunk 2 s + hh of
    // its row = LDS position (2 s + hh) ^ ((row >> 2) & 3) (block bases are multiples of 32)
    const int l31 = lane & 31, hh = lane >> 5;
    const int key = (l31 >> 2) & 3;
    const int c0 = ((0 + hh) ^ key) << 4, c1 = ((2 + hh) ^ key) << 4;
    const int xrow = (wm * (32 * MI) + l31) * 64;
    const int wrow = XT_BYTES + (wn * (32 * NI) + l31) * 64;

    vec8 wf0[NI] = {}, wf1[NI] = {}, xf0[MI] = {}, xf1[MI] = {};
    auto read_frags = [&](int slot) __attribute__((always_inline)) {
        if (UR_PP_ABLATE & 4) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { asm volatile("" : "+v"(wf0[ni])); asm volatile("" : "+v"(wf1[ni])); }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { asm volatile("" : "+v"(xf0[mi])); asm volatile("" : "+v"(xf1[mi])); }
            return;
        }
        const char* base = smem + slot * STAGE;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf0[ni] = *reinterpret_cast<const vec8*>(base + wrow + ni * 2048 + c0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf0[mi] = *reinterpret_cast<const vec8*>(base + xrow + mi * 2048 + c0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf1[ni] = *reinterpret_cast<const vec8*>(base + wrow + ni * 2048 + c1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf1[mi] = *reinterpret_cast<const vec8*>(base + xrow + mi * 2048 + c1);
    };
    auto multiply = [&](int islot_issue) __attribute__((always_inline)) {
        if (UR_PP_ABLATE & 2) {  // keep the fragments live (the reads must not be dead code)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { asm volatile("" ::"v"(wf0[ni])); asm volatile("" ::"v"(wf1[ni])); }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { asm volatile("" ::"v"(xf0[mi])); asm volatile("" ::"v"(xf1[mi])); }
            return;
        }
        constexpr int TM = 2 * MI * NI, NP = XI + WI;
        int cnt = 0, pj = 0;  // compile-time after unrolling
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = h ? mfma32(wf1[ni], xf1[mi], acc[mi][ni]) : mfma32(wf0[ni], xf0[mi], acc[mi][ni]);
                    ++cnt;
                    if ((UR_PP_VARIANT & 2) && pj < NP && cnt == (pj + 1) * TM / (NP + 1)) {
                        // one copy in the shadow of the MFMAs just issued (pinned: the scheduler would cluster them)
                        __builtin_amdgcn_sched_barrier(0);
                        if (islot_issue >= 0) issue_piece(pj, islot_issue);
                        __builtin_amdgcn_sched_barrier(0);
                        ++pj;
                    }
                }
    };

    auto barrier = [&]() __attribute__((always_inline)) {
        if (!(UR_PP_ABLATE & 8)) __builtin_amdgcn_s_barrier();
    };
    const int nst = 2 * (kend - kbeg);  // stages of this workgroup
    if (nst > 0) {
        constexpr int D = (UR_PP_VARIANT & 2) ? NS - 1 : NS - 2;  // prefetch distance in stages
        // prologue: stages 0 .. D-1 in flight, stage 0 landed and published
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nst) stage(s);
        wait_newer(min(D - 1, nst - 1));
        barrier();                 // global barrier 0
        if (grp == 1) barrier();   // group 1 runs one barrier behind
        asm volatile("" ::: "memory");
        int slot = 0, islot = D % NS;
        for (int s = 0; s < nst; ++s) {
            const bool more = s + D < nst;
            // ---- READ(s): fragments of stage s (variant 0: then this wave's copies of stage s + D)
            read_frags(slot);
            __builtin_amdgcn_sched_barrier(0);
            if (!(UR_PP_VARIANT & 2) && more) stage(islot);
            // stage s+1 must have landed (for this wave's pieces) before the EVEN global barrier 2(s+1): for group 1 that
            // is the barrier ending its READ, for group 0 the one ending its MFMA block.  Newer stages stay in flight:
            // issued so far = up to stage s + D (variant 2, group 1 at this point: s + D - 1).
            const int newer0 = min(D - 1, nst - 2 - s);
            const int newer1 = (UR_PP_VARIANT & 2) ? min(D - 2, nst - 2 - s) : newer0;
            if (grp == 1 && s + 1 < nst) wait_newer(newer1);
            barrier();
            asm volatile("" ::: "memory");
            // ---- MFMA(s) (variant 2: with the copies of stage s + D between the MFMAs)
            if (!(UR_PP_VARIANT & 1)) __builtin_amdgcn_s_setprio(1);
            multiply(more ? islot : -1);
            if (!(UR_PP_VARIANT & 1)) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if ((UR_PP_VARIANT & 2) && more) advance();
            if (grp == 0 && s + 1 < nst) wait_newer(newer0);
            barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1 == NS) ? 0 : slot + 1;
            islot = (islot + 1 == NS) ? 0 : islot + 1;
        }
        if (grp == 0) barrier();   // same barrier count for both groups
    }

    // ---- epilogue (igemm.hip's MF = 32 form): lane (j = lane & 31, h = lane >> 5) holds pixel row j of a 32-row block and
    // channels h*16 .. h*16+15 of a 32-block
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
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int m = m0 + wm * (32 * MI) + mi * 32 + l31;
            const int nc = n0 + wn * (32 * NI) + ni * 32 + hh * 16;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[mi][ni][r];
            finish(m, nc, v);
        }
}

template <typename T, int BM, int BN, int WM, int WN, int NS>
static int pp_launch_cfg(const ur_igemm_desc& d, hipStream_t s) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, d.zbatch * d.splitk);
    constexpr int lds = NS * (BM + BN) * 64;
    static_assert(lds <= 160 * 1024, "ring does not fit the CU's LDS");
    if (d.taps == 9) {
        static std::atomic<uint64_t> done{0};
        set_lds_limit_once(done, reinterpret_cast<const void*>(&igemm_pp_kernel<T, BM, BN, WM, WN, NS, true>), lds);
        hipLaunchKernelGGL((igemm_pp_kernel<T, BM, BN, WM, WN, NS, true>), grid, dim3(512), lds, s, d);
    } else {
        static std::atomic<uint64_t> done{0};
        set_lds_limit_once(done, reinterpret_cast<const void*>(&igemm_pp_kernel<T, BM, BN, WM, WN, NS, false>), lds);
        hipLaunchKernelGGL((igemm_pp_kernel<T, BM, BN, WM, WN, NS, false>), grid, dim3(512), lds, s, d);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
static int pp_launch_dtype(const ur_igemm_desc& d, hipStream_t s) {
    switch (d.tile) {
        case UR_TILE_PP_128x320: return pp_launch_cfg<T, 128, 320, 4, 2, 5>(d, s);
        case UR_TILE_PP_128x320_S4: return pp_launch_cfg<T, 128, 320, 4, 2, 4>(d, s);
        case UR_TILE_PP_256x128: return pp_launch_cfg<T, 256, 128, 4, 2, 5>(d, s);
        case UR_TILE_PP_128x256: return pp_launch_cfg<T, 128, 256, 2, 4, 5>(d, s);
        case UR_TILE_PP_256x256: return pp_launch_cfg<T, 256, 256, 4, 2, 4>(d, s);
        case UR_TILE_PP_128x128: return pp_launch_cfg<T, 128, 128, 4, 2, 5>(d, s);
        case UR_TILE_PP_256x320: return pp_launch_cfg<T, 256, 320, 4, 2, 4>(d, s);
    }
    return UR_E_BADARG;
}

// main pass of a ping-pong tile (the caller, igemm.hip, runs the shared split-K second pass)
int igemm_pp_launch(const ur_igemm_desc& d, hipStream_t s) {
    if (d.dtype == UR_DT_F16) return pp_launch_dtype<f16>(d, s);
    if (d.dtype == UR_DT_BF16) return pp_launch_dtype<bf16>(d, s);
    return UR_E_BADARG;
}

}  // namespace ur
