// Weight-streaming 3x3 conv (+ 1x1 tail) for NHWC maps whose channel counts are multiples of 320 -- every resnet conv
// of the SD-1.x UNets at the 64x64 / 32x32 / 16x16 levels (reference: diffusers 0.24 ResnetBlock2D.conv1 / conv2 /
// conv_shortcut as instantiated by models/unet_2d_blocks.py).  Same contract as ur_igemm with
// taps == 9 (it is dispatched from there as tile UR_TILE_WS320): out = conv3x3(x0) [+ W_tail . (t0 | t1)] + bias +
// rowadd [+ res], or fp32 split-K slabs for igemm_splitk_reduce.
//
// Why a second conv kernel: the LDS-tiled implicit GEMM (igemm.hip) keeps BOTH operands in LDS and reads each of them
// (M + N) / (M N) times per MFMA; at 128 x 320 tiles its matrix pipe is busy a third of the time (profiles/r03_*: 44 %
// of the wave-cycles parked at the barrier / waitcnt).  Here the organisation of tchain.hip is used instead:
//
//   * a workgroup = 4 waves = one per SIMD computes 128 pixels x 320 output channels; a wave owns 32 pixels and ALL 320
//     channels (accumulator 160 AGPRs), so the activations are read once per wave: 80 VGPRs hold the wave's 32 x 320
//     operand block of the current (channel block, tap) and feed 200 MFMAs (5 weight stages);
//   * only the WEIGHTS stream through the shared LDS ring (40-KiB stage images, pre-swizzled by the host:
//     tchain.py / wsconv_images), 2 slots, prefetch distance 1, one s_barrier per 40 MFMAs per wave;
//   * the next operand block is copied global -> LDS by LDS-DMA into a wave-private 20-KiB region while the current one
//     is multiplied (5 pieces per stage, inside the hand-scheduled MFMA stream, tchain_asm.inc), and moved LDS ->
//     registers at the block boundary.  Zero padding costs nothing: the copy is a `buffer_load ... lds` through a
//     descriptor of the source tensor, and a lane whose tap falls outside the image sets bit 31 of its offset -- out of
//     range, the hardware delivers zeros;
//   * K order: (channel block of 320, tap) outer, then the 320 channels -- the `cblock = 320` order of ur_igemm_desc,
//     followed by the tail channels in blocks of 320.
//
// LDS: 2 x 40960 (weights) + 4 x 20480 (operand staging) = the whole 160 KiB; constants are read from global memory in
// the epilogue.
#include "ur_common.h"
#include <type_traits>
#include "../../include/ur_kernels.h"
#include "tchain_asm.inc"

namespace ur {

#ifndef WS_SKEW
#define WS_SKEW 0
#endif
constexpr int WS_C = 320;
constexpr int WS_NT = 10;                  // 32-row output-channel tiles
constexpr int WS_STAGE = 40960;            // one weight stage image: 320 rows x 64 k
constexpr int WS_PIECES = 10;              // weight LDS-DMA pieces per wave per stage
constexpr int WS_STG = 20480;              // operand staging bytes per wave: 32 rows x 640 B
constexpr int WS_OPIECES = 20;             // operand LDS-DMA pieces per wave per block
constexpr int WS_LDS = 2 * WS_STAGE + 4 * WS_STG;
constexpr int WS_IO_ROW = WS_C * 2 + 16;   // epilogue staging: padded row (bytes)
constexpr int WS_IO_WAVE = 32 * WS_IO_ROW; // 20992 bytes per wave
static_assert(WS_LDS == 160 * 1024, "LDS budget");
static_assert(4 * WS_IO_WAVE <= WS_LDS, "epilogue staging");

typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;

template <typename T>
__device__ __forceinline__ void wsconv_body(const ur_igemm_desc& p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int tiles = p.M >> 7, ntn = p.N / WS_C;
    const int sk = p.splitk > 1 ? p.splitk : 1;
    // consecutive workgroups of an XCD share (z, split, n tile): one weight stream per L2
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % tiles; lid /= tiles;
    const int nt = lid % ntn; lid /= ntn;
    const int zidx = lid, zb = zidx / sk, ks = zidx - zb * sk;
    const int m0w = tile * 128 + wave * 32;   // first pixel (row of the GEMM) of this wave
    const int HW = p.Hout * p.Wout;

    // ---- K blocks of this split-K slice ----
    const int nb0 = p.c0 / WS_C, nbt0 = p.ct0 / WS_C, nbt1 = p.ct1 / WS_C;
    const int nblk = 9 * nb0 + nbt0 + nbt1;
    const int per = (nblk + sk - 1) / sk;
    const int kb = ks * per, ke = min(nblk, kb + per);

    // ---- weight stream of (z, n tile): nblk * 5 stage images; reads past the end are out of range = zeros ----
    const ws_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w)) + ((int64_t)zb * p.zw) * (int64_t)sizeof(T) +
            (int64_t)nt * nblk * 5 * WS_STAGE, 0, nblk * 5 * WS_STAGE, 0x00020000);
    const int voff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned stg0 = lds0 + 2 * WS_STAGE + wave * WS_STG;   // this wave's operand staging (LDS byte address)
    char* const stg = smem + 2 * WS_STAGE + wave * WS_STG;

    // ---- operand sources: descriptor bases moved back by (W + 1) pixels so that every tap offset is >= 0 ----
    const int64_t zx = (int64_t)(p.zx_div > 1 ? zb / p.zx_div : zb) * p.zx;
    const int ldb0 = (int)p.ldx0 * (int)sizeof(T), ldbt0 = (int)p.ldt0 * (int)sizeof(T), ldbt1 = (int)p.ldt1 * (int)sizeof(T);
    const int back0 = (p.Wout + 1) * ldb0;
    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x0)) + zx * (int64_t)sizeof(T) - back0, 0, 0x80000000u, 0x00020000);
    const ws_rsrc_t rs_t0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.t0 ? p.t0 : p.x0)) + (int64_t)zb * p.zt0 * (int64_t)sizeof(T), 0, 0x80000000u, 0x00020000);
    const ws_rsrc_t rs_t1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.t1 ? p.t1 : p.x0)) + (int64_t)zb * p.zt1 * (int64_t)sizeof(T), 0, 0x80000000u, 0x00020000);

    // ---- per-lane geometry of the 20 operand pieces: piece k = 5 s + j covers 16-byte element e = 64 k + lane of the
    // wave's [32 rows][40 chunks] block: row (e / 40), chunk (e % 40).  e + 320 = 8 rows further, same chunk: only five
    // (row, chunk) pairs are kept, the others follow by + 8 s rows.  inval[k]: bit t set = tap t of that row falls outside
    // the image (bit 9: always set, used to copy zeros when there is no next block). ----
    int pm5[5], c5[5];
    unsigned inval[WS_OPIECES];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int e = 64 * j + lane, r = e / 40;
        pm5[j] = m0w + r;
        c5[j] = (e - r * 40) * 16;
    }
#pragma unroll
    for (int k = 0; k < WS_OPIECES; ++k) {
        const int pm = pm5[k % 5] + 8 * (k / 5);
        const int pix = pm % HW, y = pix / p.Wout, x = pix - y * p.Wout;
        // taps t = 3 dy + dx: the top row loses dy = 0, the bottom row dy = 2, the left column dx = 0, the right column dx = 2
        const unsigned mk = (1u << 9) | (y == 0 ? 0007u : 0u) | (y == p.Hout - 1 ? 0700u : 0u) | (x == 0 ? 0111u : 0u) |
                            (x == p.Wout - 1 ? 0444u : 0u);
        inval[k] = mk;
    }
    // wave-uniform parameters of K block bi: source (0 = x0, 1 = t0, 2 = t1), bytes per pixel, tap (9 = no such block: copy
    // zeros), descriptor offset of (tap, channel block).  (Plain scalars: a struct holding a buffer descriptor does not
    // instantiate on the host pass.)
    struct Blk { int src; int ldb; int tap; unsigned soff; };
    auto blk_of = [&](int bi) __attribute__((always_inline)) -> Blk {
        Blk b;
        if (bi >= ke) { b.src = 0; b.ldb = ldb0; b.tap = 9; b.soff = 0; return b; }
        if (bi < 9 * nb0) {
            const int cb = bi / 9, t = bi - 9 * cb;
            b.src = 0; b.ldb = ldb0; b.tap = t;
            b.soff = (unsigned)(back0 + ((t / 3 - 1) * p.Wout + (t % 3 - 1)) * ldb0 + cb * (WS_C * (int)sizeof(T)));
        } else if (bi < 9 * nb0 + nbt0) {
            b.src = 1; b.ldb = ldbt0; b.tap = 4; b.soff = (unsigned)((bi - 9 * nb0) * (WS_C * (int)sizeof(T)));
        } else {
            b.src = 2; b.ldb = ldbt1; b.tap = 4; b.soff = (unsigned)((bi - 9 * nb0 - nbt0) * (WS_C * (int)sizeof(T)));
        }
        return b;
    };
    auto rs_of = [&](const Blk& b) __attribute__((always_inline)) -> ws_rsrc_t { return b.src == 0 ? rs_x : (b.src == 1 ? rs_t0 : rs_t1); };
    auto piece_off = [&](const Blk& b, int k) __attribute__((always_inline)) -> unsigned {
        const unsigned off = (unsigned)((pm5[k % 5] + 8 * (k / 5)) * b.ldb + c5[k % 5]);
        return off | (((inval[k] >> b.tap) & 1u) << 31);
    };

    // ---- pipeline fill: weights of the first stage, operand block kb ----
    {
        const int sbase = kb * 5 * WS_STAGE + wave * (WS_PIECES * 1024);
#pragma unroll
        for (int i = 0; i < WS_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(smem + ((kb * 5) & 1) * WS_STAGE + wave * (WS_PIECES * 1024) + i * 1024),
                                                     16, voff, sbase + i * 1024, 0, 0);
        const Blk b = blk_of(kb);
#pragma unroll
        for (int k = 0; k < WS_OPIECES; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_of(b), (__attribute__((address_space(3))) void*)(stg + k * 1024), 16,
                                                     piece_off(b, k), b.soff, 0, 0);
    }

    // fragment addresses inside ring slot 0 (see tchain.hip)
    const int key = (l31 >> 1) & 7;
    unsigned fa0[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) fa0[s4] = lds0 + l31 * 128 + (((2 * s4 + hh) ^ key) << 4);

    f32x16 acc[WS_NT];
#pragma unroll
    for (int t = 0; t < WS_NT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    // The four waves leave every barrier together and would reach each LDS-DMA instruction of their (identical) streams in
    // the same cycle: one piece keeps the CU's texture-address path busy for 16 cycles (1 KiB at 64 B/clk), so three of
    // them queue.  Wave w starts 16 w cycles late instead: the requests interleave.
    auto skew = [&]() __attribute__((always_inline)) {
#if WS_SKEW
        if (wave & 1) asm volatile("s_nop 15");
        if (wave & 2) asm volatile("s_nop 15\n\ts_nop 15");
#endif
    };
    int g = kb * 5;  // global stage index (its slot: g & 1)
    for (int bi = kb; bi < ke; ++bi) {
        // block boundary: every copy issued so far has landed (the last stream issued weights only)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        skew();
        vec8 bop[20];
#pragma unroll
        for (int s = 0; s < 20; ++s) bop[s] = *reinterpret_cast<const vec8*>(stg + l31 * (WS_C * 2) + (2 * s + hh) * 16);
        const Blk nb = blk_of(bi + 1);
        const ws_rsrc_t nrs = rs_of(nb);
        int base5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) base5[j] = pm5[j] * nb.ldb + c5[j];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            if (s > 0) {
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // weights of this stage; the 5 operand pieces may fly on
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                skew();
            }
            const unsigned so = (g & 1) * WS_STAGE;
            const unsigned A0 = fa0[0] + so, A1 = fa0[1] + so, A2 = fa0[2] + so, A3 = fa0[3] + so;
#if defined(WS_DEBUG) && (WS_DEBUG & 2)
            const unsigned dso = (p.act & 2) ? 0x7ff00000u : (unsigned)((g + 1) * WS_STAGE + wave * (WS_PIECES * 1024));  // experiment: no weight traffic
#else
            const unsigned dso = (unsigned)((g + 1) * WS_STAGE + wave * (WS_PIECES * 1024));
#endif
            const unsigned dld = lds0 + ((g + 1) & 1) * WS_STAGE + wave * (WS_PIECES * 1024);
            unsigned t_dso;
            vec8 f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
#define WS_OUTS                                                                                                               \
            [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), [c5] "+a"(acc[5]),  \
            [c6] "+a"(acc[6]), [c7] "+a"(acc[7]), [c8] "+a"(acc[8]), [c9] "+a"(acc[9]), [f0] "=&v"(f0), [f1] "=&v"(f1),        \
            [f2] "=&v"(f2), [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6), [f7] "=&v"(f7), [f8] "=&v"(f8),    \
            [f9] "=&v"(f9), [f10] "=&v"(f10), [f11] "=&v"(f11), [dso] "=&s"(t_dso)
#define WS_INS                                                                                                                \
            [a0] "v"(A0), [a1] "v"(A1), [a2] "v"(A2), [a3] "v"(A3), [b0] "v"(bop[4 * s]), [b1] "v"(bop[4 * s + 1]),            \
            [b2] "v"(bop[4 * s + 2]), [b3] "v"(bop[4 * s + 3]), [vo] "v"(voff), [rs] "s"(wrs), [so0] "s"(dso), [ld0] "s"(dld)
            if (s < 4) {
                unsigned q[5];
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    q[j] = (unsigned)(base5[j] + s * 8 * nb.ldb) | (((inval[5 * s + j] >> nb.tap) & 1u) << 31);
#if defined(WS_DEBUG) && (WS_DEBUG & 1)
                if (p.act & 1) for (int j = 0; j < 5; ++j) q[j] |= 0x80000000u;   // experiment: no operand traffic
#endif
                const unsigned ob = stg0 + s * 5 * 1024;
                if constexpr (__is_same(T, f16))
                    asm volatile(TC_ASM_CONV_STAGE("f16") : WS_OUTS : WS_INS, [ors] "s"(nrs), [oso] "s"(nb.soff), [ob] "s"(ob),
                                 [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [q4] "v"(q[4]) : "memory", "scc");
                else
                    asm volatile(TC_ASM_CONV_STAGE("bf16") : WS_OUTS : WS_INS, [ors] "s"(nrs), [oso] "s"(nb.soff), [ob] "s"(ob),
                                 [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [q4] "v"(q[4]) : "memory", "scc");
            } else {
                if constexpr (__is_same(T, f16)) asm volatile(TC_ASM_GEMM_STAGE_DMA("f16") : WS_OUTS : WS_INS : "memory", "scc");
                else asm volatile(TC_ASM_GEMM_STAGE_DMA("bf16") : WS_OUTS : WS_INS : "memory", "scc");
            }
#undef WS_OUTS
#undef WS_INS
            g += 1;
        }
    }
    // MFMA results may be read 12+ states after issue; every copy (the speculative next weight stage) has landed
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]),
                   "+a"(acc[7]), "+a"(acc[8]), "+a"(acc[9]) :: "memory");

    // =============================== epilogue ===============================
    // accumulator arrangement: acc[t][4 q + r] = channel 32 t + 8 q + 4 hh + r of pixel m0w + l31
    const int n0 = nt * WS_C;
    const int m = m0w + l31;
    if (p.splitk > 1) {
        float* pp = p.partial + ((int64_t)zidx * p.M + m) * p.ldp + n0 + 4 * hh;
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(pp + 32 * t + 8 * q) =
                    make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        return;
    }
    __syncthreads();  // the LDS becomes the staging area of the epilogue (rows padded by 16 bytes, one slice per wave)
    char* const io = smem + wave * WS_IO_WAVE;
    constexpr int HI_ROW = WS_C * (int)sizeof(T), LO_ROW = WS_C * (int)sizeof(lo_t<T>);
    if (p.bias) {
        const float* b = p.bias + (int64_t)zb * p.zbias + n0 + 4 * hh;
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(b + 32 * t + 8 * q);
                acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
            }
    }
    if (p.rowadd) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        const T* ra = reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow + (int64_t)(m0w / p.rows_per_b) * p.ld_rowadd + n0 + 4 * hh;
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const t4 v = *reinterpret_cast<const t4*>(ra + 32 * t + 8 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += (float)v[r];
            }
    }
    // [32 rows][320 channels] tile of a row-major tensor (leading dimension ld elements) <-> staging slice, 16 bytes per lane
    auto stage_in = [&](const char* base, int64_t ld_bytes, int row_bytes) __attribute__((always_inline)) {
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024;
#pragma unroll
        for (int k0 = 0; k0 < 20; k0 += 5) {
            u32x4 v[5];
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < np) {
                    const int e = 64 * k + lane, r = e / cpr, c = e - r * cpr;
                    v[k - k0] = *reinterpret_cast<const u32x4*>(base + (int64_t)(m0w + r) * ld_bytes + c * 16);
                }
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < np) {
                    const int e = 64 * k + lane, r = e / cpr, c = e - r * cpr;
                    *reinterpret_cast<u32x4*>(io + r * (row_bytes + 16) + c * 16) = v[k - k0];
                }
        }
    };
    auto stage_out = [&](char* base, int64_t ld_bytes, int row_bytes) __attribute__((always_inline)) {
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024;
#pragma unroll
        for (int k = 0; k < 20; ++k)
            if (k < np) {
                const int e = 64 * k + lane, r = e / cpr, c = e - r * cpr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(io + r * (row_bytes + 16) + c * 16);
                *reinterpret_cast<u32x4*>(base + (int64_t)(m0w + r) * ld_bytes + c * 16) = v;
            }
    };
    if (p.res) {
        stage_in(reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.res) + (int64_t)zb * p.zres + n0), p.ldres * (int64_t)sizeof(T), HI_ROW);
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const t4 v = *reinterpret_cast<const t4*>(reinterpret_cast<const T*>(io + l31 * (HI_ROW + 16)) + 32 * t + 8 * q + 4 * hh);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += (float)v[r];
            }
        if (p.res_lo) {
            stage_in(reinterpret_cast<const char*>(reinterpret_cast<const lo_t<T>*>(p.res_lo) + (int64_t)zb * p.zres + n0),
                     p.ldres * (int64_t)sizeof(lo_t<T>), LO_ROW);
#pragma unroll
            for (int t = 0; t < WS_NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float l[4];
                    load_lo<4>(reinterpret_cast<const lo_t<T>*>(io + l31 * (LO_ROW + 16)) + 32 * t + 8 * q + 4 * hh, l);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += l[r];
                }
        }
    }
    if (p.out_scale != 1.0f) {
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] *= p.out_scale;
    }
#pragma unroll
    for (int t = 0; t < WS_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = from_f<T>(acc[t][4 * q + r]);
            *reinterpret_cast<t4*>(reinterpret_cast<T*>(io + l31 * (HI_ROW + 16)) + 32 * t + 8 * q + 4 * hh) = v;
        }
    stage_out(reinterpret_cast<char*>(reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout + n0), p.ldc * (int64_t)sizeof(T), HI_ROW);
    if (p.out_lo) {
#pragma unroll
        for (int t = 0; t < WS_NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo_t<T> b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = acc[t][4 * q + r];
                    b[r] = lo_from_f<lo_t<T>>(y - to_f(from_f<T>(y)));
                }
                __builtin_memcpy(reinterpret_cast<lo_t<T>*>(io + l31 * (LO_ROW + 16)) + 32 * t + 8 * q + 4 * hh, b, sizeof(b));
            }
        stage_out(reinterpret_cast<char*>(reinterpret_cast<lo_t<T>*>(p.out_lo) + (int64_t)zb * p.zout + n0),
                  p.ldc * (int64_t)sizeof(lo_t<T>), LO_ROW);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same kernel with EIGHT waves per workgroup = two per SIMD: wave w owns pixel group pg = w & 3 (32 pixels) and channel
// half ch = w >> 2 (160 output channels: 80 accumulator registers), so a SIMD runs two independent instruction streams
// and one can issue while the other waits for its fragment reads / copies.  (The one-wave-per-SIMD stream runs at ~50
// cycles per MFMA with NO memory traffic at all, DESIGN.md section 4: nothing but the order of its own instructions
// hides latency.)  Both waves of a pixel group multiply the same operand block (each reads it from the group's staging
// region) and split its 20 LDS-DMA pieces; the 40 weight pieces of a stage are split eight ways.  The operand pieces of
// the next block ride in stages 1 - 3 only (4 + 3 + 3 per wave): by the barrier of stage 1 every wave has its operand
// registers, so the staging region may be overwritten without a barrier of its own.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void wsconv8_body(const ur_igemm_desc& p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, ch = wave >> 2;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tiles = p.M >> 7, ntn = p.N / WS_C;
    const int sk = p.splitk > 1 ? p.splitk : 1;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % tiles; lid /= tiles;
    const int nt = lid % ntn; lid /= ntn;
    const int zidx = lid, zb = zidx / sk, ks = zidx - zb * sk;
    const int m0w = tile * 128 + pg * 32;
    const int HW = p.Hout * p.Wout;
    const int nb0 = p.c0 / WS_C, nbt0 = p.ct0 / WS_C, nbt1 = p.ct1 / WS_C;
    const int nblk = 9 * nb0 + nbt0 + nbt1;
    const int per = (nblk + sk - 1) / sk;
    const int kb = ks * per, ke = min(nblk, kb + per);
    constexpr int WP = 5;  // weight pieces per wave per stage
    const ws_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w)) + ((int64_t)zb * p.zw) * (int64_t)sizeof(T) +
            (int64_t)nt * nblk * 5 * WS_STAGE, 0, nblk * 5 * WS_STAGE, 0x00020000);
    const int voff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned stg0 = lds0 + 2 * WS_STAGE + pg * WS_STG;
    char* const stg = smem + 2 * WS_STAGE + pg * WS_STG;

    const int64_t zx = (int64_t)(p.zx_div > 1 ? zb / p.zx_div : zb) * p.zx;
    const int ldb0 = (int)p.ldx0 * (int)sizeof(T), ldbt0 = (int)p.ldt0 * (int)sizeof(T), ldbt1 = (int)p.ldt1 * (int)sizeof(T);
    const int back0 = (p.Wout + 1) * ldb0;
    const ws_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x0)) + zx * (int64_t)sizeof(T) - back0, 0, 0x80000000u, 0x00020000);
    const ws_rsrc_t rs_t0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.t0 ? p.t0 : p.x0)) + (int64_t)zb * p.zt0 * (int64_t)sizeof(T), 0, 0x80000000u, 0x00020000);
    const ws_rsrc_t rs_t1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.t1 ? p.t1 : p.x0)) + (int64_t)zb * p.zt1 * (int64_t)sizeof(T), 0, 0x80000000u, 0x00020000);

    // this wave's ten operand pieces k = 10 ch + lp: element e = 64 k + lane of the group's [32 rows][40 chunks] block;
    // k % 5 = lp % 5 and k / 5 = 2 ch + lp / 5, so five (row, chunk) pairs + 8-row steps describe them all
    int pm5[5], c5[5];
    unsigned inval[10];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int e = 64 * j + lane, r = e / 40;
        pm5[j] = m0w + r + 16 * ch;
        c5[j] = (e - r * 40) * 16;
    }
#pragma unroll
    for (int lp = 0; lp < 10; ++lp) {
        const int pm = pm5[lp % 5] + 8 * (lp / 5);
        const int pix = pm % HW, y = pix / p.Wout, x = pix - y * p.Wout;
        inval[lp] = (1u << 9) | (y == 0 ? 0007u : 0u) | (y == p.Hout - 1 ? 0700u : 0u) | (x == 0 ? 0111u : 0u) |
                    (x == p.Wout - 1 ? 0444u : 0u);
    }
    struct Blk { int src; int ldb; int tap; unsigned soff; };
    auto blk_of = [&](int bi) __attribute__((always_inline)) -> Blk {
        Blk b;
        if (bi >= ke) { b.src = 0; b.ldb = ldb0; b.tap = 9; b.soff = 0; return b; }
        if (bi < 9 * nb0) {
            const int cb = bi / 9, t = bi - 9 * cb;
            b.src = 0; b.ldb = ldb0; b.tap = t;
            b.soff = (unsigned)(back0 + ((t / 3 - 1) * p.Wout + (t % 3 - 1)) * ldb0 + cb * (WS_C * (int)sizeof(T)));
        } else if (bi < 9 * nb0 + nbt0) {
            b.src = 1; b.ldb = ldbt0; b.tap = 4; b.soff = (unsigned)((bi - 9 * nb0) * (WS_C * (int)sizeof(T)));
        } else {
            b.src = 2; b.ldb = ldbt1; b.tap = 4; b.soff = (unsigned)((bi - 9 * nb0 - nbt0) * (WS_C * (int)sizeof(T)));
        }
        return b;
    };
    auto rs_of = [&](const Blk& b) __attribute__((always_inline)) -> ws_rsrc_t { return b.src == 0 ? rs_x : (b.src == 1 ? rs_t0 : rs_t1); };
    auto piece_off = [&](const Blk& b, int lp) __attribute__((always_inline)) -> unsigned {
        const unsigned off = (unsigned)((pm5[lp % 5] + 8 * (lp / 5)) * b.ldb + c5[lp % 5]);
        return off | (((inval[lp] >> b.tap) & 1u) << 31);
    };
    {   // pipeline fill
        const int sbase = kb * 5 * WS_STAGE + wave * (WP * 1024);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(smem + ((kb * 5) & 1) * WS_STAGE + wave * (WP * 1024) + i * 1024),
                                                     16, voff, sbase + i * 1024, 0, 0);
        const Blk b = blk_of(kb);
#pragma unroll
        for (int lp = 0; lp < 10; ++lp)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_of(b), (__attribute__((address_space(3))) void*)(stg + (10 * ch + lp) * 1024), 16,
                                                     piece_off(b, lp), b.soff, 0, 0);
    }
    const int key = (l31 >> 1) & 7;
    unsigned fa0[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) fa0[s4] = lds0 + ch * (5 * 4096) + l31 * 128 + (((2 * s4 + hh) ^ key) << 4);

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    int g = kb * 5;
    for (int bi = kb; bi < ke; ++bi) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        vec8 bop[20];
#pragma unroll
        for (int s = 0; s < 20; ++s) bop[s] = *reinterpret_cast<const vec8*>(stg + l31 * (WS_C * 2) + (2 * s + hh) * 16);
        // two waves per SIMD = 256 registers per wave: the operand block lives in the AGPR half next to the accumulators
        // (MFMA B operands may be AGPRs), fragments / addresses / piece offsets in the VGPR half
        // (hipcc splits the 256 evenly, 128 + 128: accumulators 80 + the operands of stages 0 - 2 48 fill the AGPR half,
        // the operands of stages 3 / 4 stay in VGPRs)
#pragma unroll
        for (int s = 0; s < 12; ++s) asm volatile("" : "+a"(bop[s]));
        const Blk nb = blk_of(bi + 1);
        const ws_rsrc_t nrs = rs_of(nb);
        int base5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) base5[j] = pm5[j] * nb.ldb + c5[j];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            // operand pieces issued by the PREVIOUS stage's stream: none (s = 0, 1), 4 (s = 2), 3 (s = 3, 4)
            if (s == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (s == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (s >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (s > 0) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            const unsigned so = (g & 1) * WS_STAGE;
            const unsigned A0 = fa0[0] + so, A1 = fa0[1] + so, A2 = fa0[2] + so, A3 = fa0[3] + so;
            const unsigned dso = (unsigned)((g + 1) * WS_STAGE + wave * (WP * 1024));
            const unsigned dld = lds0 + ((g + 1) & 1) * WS_STAGE + wave * (WP * 1024);
            unsigned t_dso;
            vec8 f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
#define WS8_OUTS                                                                                                              \
            [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), [f0] "=&v"(f0),     \
            [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6), [f7] "=&v"(f7),    \
            [f8] "=&v"(f8), [f9] "=&v"(f9), [f10] "=&v"(f10), [f11] "=&v"(f11), [dso] "=&s"(t_dso)
#define WS8_INS_(BC)                                                                                                          \
            [a0] "v"(A0), [a1] "v"(A1), [a2] "v"(A2), [a3] "v"(A3), [b0] BC(bop[4 * s]), [b1] BC(bop[4 * s + 1]),              \
            [b2] BC(bop[4 * s + 2]), [b3] BC(bop[4 * s + 3]), [vo] "v"(voff), [rs] "s"(wrs), [so0] "s"(dso), [ld0] "s"(dld)
            if (s >= 1 && s <= 3) {
                constexpr int first[4] = {0, 0, 4, 7};  // this wave's local pieces: stage 1 -> 0..3, stage 2 -> 4..6, stage 3 -> 7..9
                const int lp0 = first[s], cnt = s == 1 ? 4 : 3;
                unsigned q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lp = lp0 + (j < cnt ? j : 0);
                    q[j] = (unsigned)(base5[lp % 5] + 8 * (lp / 5) * nb.ldb) | (((inval[lp] >> nb.tap) & 1u) << 31);
                }
                const unsigned ob = stg0 + (10 * ch + lp0) * 1024;
#define WS8_RUN(MACRO, BC, ...)                                                                                               \
                if constexpr (__is_same(T, f16)) asm volatile(MACRO("f16") : WS8_OUTS : WS8_INS_(BC) __VA_ARGS__ : "memory", "scc");  \
                else asm volatile(MACRO("bf16") : WS8_OUTS : WS8_INS_(BC) __VA_ARGS__ : "memory", "scc");
#define WS8_O4 , [ors] "s"(nrs), [oso] "s"(nb.soff), [ob] "s"(ob), [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3])
#define WS8_O3 , [ors] "s"(nrs), [oso] "s"(nb.soff), [ob] "s"(ob), [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2])
                if (s == 1) { WS8_RUN(TC_ASM_CONV8_STAGE_O4, "a", WS8_O4) }
                else if (s == 2) { WS8_RUN(TC_ASM_CONV8_STAGE_O3, "a", WS8_O3) }
                else { WS8_RUN(TC_ASM_CONV8_STAGE_O3, "v", WS8_O3) }
            } else if (s == 0) {
                WS8_RUN(TC_ASM_CONV8_STAGE_O0, "a", )
            } else {
                WS8_RUN(TC_ASM_CONV8_STAGE_O0, "v", )
            }
#undef WS8_RUN
#undef WS8_O4
#undef WS8_O3
#undef WS8_OUTS
#undef WS8_INS_
            g += 1;
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]) :: "memory");

    // =============================== epilogue: this wave's 160 channels of its 32 pixels ===============================
    const int n0 = nt * WS_C, nw = n0 + 160 * ch;
    const int m = m0w + l31;
    if (p.splitk > 1) {
        float* pp = p.partial + ((int64_t)zidx * p.M + m) * p.ldp + nw + 4 * hh;
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(pp + 32 * t + 8 * q) =
                    make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        return;
    }
    __syncthreads();
    // staging slice of the pixel GROUP (both waves): [32 rows][320 channels], rows padded by 16 bytes; a wave fills / reads
    // its channel half, the coalesced global side is split piece-wise between the two waves
    char* const io = smem + pg * WS_IO_WAVE;
    constexpr int HI_ROW = WS_C * (int)sizeof(T), LO_ROW = WS_C * (int)sizeof(lo_t<T>);
    if (p.bias) {
        const float* b = p.bias + (int64_t)zb * p.zbias + nw + 4 * hh;
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(b + 32 * t + 8 * q);
                acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
            }
    }
    if (p.rowadd) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        const T* ra = reinterpret_cast<const T*>(p.rowadd) + (int64_t)zb * p.zrow + (int64_t)(m0w / p.rows_per_b) * p.ld_rowadd + nw + 4 * hh;
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const t4 v = *reinterpret_cast<const t4*>(ra + 32 * t + 8 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += (float)v[r];
            }
    }
    auto stage_in = [&](const char* base, int64_t ld_bytes, int row_bytes) __attribute__((always_inline)) {
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024, half = (np + 1) / 2;
#pragma unroll
        for (int k0 = 0; k0 < 10; k0 += 5) {
            u32x4 v[5];
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < half && ch * half + k < np) {
                    const int e = 64 * (ch * half + k) + lane, r = e / cpr, c = e - r * cpr;
                    v[k - k0] = *reinterpret_cast<const u32x4*>(base + (int64_t)(m0w + r) * ld_bytes + c * 16);
                }
#pragma unroll
            for (int k = k0; k < k0 + 5; ++k)
                if (k < half && ch * half + k < np) {
                    const int e = 64 * (ch * half + k) + lane, r = e / cpr, c = e - r * cpr;
                    *reinterpret_cast<u32x4*>(io + r * (row_bytes + 16) + c * 16) = v[k - k0];
                }
        }
    };
    auto stage_out = [&](char* base, int64_t ld_bytes, int row_bytes) __attribute__((always_inline)) {
        const int cpr = row_bytes / 16, np = 32 * row_bytes / 1024, half = (np + 1) / 2;
#pragma unroll
        for (int k = 0; k < 10; ++k)
            if (k < half && ch * half + k < np) {
                const int e = 64 * (ch * half + k) + lane, r = e / cpr, c = e - r * cpr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(io + r * (row_bytes + 16) + c * 16);
                *reinterpret_cast<u32x4*>(base + (int64_t)(m0w + r) * ld_bytes + c * 16) = v;
            }
    };
    if (p.res) {
        stage_in(reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.res) + (int64_t)zb * p.zres + n0), p.ldres * (int64_t)sizeof(T), HI_ROW);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const t4 v = *reinterpret_cast<const t4*>(reinterpret_cast<const T*>(io + l31 * (HI_ROW + 16)) + 160 * ch + 32 * t + 8 * q + 4 * hh);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += (float)v[r];
            }
        __syncthreads();
        if (p.res_lo) {
            stage_in(reinterpret_cast<const char*>(reinterpret_cast<const lo_t<T>*>(p.res_lo) + (int64_t)zb * p.zres + n0),
                     p.ldres * (int64_t)sizeof(lo_t<T>), LO_ROW);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float l[4];
                    load_lo<4>(reinterpret_cast<const lo_t<T>*>(io + l31 * (LO_ROW + 16)) + 160 * ch + 32 * t + 8 * q + 4 * hh, l);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += l[r];
                }
            __syncthreads();
        }
    }
    if (p.out_scale != 1.0f) {
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] *= p.out_scale;
    }
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = from_f<T>(acc[t][4 * q + r]);
            *reinterpret_cast<t4*>(reinterpret_cast<T*>(io + l31 * (HI_ROW + 16)) + 160 * ch + 32 * t + 8 * q + 4 * hh) = v;
        }
    __syncthreads();
    stage_out(reinterpret_cast<char*>(reinterpret_cast<T*>(p.out) + (int64_t)zb * p.zout + n0), p.ldc * (int64_t)sizeof(T), HI_ROW);
    if (p.out_lo) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo_t<T> b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = acc[t][4 * q + r];
                    b[r] = lo_from_f<lo_t<T>>(y - to_f(from_f<T>(y)));
                }
                __builtin_memcpy(reinterpret_cast<lo_t<T>*>(io + l31 * (LO_ROW + 16)) + 160 * ch + 32 * t + 8 * q + 4 * hh, b, sizeof(b));
            }
        __syncthreads();
        stage_out(reinterpret_cast<char*>(reinterpret_cast<lo_t<T>*>(p.out_lo) + (int64_t)zb * p.zout + n0),
                  p.ldc * (int64_t)sizeof(lo_t<T>), LO_ROW);
    }
}

// Non-template entry points: hipcc (ROCm 7.2) silently drops the HOST stub of a __global__ template with this body (its
// host pass fails to substitute the local-struct lambdas and reports nothing), which leaves the launch unresolved.
__global__ void __launch_bounds__(256, 1) wsconv_kernel_f16(const ur_igemm_desc p) { wsconv_body<f16>(p); }
__global__ void __launch_bounds__(256, 1) wsconv_kernel_bf16(const ur_igemm_desc p) { wsconv_body<bf16>(p); }
__global__ void __launch_bounds__(512, 2) wsconv8_kernel_f16(const ur_igemm_desc p) { wsconv8_body<f16>(p); }
__global__ void __launch_bounds__(512, 2) wsconv8_kernel_bf16(const ur_igemm_desc p) { wsconv8_body<bf16>(p); }

// 0 when the descriptor fits this kernel, UR_E_UNSUPPORTED otherwise (the caller falls back to an LDS-tiled build)
int wsconv_supported(const ur_igemm_desc& d) {
    if (d.taps != 9 || d.stride != 1 || d.ups || d.pad != 1 || d.x1 || d.c1) return UR_E_UNSUPPORTED;
    if (d.Hin != d.Hout || d.Win != d.Wout) return UR_E_UNSUPPORTED;
    const int hw = d.Hout * d.Wout;
    if (hw % 128 || d.M % 128 || d.M != d.B * hw) return UR_E_UNSUPPORTED;
    if (d.N % WS_C || d.n_store != d.N) return UR_E_UNSUPPORTED;
    if (d.c0 <= 0 || d.c0 % WS_C || d.ct0 % WS_C || d.ct1 % WS_C) return UR_E_UNSUPPORTED;
    if (d.c0 > WS_C && d.cblock != WS_C) return UR_E_UNSUPPORTED;
    if (d.c0 == WS_C && d.cblock != 0 && d.cblock != WS_C) return UR_E_UNSUPPORTED;
#ifdef WS_DEBUG
    if (d.act != 0 && d.act < 100) return UR_E_UNSUPPORTED;
#else
    if (d.act != 0) return UR_E_UNSUPPORTED;
#endif
    if (d.K != 9 * d.c0 + d.ct0 + d.ct1) return UR_E_UNSUPPORTED;
    if (d.rowadd && (d.rows_per_b % 128 || (d.ld_rowadd & 3))) return UR_E_UNSUPPORTED;
    if ((d.ldc | d.ldres) & 7) return UR_E_UNSUPPORTED;
    // descriptor offsets are 31-bit: bytes of the largest source (+ the (W + 1)-pixel back-off)
    const int64_t esz = 2;
    if (((int64_t)d.M + d.Wout + 2) * d.ldx0 * esz >= (1ll << 31)) return UR_E_UNSUPPORTED;
    if ((int64_t)d.M * (d.ldt0 > d.ldt1 ? d.ldt0 : d.ldt1) * esz >= (1ll << 31)) return UR_E_UNSUPPORTED;
    if ((d.ldx0 | d.ldt0 | d.ldt1) & 7) return UR_E_UNSUPPORTED;
    if ((int64_t)(d.K / 64) * WS_STAGE >= (1ll << 31)) return UR_E_UNSUPPORTED;
    return 0;
}

static int launch_ws(const ur_igemm_desc& d, hipStream_t s, bool half) {
    static std::atomic<uint64_t> done16{0}, donebf{0}, done16w8{0}, donebfw8{0};
    const int sk = d.splitk > 1 ? d.splitk : 1;
    const int wgs = (d.M / 128) * (d.N / WS_C) * (d.zbatch > 1 ? d.zbatch : 1) * sk;
    if (d.tile == UR_TILE_WS320_W8) {
        if (half) {
            set_lds_limit_once(done16w8, reinterpret_cast<const void*>(&wsconv8_kernel_f16), WS_LDS);
            hipLaunchKernelGGL(wsconv8_kernel_f16, dim3(wgs), dim3(512), WS_LDS, s, d);
        } else {
            set_lds_limit_once(donebfw8, reinterpret_cast<const void*>(&wsconv8_kernel_bf16), WS_LDS);
            hipLaunchKernelGGL(wsconv8_kernel_bf16, dim3(wgs), dim3(512), WS_LDS, s, d);
        }
    } else if (half) {
        set_lds_limit_once(done16, reinterpret_cast<const void*>(&wsconv_kernel_f16), WS_LDS);
        hipLaunchKernelGGL(wsconv_kernel_f16, dim3(wgs), dim3(256), WS_LDS, s, d);
    } else {
        set_lds_limit_once(donebf, reinterpret_cast<const void*>(&wsconv_kernel_bf16), WS_LDS);
        hipLaunchKernelGGL(wsconv_kernel_bf16, dim3(wgs), dim3(256), WS_LDS, s, d);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// main pass only: the caller (igemm.hip) runs igemm_splitk_reduce behind it when d.splitk > 1
int wsconv_launch(const ur_igemm_desc& d, hipStream_t s) {
    const int rc = wsconv_supported(d);
    if (rc) return rc;
    if (d.dtype == UR_DT_F16) return launch_ws(d, s, true);
    if (d.dtype == UR_DT_BF16) return launch_ws(d, s, false);
    return UR_E_BADARG;
}

}  // namespace ur
