"""Python wrappers over the C ABI (``include/ur_kernels.h``): tensors in, tensors out, kernels enqueued on
PyTorch's current HIP stream.  PyTorch only provides device memory and the stream here; all arithmetic
happens in ``liburhip.so``.  Activations are NHWC: ``[B, H, W, C]`` == token matrix ``[B*H*W, C]``.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Optional, Tuple

import torch

from . import _experiments as X
from . import _lib
from ._lib import AttnDesc, IGemmDesc, check

DT = {torch.float16: 0, torch.bfloat16: 1}
DT_ANY = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64 = 0, 1, 2, 3
TILE_128x128_S3, TILE_128x64_S2, TILE_64x64_S4, TILE_64x64_S2 = 4, 5, 6, 7
# tile id -> (BM, BN, relative efficiency guess for the analytic planner, LDS ring depth)
_TILES = {TILE_128x128: (128, 128, 1.0, 2), TILE_128x64: (128, 64, 0.85, 3), TILE_64x64: (64, 64, 0.6, 3),
          TILE_128x128_S3: (128, 128, 0.9, 3), TILE_128x64_S2: (128, 64, 0.7, 2), TILE_64x64_S4: (64, 64, 0.6, 4),
          TILE_64x64_S2: (64, 64, 0.5, 2), 8: (256, 128, 1.1, 2), 9: (128, 320, 1.1, 2), 10: (128, 256, 1.1, 2),
          11: (256, 256, 1.2, 2),
          # register-staged loader variants (global_load -> VGPR -> ds_write), "stages" = "r"
          12: (64, 64, 0.6, "r"), 13: (128, 64, 0.85, "r"), 14: (128, 128, 1.0, "r"), 15: (128, 320, 1.1, "r"),
          16: (256, 128, 1.1, "r"),
          # few-wave workgroups: every wave owns a full 64x64 tile (0.5 KB of LDS reads per MFMA instead of 1.25)
          17: (64, 64, 0.8, "2w1"), 18: (128, 64, 0.9, "2w2"), 19: (64, 64, 0.8, "3w1"), 20: (64, 128, 0.9, "2w2n"),
          21: (64, 64, 0.8, "4w1"),
          # the same tiles on the 32x32x16 MFMA (UR_TILE_*_M32)
          22: (128, 320, 1.2, "2m32"), 23: (128, 128, 1.1, "2m32"), 24: (128, 64, 0.9, "2m32"), 25: (128, 64, 0.9, "3m32"),
          26: (64, 64, 0.7, "2m32"), 27: (64, 64, 0.7, "3m32"), 28: (256, 256, 1.3, "2m32"), 29: (256, 128, 1.2, "2m32"),
          30: (128, 256, 1.2, "2m32"),
          # wave-specialised builds: n dedicated loader waves (UR_TILE_*_L<n>); 39 is reserved / not instantiated
          31: (128, 320, 1.3, "2L2"), 32: (128, 320, 1.3, "2L4"), 33: (128, 128, 1.2, "2L2"), 34: (128, 128, 1.2, "3L2"),
          35: (128, 64, 1.0, "2L1"), 36: (128, 64, 1.0, "3L2"), 37: (64, 64, 0.8, "3L1"), 38: (256, 128, 1.3, "2L2"),
          40: (128, 256, 1.3, "2L2"), 41: (128, 256, 1.2, 3), 42: (128, 320, 1.2, "2w8m32"), 43: (256, 320, 1.2, "2w16m32"), 44: (128, 160, 1.0, "2m32"), 45: (128, 160, 1.0, "3m32"),
          46: (64, 320, 0.9, "2m32"),
          # weight-streaming conv (csrc/wsconv.hip): ``w`` is the stage-image stream of wsconv_images()
          47: (128, 320, 1.4, "ws"), 48: (128, 320, 1.4, "ws8"),
          # 8-wave ping-pong builds (csrc/igemm_pp.hip): "pp<ring slots>"
          49: (128, 320, 1.5, "pp5"), 50: (128, 320, 1.5, "pp4"), 51: (256, 128, 1.4, "pp5"), 52: (128, 256, 1.4, "pp5"),
          53: (256, 256, 1.5, "pp4"), 54: (128, 128, 1.2, "pp5"), 55: (256, 320, 1.5, "pp4"),
          # round 6: few waves with big per-wave tiles (64 x 160 / 64 x 128 per wave, one or two waves per SIMD)
          56: (256, 160, 1.3, "2w4m32"), 57: (256, 320, 1.4, "2w8m32"), 58: (128, 320, 1.3, "2w4m32"), 59: (256, 128, 1.2, "2w4m32"),
          60: (256, 256, 1.4, "2w8m32"), 61: (256, 320, 1.4, "2w10")}
TILE_PP_128x320, TILE_PP_128x320_S4, TILE_PP_256x128, TILE_PP_128x256, TILE_PP_256x256, TILE_PP_128x128, TILE_PP_256x320 = range(49, 56)
TILE_WS320, TILE_WS320_W8 = 47, 48
# which build ``conv3x3(ws=...)`` launches: 8 waves per workgroup (two instruction streams per SIMD) or 4 (one)
WSCONV_TILE = TILE_WS320_W8 if X.number("wsconv_waves", 8) == 8 else TILE_WS320
_PLANNER_TILES = (TILE_128x128, TILE_128x64, TILE_64x64)

_zero_pages = {}
_tune_table = None
_plan_cache = {}

# Optional per-launch timing (bench.py's roofline leg): when a list is installed with ``profile_into``, every
# wrapper brackets its launch with HIP events recorded on the SAME stream the kernel is enqueued on and appends
# (kernel-class key, algorithmic flops, algorithmic bytes, start, end).
_prof = None
_prof_by_shape = False


def profile_into(records, by_shape: bool = False):
    global _prof, _prof_by_shape
    _prof = records
    _prof_by_shape = by_shape


def _prof_begin():
    if _prof is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    return e


def _prof_end(e0, key, flops, nbytes):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(torch.cuda.current_stream())
    _prof.append((key, float(flops), float(nbytes), e0, e1))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device as an integer handle.  ``torch.cuda.current_stream()`` builds a
    Stream object through several Python layers (~9 us, once per launch: 8 ms of a host-bound eager training step); the
    raw accessor is the same value in ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


# zero region the implicit-GEMM loaders read padding rows from: 1 MiB so that every (workgroup, wave) has its own line
# (ur_igemm_desc.zero_page_bytes); UR_EXPERIMENT=zero_page_bytes=4096 restores the single hot page (A/B)
ZERO_PAGE_BYTES = X.number("zero_page_bytes", 1 << 20)


def zero_page(device) -> torch.Tensor:
    key = (device.type, device.index)
    z = _zero_pages.get(key)
    if z is None:
        z = torch.zeros(ZERO_PAGE_BYTES, dtype=torch.uint8, device=device)
        _zero_pages[key] = z
    return z


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("uni_renderer_amd ops run on an MI355X HIP device only (got a CPU tensor); "
                           "there is no CPU fallback")


# ---------------------------------------------------------------------------------------------
# tile / split-K planning
# ---------------------------------------------------------------------------------------------
def load_tuning_table(path: Optional[str] = None):
    """Optional table {"M,N,K,taps,z": [tile, splitk]} measured on MI355X by tools/tune_igemm.py."""
    global _tune_table
    if path is None:  # UR_IGEMM_TUNING: an alternative table (A/B runs of tools/tune_igemm.py output)
        path = os.environ.get("UR_IGEMM_TUNING") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "igemm_tuning.json")
    _tune_table = {}
    if os.path.exists(path):
        with open(path) as f:
            _tune_table = {k: tuple(v) for k, v in json.load(f).items()}
    _plan_cache.clear()
    return _tune_table


_site = None  # call-site tag of the launch being planned (fused.py sets it): lets the in-situ tuner give the SAME problem
              # shape different tiles at different places of the step ("M,N,K,taps,z@site" rows override "M,N,K,taps,z")


def set_site(name: Optional[str]):
    global _site
    _site = name


_plan_rows = None  # while set: 1x1 GEMMs are planned as if they had this many rows (see plan_rows_as)


class plan_rows_as:
    """Context: plan every Linear launched inside as if it had ``m`` rows.  A GEMM row's arithmetic depends on the (tile, split-K)
    choice only, not on how many rows the launch has -- so a table computed for n x B rows under ``plan_rows_as(B)`` holds, row for
    row, the bits the B-row launches of the individual steps produce (hoist.time_tables)."""

    def __init__(self, m: int):
        self.m = int(m)

    def __enter__(self):
        global _plan_rows
        self.prev, _plan_rows = _plan_rows, self.m
        return self

    def __exit__(self, *exc):
        global _plan_rows
        _plan_rows = self.prev
        return False


def plan_igemm(M: int, N: int, K: int, taps: int = 1, zbatch: int = 1) -> Tuple[int, int]:
    """(tile, splitk) for an implicit GEMM.  Measured table first, analytic model otherwise."""
    global _tune_table
    key = (M, N, K, taps, zbatch, _site)
    hit = _plan_cache.get(key)
    if hit is not None:
        return hit
    if _tune_table is None:
        load_tuning_table()
    t = _tune_table.get(f"{M},{N},{K},{taps},{zbatch}@{_site}") if _site else None
    if t is None:
        t = _tune_table.get(f"{M},{N},{K},{taps},{zbatch}")
    if t is None:
        per_cu = 2.5e15 / 256 * 0.4
        best, best_t = (TILE_64x64, 1), float("inf")
        for tile in _PLANNER_TILES:
            bm, bn, eff, _ = _TILES[tile]
            wgs = math.ceil(M / bm) * math.ceil(N / bn) * zbatch
            for sk in (1, 2, 4, 8):
                if sk > 1 and (zbatch > 4 or K // 64 < 8 * sk):
                    continue
                waves = math.ceil(wgs * sk / 256)
                tt = waves * bm * bn * (K / sk) * 2 / (per_cu * eff)
                if sk > 1:
                    tt += M * N * 4 * (sk + 1) / 4e12 + 3e-6
                if tt < best_t:
                    best, best_t = (tile, sk), tt
        t = best
    _plan_cache[key] = t
    return t


# ---------------------------------------------------------------------------------------------
# implicit GEMM
# ---------------------------------------------------------------------------------------------
DXS_TRACE = None  # a list to collect ur_igemm_uses_dxs() of every launch (tests)


def dxs_active() -> bool:
    """The dx-tap-sharing conv kernel (csrc/igemm_dxs.hip) is built in (`make DXS=1`) and switched on (UR_DXS=1)."""
    d = IGemmDesc()
    d.taps, d.stride, d.pad, d.tile = 9, 1, 1, 9
    d.Hin = d.Win = d.Hout = d.Wout = 64
    d.M = 64 * 64
    return bool(_lib.load().ur_igemm_uses_dxs(C.byref(d)))


def igemm(*, x0, w, out, M, N, K, c0, c1=0, x1=None, ldx0, ldx1=0, ldw, ldc, taps=1, conv=None, stride=1, ups=0,
          bias=None, rowadd=None, rows_per_b=0, res=None, ldres=0, n_store=0, act=ACT_NONE, out_scale=1.0,
          zbatch=1, zx=0, zw=0, zout=0, zx1=0, zbias=0, zrow=0, zres=0, zx_div=1, tile=None, splitk=None,
          res_lo=None, out_lo=None, cblock=0, t0=None, t1=None, ldt0=0, ldt1=0, zt0=0, zt1=0, ct0=0, ct1=0, pad=1,
          out_vt=None, vt_n0=0, vt_rows=0, zvt=0, gn=None):
    """``gn`` = (gamma, beta, per-z stride, eps, groups, silu): split-K launches only -- the second pass normalises
    (ur_igemm_splitk_gn) and ``out`` receives GroupNorm(conv) instead of the conv."""
    _require_gpu(x0)
    lib = _lib.load()
    if tile is None or splitk is None:
        pt, ps = plan_igemm(_plan_rows if (_plan_rows is not None and taps == 1) else M, N, K, taps, zbatch)
        tile = pt if tile is None else tile
        splitk = ps if splitk is None else splitk
    d = IGemmDesc()  # zero-initialised: only what differs from 0 / NULL is written (every field set is host time,
    d.x0, d.w, d.out = x0.data_ptr(), w.data_ptr(), out.data_ptr()  # ~1300 times per eager training step)
    if x1 is not None:
        d.x1, d.ldx1, d.c1, d.zx1 = x1.data_ptr(), ldx1, c1, zx1
    if bias is not None:
        d.bias, d.zbias = bias.data_ptr(), zbias
    if rowadd is not None:
        d.rowadd, d.ld_rowadd, d.rows_per_b, d.zrow = rowadd.data_ptr(), rowadd.stride(0), rows_per_b, zrow
    if res is not None:
        d.res, d.ldres, d.zres = res.data_ptr(), ldres, zres
        if res_lo is not None:
            d.res_lo = res_lo.data_ptr()
    if out_lo is not None:
        d.out_lo = out_lo.data_ptr()
    if cblock:
        d.cblock = cblock
    if out_vt is not None:  # columns >= vt_n0 leave transposed: out_vt[sample][n - vt_n0][token] (include/ur_kernels.h)
        d.out_vt, d.ldvt, d.vt_bstride, d.zvt = out_vt.data_ptr(), out_vt.stride(-2), out_vt.stride(0), zvt
        d.vt_n0, d.vt_rows = vt_n0, vt_rows
    if t0 is not None or t1 is not None:
        d.t0, d.t1, d.ldt0, d.ldt1, d.zt0, d.zt1, d.ct0, d.ct1 = _ptr(t0), _ptr(t1), ldt0, ldt1, zt0, zt1, ct0, ct1
    d.zero_page = zero_page(x0.device).data_ptr()
    d.zero_page_bytes = ZERO_PAGE_BYTES
    d.ldx0, d.ldw, d.ldc, d.c0 = ldx0, ldw, ldc, c0
    if zbatch > 1 or zx or zw or zout:
        d.zx, d.zw, d.zout = zx, zw, zout
    d.zx_div = zx_div
    if conv is not None:
        d.B, d.Hin, d.Win, d.Hout, d.Wout = conv
    d.taps, d.stride, d.ups, d.pad = taps, stride, ups, pad
    d.M, d.N, d.K = M, N, K
    if n_store:
        d.n_store = n_store
    d.act, d.out_scale = act, out_scale
    d.zbatch, d.splitk, d.tile, d.dtype = zbatch, splitk, tile, DT[x0.dtype]
    part = None
    if splitk > 1:
        part = torch.empty(int(lib.ur_igemm_partial_floats(C.byref(d))), dtype=torch.float32, device=x0.device)
        d.partial = part.data_ptr()
    if bias is not None and bias.dtype != torch.float32:
        raise RuntimeError("bias must be fp32")
    if DXS_TRACE is not None:  # tests: which launches take the dx-tap-sharing conv kernel (csrc/igemm_dxs.hip)
        DXS_TRACE.append(int(lib.ur_igemm_uses_dxs(C.byref(d))))
    e0 = _prof_begin()
    if gn is not None:
        if splitk <= 1:
            raise RuntimeError("igemm(gn=...) is the fused second pass of a split-K launch")
        gam, bet, zgn, eps, groups, silu = gn
        check(lib.ur_igemm_splitk_gn(C.byref(d), gam.data_ptr(), bet.data_ptr(), int(zgn), float(eps), int(groups), int(silu),
                                     _stream()), "ur_igemm_splitk_gn")
    else:
        check(lib.ur_igemm(C.byref(d), _stream()), "ur_igemm")
    if e0 is not None:
        el = x0.element_size()
        z = max(zbatch, 1)
        key = (f"igemm_{_TILES[tile][0]}x{_TILES[tile][1]}s{_TILES[tile][3]}_{'conv3x3' if taps == 9 else 'gemm'}"
               + ("_splitk" if splitk > 1 else "") + ("_gn" if gn is not None else ""))
        if _prof_by_shape:
            key += f"|M{M}_N{N}_K{K}_z{z}_sk{splitk}"
        # algorithmic bytes: every operand once -- activations (the conv reads each input pixel once: M*stride^2/4^ups
        # pixels), weights, output, plus the residual operand and the low parts of the (hi, lo) stream when present
        in_rows = M * (stride * stride if taps == 9 else 1) / (4 ** ups if taps == 9 else 1)
        n_out = N / 2 if act == ACT_GEGLU else N
        lo_el = 1 if x0.dtype == torch.float16 else el  # low parts: one e5m2 byte (fp16) / bf16
        lo_bytes = M * n_out * lo_el * ((res_lo is not None) + (out_lo is not None))
        _prof_end(e0, key, 2.0 * M * N * K * z,
                  ((in_rows * (K - ct0 - ct1) / taps + M * (ct0 + ct1) + N * K + M * n_out * (1 + (res is not None))) * el
                   + lo_bytes) * z)
    return out


# ---------------------------------------------------------------------------------------------
# (hi, lo) residual stream (include/ur_kernels.h): the low part of a residual-stream tensor travels as the attribute
# ``.lo`` of the ordinary (hi) tensor.  Views / slices drop it, which is the safe default: a consumer that does not
# know about it just sees the ordinary rounded tensor.
# ---------------------------------------------------------------------------------------------
PRECISE_RESIDUAL = os.environ.get("UR_PRECISE_RESIDUAL", "1") != "0"  # default of the ``hilo`` producers below


def lo_of(t):
    return getattr(t, "lo", None) if t is not None else None


def view_hilo(t, *shape):
    """``t.view(*shape)`` that keeps the low part attached."""
    v = t.view(*shape)
    lo = lo_of(t)
    if lo is not None:
        v.lo = lo.view(*shape)
    return v


def lo_dtype(dtype):
    """Storage type of the low parts: one e5m2 byte (= the high byte of the fp16 encoding of the remainder; three
    significant bits put the pair at 2^-14 relative) for fp16 streams, bf16 for bf16 streams (csrc/ur_common.h)."""
    return torch.uint8 if dtype == torch.float16 else dtype


def lo_float(lo: torch.Tensor) -> torch.Tensor:
    """fp32 values of a low-part tensor (tests / diagnostics)."""
    if lo.dtype == torch.uint8:
        return (lo.to(torch.int16) << 8).view(torch.float16).float()
    return lo.float()


def lo_encode(v: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 remainders -> low-part storage of a ``dtype`` stream (round to nearest even; tests / glue)."""
    if dtype != torch.float16:
        return v.to(dtype)
    bits = v.to(torch.float16).view(torch.int16).to(torch.int32) & 0xFFFF
    return ((bits + 0x7F + ((bits >> 8) & 1)) >> 8).to(torch.uint8)


def _with_lo(out, want: bool):
    if want:
        out.lo = torch.empty(out.shape, dtype=lo_dtype(out.dtype), device=out.device)
    return out


def linear(x, w, bias=None, *, x1=None, res=None, act=ACT_NONE, out_scale=1.0, rowadd=None, rows_per_b=0, out=None,
           tile=None, splitk=None, streams=1, res_zstride=None, hilo=False, res_lo=None, vt_cols=0, vt_tokens=0):
    """y[..., N] = epilogue(x[..., K] @ w[N, K]^T).  ``x1``: second source concatenated along K.
    ``vt_cols`` = Cv > 0 (with ``vt_tokens`` = T rows per sample, a multiple of 64): the LAST Cv rows of ``w`` are a
    value projection whose result leaves TRANSPOSED -- returns ``(y[..., N - Cv], vt[samples, Cv, T])`` from one launch
    (``out_scale`` applies to y only): a self-attention's q | k | v projection.
    ``hilo``: also produce the rounding remainder (``y.lo``); the low part of ``res`` (``res.lo`` or ``res_lo``) is
    added when present.

    ``streams=S`` > 1 runs S independent problems of one shape as ONE grouped launch: x (x1, res, out) hold the
    S row-blocks back to back, ``w`` is [S, N, K], ``bias`` [S, N], ``rowadd`` has its rows grouped per stream.
    ``res_zstride`` overrides the per-stream element stride of ``res`` (the exchange GEMMs read the OTHER stream's
    tensor as residual: pointer at the second half, negative stride)."""
    K0 = x.shape[-1]
    K1 = x1.shape[-1] if x1 is not None else 0
    M = x.numel() // K0 // streams
    N = w.shape[-2]
    n_out = N // 2 if act == ACT_GEGLU else N
    vt, vkw = None, {}
    if vt_cols:
        if act != ACT_NONE or res is not None or rowadd is not None or vt_tokens <= 0 or vt_tokens % 64 or (M % vt_tokens):
            raise ValueError("linear(vt_cols=...): plain projection of whole samples with a multiple of 64 tokens")
        n_out = N - vt_cols
        vt = torch.empty(streams * (M // vt_tokens), vt_cols, vt_tokens, dtype=x.dtype, device=x.device)
        vkw = dict(out_vt=vt, vt_n0=n_out, vt_rows=vt_tokens, zvt=(M // vt_tokens) * vt_cols * vt_tokens, n_store=n_out)
    if out is None:
        out = torch.empty(*x.shape[:-1], n_out, dtype=x.dtype, device=x.device)
    _with_lo(out, hilo and act != ACT_GEGLU)
    if res_lo is None:
        res_lo = lo_of(res)
    z = {}
    if streams > 1:
        z = dict(zbatch=streams, zx=M * K0, zx1=M * K1, zw=w.stride(0), zout=M * n_out,
                 zbias=(bias.stride(0) if bias is not None else 0),
                 zres=(res_zstride if res_zstride is not None else M * n_out) if res is not None else 0,
                 zrow=(rowadd.stride(0) * (rowadd.shape[0] // streams)) if rowadd is not None else 0)
    igemm(x0=x, x1=x1, w=w, out=out, M=M, N=N, K=K0 + K1, c0=K0, c1=K1, ldx0=K0, ldx1=K1, ldw=w.stride(-2), ldc=n_out,
          bias=bias, res=res, ldres=(n_out if res is not None else 0), act=act, out_scale=out_scale, rowadd=rowadd,
          rows_per_b=rows_per_b, tile=tile, splitk=splitk, res_lo=res_lo, out_lo=lo_of(out), **z, **vkw)
    return out if vt is None else (out, vt)


CONV_CBLOCK = X.number("conv_cblock", 320)
FOLD_SHORTCUT = X.flag("fold_shortcut", True)  # resnet conv_shortcut as the 1x1 tail of conv2 (conv3x3 ``tail``)


def conv_cblock(cin: int) -> int:
    """Channel-block size of the block-outer K order for a ``cin``-channel 3x3 conv (0 = tap-outer order): wide inputs
    (640 .. 2560 channels) walk K as (block of 320 channels, tap) so that the nine taps of a block re-read their input
    lines out of the XCD's L2 (include/ur_kernels.h, ``cblock``).  Weight packer and launch must agree."""
    return CONV_CBLOCK if (CONV_CBLOCK > 0 and cin > CONV_CBLOCK and cin % CONV_CBLOCK == 0) else 0


# Weight-streaming conv kernel (csrc/wsconv.hip) in the executors: OFF by default.  Isolated it matches or beats the tuned
# LDS-tiled build by 2-6 % from K = 5760 up (tools/wsconv_bench.py), but inside the step -- weights cold in L2, one stage
# of prefetch with one wave per SIMD -- the same launches are ~10 % slower: 11.88 -> 12.20 ms per step with it on for
# K >= 5000 (tools/experiments/r03_run13.sh, two alternating repetitions on one box).  ``conv3x3(..., ws=...)`` still takes it
# explicitly (tests/test_wsconv_gpu.py).
WSCONV = X.flag("wsconv", False)
WS_C = 320


def wsconv_images(w: torch.Tensor, n_out: Optional[int] = None) -> torch.Tensor:
    """Packed conv weights [Npad >= N, K] (the cblock = 320 K order of pack_conv3x3, tail columns appended) -> the weight
    stream of csrc/wsconv.hip: [N / 320][K / 64] stage images of 320 rows x 64 k (40960 bytes), the eight 16-byte chunks
    of row r stored at position c ^ ((r >> 1) & 7) (tchain.py, ``_swizzle_rows``)."""
    N = n_out if n_out is not None else w.shape[0]
    K = w.shape[1]
    if N % WS_C or K % WS_C:
        raise RuntimeError("wsconv_images: N and K must be multiples of 320")
    v = w[:N].reshape(N // WS_C, WS_C, K // 64, 8, 8).permute(0, 2, 1, 3, 4)      # [nt, stage, row, chunk, 8]
    r = torch.arange(WS_C, device=w.device)
    pos = torch.arange(8, device=w.device)[None, :] ^ ((r >> 1) & 7)[:, None]     # position c' holds chunk c' ^ key
    idx = pos[None, None, :, :, None].expand(v.shape[0], v.shape[1], WS_C, 8, 8)
    return torch.gather(v, 3, idx).reshape(-1).contiguous()


def pp_built() -> bool:
    """The 8-wave ping-pong tiles (TILE_PP_*) are an opt-in build of the library (``make PP=1``, csrc/Makefile)."""
    return bool(_lib.load().ur_has_pp())


def wsconv_built() -> bool:
    """The weight-streaming conv tiles are an opt-in build of the library (``make WSCONV=1``, csrc/Makefile)."""
    return bool(_lib.load().ur_has_wsconv())


def wsconv_ok(x, N, *, x1=None, stride=1, ups=False, pad=1, tail=None, streams=1) -> bool:
    """Whether conv3x3 over ``x`` can take the weight-streaming kernel (mirror of wsconv_supported, csrc/wsconv.hip)."""
    if x1 is not None or stride != 1 or ups or pad != 1:
        return False
    Bt, H, W, C0 = x.shape
    if (H * W) % 128 or N % WS_C or C0 % WS_C or (C0 > WS_C and conv_cblock(C0) != WS_C):
        return False
    if tail is not None and any(t is not None and t.shape[-1] % WS_C for t in tail):
        return False
    cmax = max([C0] + [t.shape[-1] for t in (tail or ()) if t is not None])
    return (Bt // streams * H * W + W + 2) * cmax * 2 < 2 ** 31


WSCONV_MIN_K = X.number("wsconv_min_k", 5000)


def wsconv_prefer(x, N, K, **kw) -> bool:
    """Policy: the weight-streaming kernel where it measured faster than the tuned LDS-tiled build (tools/wsconv_bench.py:
    +2 .. +6 % from K = 5760 up, -2 .. -20 % below: its per-workgroup prologue / epilogue is longer)."""
    return WSCONV and K >= WSCONV_MIN_K and wsconv_ok(x, N, **kw) and wsconv_built()


def wsconv_splitk(M, N, K, zbatch) -> int:
    """Split-K of the weight-streaming conv: enough workgroups (128 pixels x 320 channels each) for the 256 CUs, never
    fewer than three 320-channel K blocks per slice."""
    t = _ws_table.get((M, N, K, zbatch))
    if t is not None:
        return t
    wgs = (M // 128) * (N // WS_C) * zbatch
    nblk = K // WS_C
    sk = 1
    while wgs * sk < 200 and nblk // (sk * 2) >= 3:
        sk *= 2
    return sk


_ws_table: dict = {}


def conv3x3(x, w, bias=None, *, x1=None, stride=1, ups=False, rowadd=None, res=None, out_scale=1.0, n_out=None,
            tile=None, splitk=None, streams=1, hilo=False, cblock=0, tail=None, pad=1, ws=None, gn=None):
    """3x3 conv, pad 1, over NHWC ``x`` (optionally cat(x, x1) on channels, optionally after a nearest-2x
    upsample).  ``w`` is [Npad >= n_out, 9*Cin] with k = (ky*3+kx)*Cin + c, or, with ``cblock`` > 0, in the
    block-outer order k = (c // cblock)*9*cblock + (ky*3+kx)*cblock + c % cblock.  Output [B, Ho, Wo, n_out].
    ``tail=(t0, t1 | None)``: a 1x1 conv over cat(t0, t1) (NHWC, the OUTPUT's spatial size) added in the same K loop;
    its [N, Ct0 + Ct1] weight matrix is appended to ``w`` along K (stride 1, no upsampling only).
    ``streams=S``: x is [S*B, H, W, C] (stream-major), ``w`` [S, N, 9*Cin], ``bias`` [S, N]: one grouped launch.
    ``ws``: the same weights as stage images (``wsconv_images``; [S, N*K] with streams): when given and the shape fits
    (``wsconv_ok``) the weight-streaming kernel runs instead of the LDS-tiled one.
    ``gn`` = (gamma, beta, eps, groups, silu): the result is GroupNorm(conv(x)) (-> SiLU): where the conv runs split-K and
    the map is small (``splitk_gn_ok``) the normalisation IS the split-K second pass (ur_igemm_splitk_gn) and the conv
    output is never written; otherwise conv, then ``groupnorm``."""
    Bt, H, W, C0 = x.shape
    B = Bt // streams
    C1 = x1.shape[-1] if x1 is not None else 0
    if ups:
        Ho, Wo = 2 * H, 2 * W
    elif pad == 0:  # the VAE encoder's Downsample2D: F.pad(x, (0, 1, 0, 1)) + conv(stride 2, padding 0)
        if stride != 2:
            raise RuntimeError("conv3x3: pad=0 is the asymmetric (0, 1, 0, 1) padding of the stride-2 VAE downsample")
        Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
    else:
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    N = n_out if n_out is not None else w.shape[-2]
    out = _with_lo(torch.empty(Bt, Ho, Wo, N, dtype=x.dtype, device=x.device), hilo)
    M = B * Ho * Wo
    z = {}
    if streams > 1:
        z = dict(zbatch=streams, zx=B * H * W * C0, zx1=B * H * W * C1, zw=w.stride(0), zout=M * N,
                 zbias=(bias.stride(0) if bias is not None else 0), zres=(M * N if res is not None else 0),
                 zrow=(rowadd.stride(0) * B) if rowadd is not None else 0)
    tl = {}
    if tail is not None:
        ta, tb = tail
        ca, cb_ = ta.shape[-1], (tb.shape[-1] if tb is not None else 0)
        tl = dict(t0=ta, t1=tb, ldt0=ca, ldt1=cb_, ct0=ca, ct1=cb_,
                  zt0=(M * ca if streams > 1 else 0), zt1=(M * cb_ if streams > 1 else 0))
    Kt = 9 * (C0 + C1) + sum(tl.get(k, 0) for k in ("ct0", "ct1"))
    ldw = w.stride(-2)
    if ws is not None and tile in (None, TILE_WS320, TILE_WS320_W8) and wsconv_built() and wsconv_ok(x, N, x1=x1, stride=stride, ups=ups, pad=pad, tail=tail, streams=streams):
        if C0 > WS_C and cblock != WS_C:
            raise RuntimeError("conv3x3: the weight-streaming kernel walks K in the cblock = 320 order")
        w, tile, ldw = ws, (WSCONV_TILE if tile is None else tile), 8
        if streams > 1:
            z["zw"] = ws.stride(0)
        if splitk is None:
            splitk = wsconv_splitk(M, N, Kt, streams)
    gkw = {}
    if gn is not None:
        gam, bet, eps, groups, silu = gn
        if tile is None or splitk is None:
            pt, ps = plan_igemm(M, N, Kt, 9, streams)
            tile, splitk = (pt if tile is None else tile), (ps if splitk is None else splitk)
        if SPLITK_GN and splitk > 1 and res is None and not hilo and splitk_gn_ok(Ho * Wo, N, groups) and min(splitk, Kt // 64) > 1:
            gkw = dict(gn=(gam, bet, (gam.stride(0) if streams > 1 else 0), eps, groups, silu))
    igemm(x0=x, x1=x1, w=w, out=out, M=M, N=N, K=Kt, c0=C0, c1=C1, ldx0=C0, ldx1=C1,
          ldw=ldw, ldc=N, taps=9, conv=(B, H, W, Ho, Wo), stride=stride, ups=int(ups), bias=bias,
          rowadd=rowadd, rows_per_b=Ho * Wo, res=res, ldres=(N if res is not None else 0), out_scale=out_scale,
          tile=tile, splitk=splitk, res_lo=lo_of(res), out_lo=lo_of(out), cblock=cblock, pad=pad, **tl, **z, **gkw)
    if gn is not None and not gkw:
        return groupnorm(out, gam, bet, eps, groups=groups, silu=silu, streams=streams)
    return out


# conv -> GroupNorm (+ SiLU) with the normalisation as the split-K second pass (ur_igemm_splitk_gn).  OFF by default: parity is
# green (tests/test_ops_gpu.py), but the step is 0.07 ms SLOWER with it (85.42 / 85.65 -> 85.01 / 84.92 steps/s alternating
# on one box, profiles/r04_splitk_gn_ab.txt).  Round 4 blamed the 160-byte runs in which one workgroup per (sample, group) reads
# its strip of the row-major slabs; round 6 tried GROUP-BLOCKED slabs (every strip one contiguous run) with four slabs' loads in
# flight -- still 0.05-0.06 ms slower (profiles/r06_splitk_gn_ab.txt: 256 workgroups with two block-wide reductions lose to 4096
# independent reduce blocks + the one-launch GroupNorm), AND the extra code in igemm_kernel's split-K epilogue cost every other
# launch 1.3 % (profiles/r06_slab_binary_ab.txt), so that variant is a patch (tools/patches/r06_group_blocked_slabs.patch), not
# in the tree.  UR_EXPERIMENT=splitk_gn enables the round-4 form.
SPLITK_GN = X.flag("splitk_gn", False)


def splitk_gn_ok(rows: int, N: int, groups: int) -> bool:
    """Shapes ur_igemm_splitk_gn takes: a (sample, group) strip of at most 16384 values, groups of a multiple of 4 channels."""
    return N % groups == 0 and (N // groups) % 4 == 0 and rows * (N // groups) <= 16384


def vt_proj(x, wv, streams=1, shared_x=False):
    """Transposed value projection: Vt[b] = wv @ x[b]^T  -> [B, Cout, Tpad] (columns >= T are zero).
    ``streams=S``: ``wv`` is [S, Cout, Cin] and sample b of stream s uses wv[s]; ``x`` is [S*B, T, Cin], or
    [B, T, Cin] shared by all streams when ``shared_x`` (the prompt embedding); output [S*B, Cout, Tpad]."""
    Bx, T, Cin = x.shape
    B = Bx if shared_x else Bx // streams
    Cout = wv.shape[-2]
    Tpad = (T + 63) // 64 * 64
    out = torch.empty(streams * B, Cout, Tpad, dtype=x.dtype, device=x.device)
    if shared_x and streams > 1:  # z = s * B + b reads x[b]: issue per stream (B problems each)
        for s_ in range(streams):
            igemm(x0=wv[s_], w=x, out=out[s_ * B:(s_ + 1) * B], M=Cout, N=T, K=Cin, c0=Cin, ldx0=wv.stride(-2), ldw=Cin,
                  ldc=Tpad, n_store=Tpad, zbatch=B, zx=0, zw=T * Cin, zout=Cout * Tpad, splitk=1)
        return out
    igemm(x0=wv, w=x, out=out, M=Cout, N=T, K=Cin, c0=Cin, ldx0=wv.stride(-2), ldw=Cin, ldc=Tpad, n_store=Tpad,
          zbatch=streams * B, zx=(wv.stride(0) if streams > 1 else 0), zx_div=B, zw=T * Cin, zout=Cout * Tpad, splitk=1)
    return out


# ---------------------------------------------------------------------------------------------
# norms, attention, glue
# ---------------------------------------------------------------------------------------------
_GN_STAT_KB = X.number("gn_stat_kb", 64)
_GN_APPLY_KB = X.number("gn_apply_kb", 20)
_GN_APPLY_MAX = X.number("gn_apply_max", 128)


def _gn_chunks_bytes(B: int, rows: int, C: int, esize: int = 2):
    """Chunking by BYTES per workgroup (measured on MI355X with tools/bench_ops.py --what gn): a stats workgroup
    streams ~64 KB, an apply workgroup ~20 KB in + 20 KB out; never fewer than 2 rows per chunk, stats partials capped
    at 32 per sample (every apply workgroup re-reduces them in its prologue).  UR_EXPERIMENT entries gn_stat_kb / gn_apply_kb /
    gn_apply_max override the three constants (in-situ A/B runs)."""
    sample_bytes = rows * C * esize
    nstat = max(1, min(sample_bytes // (_GN_STAT_KB << 10), rows // 2, 32))
    napply = max(1, min(sample_bytes // (_GN_APPLY_KB << 10), rows // 2, _GN_APPLY_MAX, max(1, 4096 // max(B, 1))))  # > 128: slower (gn_bench)
    return int(nstat), int(napply)


def resize_nearest(x, size):
    """NHWC nearest-neighbour resize to ``size`` = (H, W) with F.interpolate's index rule."""
    _require_gpu(x)
    lib = _lib.load()
    B, H, W, Cc = x.shape
    out = torch.empty(B, int(size[0]), int(size[1]), Cc, dtype=x.dtype, device=x.device)
    check(lib.ur_resize_nearest(x.data_ptr(), out.data_ptr(), B, H, W, int(size[0]), int(size[1]), Cc, DT[x.dtype], _stream()),
          "ur_resize_nearest")
    return out


# maps of at most this many pixels per sample take the one-launch GroupNorm.  Round 4: with the groups of one sample placed on
# ONE XCD (csrc/norm.hip: neighbouring groups share cache lines) the one-launch kernel also wins at the 32x32
# level (12-15 us against 14-16 for stats + apply; at 64x64 31 against 26: tools/gn_bench.py, profiles/r04_gnf_xcd_ab.txt);
# in the step 256 -> 1024 rows is +0.45 % steps/s (tools/experiments/r04_run11.sh)
GN_FUSED_MAX_ROWS = X.number("gn_fused_max_rows", 1024)
# round 6: small strips of the one-launch GroupNorm stay in registers (one memory round trip); no_gn_resident = always two sweeps
GN_RESIDENT = X.flag("gn_resident", True)
# ... and maps LARGER than GN_FUSED_MAX_ROWS pixels take the one-launch kernel too when their strips still fit in registers (the
# 64x64 level at 320 channels: 20 four-byte pieces per thread) -- one read of the map instead of stats + apply's two
GN_RESIDENT_MAX_ROWS = X.number("gn_resident_max_rows", 1024)


def gn_resident_fits(rows: int, c0: int, c1: int, groups: int, dtype) -> bool:
    """Mirror of csrc/norm.hip's launch_gn_fused: does the register-resident single-sweep kernel take this GroupNorm?"""
    cpg = (c0 + c1) // groups
    P = 8 if cpg % 8 == 0 else 4 if cpg % 4 == 0 else 2 if cpg % 2 == 0 else 0
    if not P or cpg > 128 or not (c1 == 0 or c0 % cpg == 0):
        return False
    per_thread = (rows * (cpg // P) + 1023) // 1024
    most = (8 if dtype == torch.float16 else 4) if P == 8 else 16 if P == 4 else 20
    return 1 <= per_thread <= most


def groupnorm(x, gamma, beta, eps, *, x1=None, groups=32, silu=False, nstat=None, napply=None, streams=1, fused=None,
              return_stats=False, resident=None):
    """GroupNorm over NHWC x (or over cat(x, x1)); returns one contiguous [B,H,W,C0+C1] tensor.
    ``fused``: one launch (ur_groupnorm_fused) instead of stats + apply; default: maps of <= GN_FUSED_MAX_ROWS pixels.
    ``return_stats``: (out, partial statistics of the stats pass | None on the one-launch path) -- the training backward
    reuses them instead of reading x once more."""
    _require_gpu(x)
    lib = _lib.load()
    B = x.shape[0]
    C0 = x.shape[-1]
    C1 = x1.shape[-1] if x1 is not None else 0
    rows = x.numel() // (B * C0)
    if fused is None:
        cpg = (C0 + C1) // groups
        fused = nstat is None and napply is None and cpg % 2 == 0 and cpg <= 128 and (
            rows <= GN_FUSED_MAX_ROWS
            or (GN_RESIDENT and resident is not False and rows <= GN_RESIDENT_MAX_ROWS and gn_resident_fits(rows, C0, C1, groups, x.dtype)))
    if fused:
        out = torch.empty(*x.shape[:-1], C0 + C1, dtype=x.dtype, device=x.device)
        e0 = _prof_begin()
        check(lib.ur_groupnorm_fused(_ptr(x), _ptr(x1), _ptr(lo_of(x)), _ptr(lo_of(x1)), C0, C1, B, rows, groups,
                                     gamma.data_ptr(), beta.data_ptr(), float(eps),
                                     int(bool(silu)) | (0 if (GN_RESIDENT if resident is None else resident) else 2),
                                     (B // streams if streams > 1 else 0), (C0 + C1 if streams > 1 else 0), out.data_ptr(),
                                     DT[x.dtype], _stream()), "ur_groupnorm_fused")
        _prof_end(e0, "gn_fused", 0.0, 2.0 * out.numel() * out.element_size())
        return (out, None) if return_stats else out
    _ns, _na = _gn_chunks_bytes(B, rows, C0 + C1, x.element_size())
    nstat, napply = (nstat or _ns), (napply or _na)
    part = torch.empty(B * nstat * groups * 2, dtype=torch.float32, device=x.device)
    out = torch.empty(*x.shape[:-1], C0 + C1, dtype=x.dtype, device=x.device)
    s = _stream()
    e0 = _prof_begin()
    xl, x1l = lo_of(x), lo_of(x1)
    # The statistics are taken over the hi parts only: the low parts are zero-mean rounding remainders (|lo| <= ulp/2),
    # their contribution to a mean / variance over >= 10^3 elements is ~1e-6 relative -- not worth a second read.
    check(lib.ur_groupnorm_stats(_ptr(x), _ptr(x1), None, None, C0, C1, B, rows, groups, nstat, part.data_ptr(),
                                 DT[x.dtype], s), "ur_groupnorm_stats")
    _prof_end(e0, "gn_stats", 0.0, 1.0 * out.numel() * out.element_size())
    e1 = _prof_begin()
    check(lib.ur_groupnorm_apply(_ptr(x), _ptr(x1), _ptr(xl), _ptr(x1l), C0, C1, B, rows, groups, nstat, napply, part.data_ptr(),
                                 gamma.data_ptr(), beta.data_ptr(), float(eps), int(silu),
                                 (B // streams if streams > 1 else 0), (C0 + C1 if streams > 1 else 0), out.data_ptr(),
                                 DT[x.dtype], s),
          "ur_groupnorm_apply")
    _prof_end(e1, "gn_apply", 0.0, 2.0 * out.numel() * out.element_size())
    return (out, part) if return_stats else out


def layernorm(x, gamma, beta, eps=1e-5, streams=1):
    _require_gpu(x)
    lib = _lib.load()
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    out = torch.empty_like(x)
    e0 = _prof_begin()
    check(lib.ur_layernorm(x.data_ptr(), _ptr(lo_of(x)), gamma.data_ptr(), beta.data_ptr(), float(eps), rows, Cn,
                           (rows // streams if streams > 1 else 0), (Cn if streams > 1 else 0), out.data_ptr(),
                           DT[x.dtype], _stream()), "ur_layernorm")
    _prof_end(e0, "layernorm", 0.0, 2.0 * out.numel() * out.element_size())
    return out


def attention(q, k, vt, *, B, H, Tq, Tk, d, ldq, ldk, q_off=0, k_off=0, scale=None, lse=None, q_hstride=0, k_hstride=0):
    """q/k are token matrices holding head h at columns off + h*d (row strides ldq/ldk, batches contiguous);
    vt is a [B, H*d, Tk_pad] tensor or a row-slice view of a wider batched projection.
    ``q_hstride`` / ``k_hstride`` > 0 (d <= 64): that operand is a HEAD-MAJOR image [B, H, T, d] instead (head stride in
    elements, normally T * d; ldq / ldk still size one sample: H * d) -- what ``tchain.chain_pre(head_major=True)`` writes.
    ``scale=None``: d**-0.5.  ``scale=0``: q.k is already in log2 units (the projections folded scale*log2(e) in),
    which for head dims with a zero-padded k column (d = 40) also selects the kernel without per-score multiply-adds.
    ``lse``: optional contiguous fp32 [B*H, Tq] output, the row log-sum-exp in log2 units (training: the flash backward
    starts from it); needs ``scale`` > 0."""
    _require_gpu(q)
    lib = _lib.load()
    o = torch.empty(B, Tq, H * d, dtype=q.dtype, device=q.device)
    a = AttnDesc()
    a.q, a.k, a.vt, a.o = q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr()
    a.zero_page = zero_page(q.device).data_ptr()
    a.ldq, a.ldk, a.ldvt, a.ldo = ldq, ldk, vt.stride(1), H * d
    a.vt_bstride = vt.stride(0)
    a.q_off, a.k_off = q_off, k_off
    a.q_hstride, a.k_hstride = int(q_hstride), int(k_hstride)
    a.B, a.H, a.Tq, a.Tk, a.d = B, H, Tq, Tk, d
    a.scale = float(scale if scale is not None else d ** -0.5)
    a.dtype = DT[q.dtype]
    if lse is not None:
        if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() < B * H * Tq:
            raise RuntimeError("attention: lse must be a contiguous fp32 [B*H, Tq] tensor")
        a.lse = lse.data_ptr()
    e0 = _prof_begin()
    check(lib.ur_attention(C.byref(a), _stream()), "ur_attention")
    _prof_end(e0, f"attention_d{d}", 4.0 * B * H * Tq * Tk * d, (2.0 * B * Tq + 2.0 * B * Tk) * H * d * q.element_size())
    return o


def ddim_update(pred_nhwc, c0: int, lat_nchw, coef, step, nsteps: int, master=None, round_master: bool = False,
                guidance: Optional[float] = None, cfg_channels: int = 0):
    """In-place DDIM (x0-prediction) update of the NCHW latent slice ``lat_nchw`` [B, C, H, W] (batch stride free,
    channel planes contiguous) from channels c0.. of the NHWC prediction ``pred_nhwc`` [B, H, W, Cp]; the four step
    scalars come from the device table ``coef`` [nsteps, 4] at the device counter ``step`` (include/ur_kernels.h).
    ``master``: optional contiguous fp32 [B, C, H, W] copy that carries the latents between steps.
    ``guidance`` (classifier-free): pred / lat hold 2B samples (cond, uncond), ``master`` B; the first ``cfg_channels``
    channels are guided."""
    _require_gpu(pred_nhwc)
    lib = _lib.load()
    B, Cc, H, W = lat_nchw.shape
    cfg = guidance is not None
    if cfg:
        if B % 2 or pred_nhwc.shape[0] != B:
            raise RuntimeError("ddim_update: guidance needs cond + uncond halves")
        B //= 2
    if lat_nchw.stride(3) != 1 or lat_nchw.stride(2) != W or lat_nchw.stride(1) != H * W:
        raise RuntimeError("ddim_update: latent slice must have contiguous channel planes")
    if pred_nhwc.dtype != lat_nchw.dtype or coef.dtype != torch.float32 or step.dtype != torch.int32:
        raise RuntimeError("ddim_update: dtype mismatch")
    if master is not None and (master.dtype != torch.float32 or not master.is_contiguous()
                               or tuple(master.shape) != (B, Cc, H, W)):
        raise RuntimeError("ddim_update: master must be a contiguous fp32 [B, C, H, W] tensor")
    check(lib.ur_ddim_update(pred_nhwc.data_ptr(), pred_nhwc.shape[-1], c0, lat_nchw.data_ptr(), lat_nchw.stride(0), Cc,
                             B, H * W, coef.data_ptr(), step.data_ptr(), nsteps, _ptr(master), int(round_master),
                             int(cfg), float(guidance or 0.0), int(cfg_channels), DT[lat_nchw.dtype], _stream()),
          "ur_ddim_update")


def unipc_update(pred_nhwc, c0: int, lat_nchw, coef, step, nsteps: int, last, xmaster, hist, round_master: bool = False,
                 guidance: Optional[float] = None, cfg_channels: int = 0):
    """In-place UniPC (order <= 2, x0 prediction, bh2) predictor-corrector update of the NCHW latent slice ``lat_nchw``
    from channels c0.. of the NHWC prediction (include/ur_kernels.h ``ur_unipc_update``); ``coef`` [nsteps, 8] from
    ``schedulers.UniPCMultistepScheduler.coefficient_table``; ``last`` / ``xmaster`` fp32 [B, C, H, W], ``hist`` fp32
    [2, B, C, H, W] (zero before step 0).  ``guidance``: as in ``ddim_update``."""
    _require_gpu(pred_nhwc)
    lib = _lib.load()
    B, Cc, H, W = lat_nchw.shape
    cfg = guidance is not None
    if cfg:
        if B % 2 or pred_nhwc.shape[0] != B:
            raise RuntimeError("unipc_update: guidance needs cond + uncond halves")
        B //= 2
    if lat_nchw.stride(3) != 1 or lat_nchw.stride(2) != W or lat_nchw.stride(1) != H * W:
        raise RuntimeError("unipc_update: latent slice must have contiguous channel planes")
    if pred_nhwc.dtype != lat_nchw.dtype or coef.dtype != torch.float32 or step.dtype != torch.int32 or coef.shape[-1] != 8:
        raise RuntimeError("unipc_update: dtype / table mismatch")
    for t, shp in ((last, (B, Cc, H, W)), (xmaster, (B, Cc, H, W)), (hist, (2, B, Cc, H, W))):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shp:
            raise RuntimeError("unipc_update: state buffers must be contiguous fp32 [B, C, H, W] / [2, B, C, H, W]")
    check(lib.ur_unipc_update(pred_nhwc.data_ptr(), pred_nhwc.shape[-1], c0, lat_nchw.data_ptr(), lat_nchw.stride(0), Cc,
                              B, H * W, coef.data_ptr(), step.data_ptr(), nsteps, last.data_ptr(), xmaster.data_ptr(),
                              hist.data_ptr(), int(round_master), int(cfg), float(guidance or 0.0), int(cfg_channels),
                              DT[lat_nchw.dtype], _stream()), "ur_unipc_update")


def sampler_advance(step, tsteps, nsteps: int, t_out=None):
    lib = _lib.load()
    check(lib.ur_sampler_advance(step.data_ptr(), tsteps.data_ptr(), nsteps, _ptr(t_out),
                                 (t_out.numel() if t_out is not None else 0), _stream()), "ur_sampler_advance")


def select_step_rows(tables, outs, step, nsteps: int):
    """outs[k] = tables[k][*step] for up to four per-step tables ``[nsteps, ...]`` in ONE launch (``ur_select_step_rows``): the
    current step's rows of what a sampling loop computed for all of its steps up front, picked by the device-side step counter
    so that the step stays a pure graph replay."""
    _require_gpu(tables[0])
    import ctypes as C
    lib = _lib.load()
    k = len(tables)
    src = (C.c_void_p * k)(*[t.data_ptr() for t in tables])
    dst = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
    nbytes = (C.c_int64 * k)(*[o.numel() * o.element_size() for o in outs])
    for t, o in zip(tables, outs):
        if t.shape[0] != nsteps or t.numel() != nsteps * o.numel() or t.dtype != o.dtype or not t.is_contiguous() or not o.is_contiguous():
            raise ValueError("select_step_rows: tables are contiguous [nsteps, ...] stacks of the outputs")
    check(lib.ur_select_step_rows(src, dst, nbytes, k, step.data_ptr(), int(nsteps), _stream()), "ur_select_step_rows")
    return outs


def add(a, b, alpha: float = 1.0, hilo=False):
    """a + alpha * b.  Low parts (``a.lo`` / ``b.lo``) are included when present; ``hilo`` also returns ``out.lo``."""
    _require_gpu(a)
    lib = _lib.load()
    out = _with_lo(torch.empty_like(a), hilo)
    al, bl = lo_of(a), lo_of(b)
    e0 = _prof_begin()
    if al is None and bl is None and not hilo:
        check(lib.ur_add(a.data_ptr(), b.data_ptr(), float(alpha), out.data_ptr(), a.numel(), DT[a.dtype], _stream()),
              "ur_add")
    else:
        check(lib.ur_add_hilo(a.data_ptr(), _ptr(al), b.data_ptr(), _ptr(bl), float(alpha), out.data_ptr(),
                              _ptr(lo_of(out)), a.numel(), DT[a.dtype], _stream()), "ur_add_hilo")
    _prof_end(e0, "add", 0.0, 3.0 * out.numel() * out.element_size())
    return out


class _AddItem(C.Structure):
    _fields_ = [("a", C.c_void_p), ("a_lo", C.c_void_p), ("b", C.c_void_p), ("b_lo", C.c_void_p), ("out", C.c_void_p),
                ("out_lo", C.c_void_p), ("n", C.c_int64)]


_lib.register_layout("ur_sizeof_add_item", _AddItem)
ADD_MULTI_MAX = 16


def add_multi(pairs, hilo=False):
    """[a_i + b_i for (a_i, b_i) in pairs] in ONE launch per 16 pairs (``ur_add_hilo_multi``): the low parts of either
    operand are included when present, ``hilo`` also returns ``out.lo``.  All tensors of one dtype, contiguous."""
    pairs = list(pairs)
    if not pairs:
        return []
    _require_gpu(pairs[0][0])
    lib = _lib.load()
    outs = [_with_lo(torch.empty_like(a), hilo) for a, _ in pairs]
    dt = pairs[0][0].dtype
    e0 = _prof_begin()
    for i in range(0, len(pairs), ADD_MULTI_MAX):
        chunk = pairs[i:i + ADD_MULTI_MAX]
        arr = (_AddItem * len(chunk))()
        for k, (a, b) in enumerate(chunk):
            if a.dtype != dt or b.dtype != dt or a.shape != b.shape or not (a.is_contiguous() and b.is_contiguous()):
                raise RuntimeError("add_multi: operands of one pair must be contiguous tensors of one shape and dtype")
            o = outs[i + k]
            arr[k].a, arr[k].a_lo, arr[k].b, arr[k].b_lo = a.data_ptr(), _ptr(lo_of(a)), b.data_ptr(), _ptr(lo_of(b))
            arr[k].out, arr[k].out_lo, arr[k].n = o.data_ptr(), _ptr(lo_of(o)), a.numel()
        check(lib.ur_add_hilo_multi(arr, len(chunk), DT[dt], _stream()), "ur_add_hilo_multi")
    _prof_end(e0, "add_multi", 0.0, sum(3.0 * o.numel() * o.element_size() for o in outs))
    return outs


def timestep_embedding(t: torch.Tensor, B: int, dim: int, flip: bool, shift: float, dtype) -> torch.Tensor:
    _require_gpu(t)
    lib = _lib.load()
    t = t.to(torch.float32).contiguous()
    out = torch.empty(B, dim, dtype=dtype, device=t.device)
    check(lib.ur_timestep_embedding(t.data_ptr(), t.numel(), B, dim, int(flip), float(shift), out.data_ptr(),
                                    DT[dtype], _stream()), "ur_timestep_embedding")
    return out


def to_nhwc(x_nchw: torch.Tensor, dtype, cpad: Optional[int] = None) -> torch.Tensor:
    """[B,C,H,W] (any float dtype, any strides) -> contiguous NHWC [B,H,W,cpad] in ``dtype`` (zero padded)."""
    _require_gpu(x_nchw)
    B, Cc, H, W = x_nchw.shape
    cpad = Cc if cpad is None else cpad
    if cpad == Cc and x_nchw.dtype == dtype and x_nchw.permute(0, 2, 3, 1).is_contiguous():
        v = x_nchw.permute(0, 2, 3, 1)  # already channels-last memory: zero copy
        lo = lo_of(x_nchw)  # a (hi, lo) residual-stream tensor handed back by the caller keeps its low part
        if lo is not None and lo.shape == x_nchw.shape and lo.permute(0, 2, 3, 1).is_contiguous():
            v.lo = lo.permute(0, 2, 3, 1)
        return v
    lib = _lib.load()
    src = x_nchw.contiguous()
    out = torch.empty(B, H, W, cpad, dtype=dtype, device=src.device)
    check(lib.ur_nchw_to_nhwc(src.data_ptr(), DT_ANY[src.dtype], B, Cc, H, W, out.data_ptr(), cpad, DT[dtype],
                              _stream()), "ur_nchw_to_nhwc")
    return out


def to_nchw(x_nhwc: torch.Tensor, dtype=None) -> torch.Tensor:
    """NHWC -> contiguous NCHW copy (dtype conversion fused)."""
    _require_gpu(x_nhwc)
    lib = _lib.load()
    B, H, W, Cc = x_nhwc.shape
    dtype = x_nhwc.dtype if dtype is None else dtype
    out = torch.empty(B, Cc, H, W, dtype=dtype, device=x_nhwc.device)
    check(lib.ur_nhwc_to_nchw(x_nhwc.data_ptr(), DT[x_nhwc.dtype], B, Cc, H, W, out.data_ptr(), DT_ANY[dtype],
                              _stream()), "ur_nhwc_to_nchw")
    return out


def as_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    """Zero-copy logical-NCHW view of an NHWC tensor (channels_last strides); the low part of a (hi, lo) pair stays
    attached (as the same kind of view)."""
    v = x_nhwc.permute(0, 3, 1, 2)
    lo = lo_of(x_nhwc)
    if lo is not None:
        v.lo = lo.permute(0, 3, 1, 2)
    return v
