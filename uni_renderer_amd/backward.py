"""Backward building blocks of the hot path (device op 11) -- the gradients of every leaf op of
the denoise step, each on the MI355X:

  * GEMM-shaped gradients run on the forward implicit-GEMM kernel (``ur_igemm``) over transposed operands
    (``ur_transpose2d``, ``ur_im2col3x3_t``):  dX = dY.W,  dW = dY^T.X,  conv dX = conv3x3(dY, rot180(W)^T),
    conv dW = dY^T . im2col(X);
  * bias / time-embedding gradients are column sums (``ur_colsum``);
  * SiLU, GEGLU, GroupNorm(+SiLU), LayerNorm backward are HIP kernels of their own (csrc/backward.hip).

These are op-level functions with parity tests against autograd (tests/test_backward_gpu.py); the autograd wiring of
the modules and the training step (cfg 4: train/train.py:1258-1427) are not assembled yet.
Reference semantics: ``nn.Linear`` / ``nn.Conv2d`` / ``nn.GroupNorm`` / ``nn.LayerNorm`` / ``F.silu`` / diffusers
``GEGLU`` as used by models/unet_2d_blocks.py:1100-1126.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _experiments as X
from . import _lib, ops
from ._lib import check
from .ops import DT, _ptr, _require_gpu, _stream


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    """[..., R, C] -> [..., C, ceil8(R)] (contiguous; the columns past R are zeros), batch = product of the leading
    dims.  C must be a multiple of 8."""
    _require_gpu(x)
    lib = _lib.load()
    R, Cc = x.shape[-2:]
    Rp = (R + 7) // 8 * 8
    if x.dim() == 3 and x.stride(2) == 1 and x.stride(1) % 8 == 0 and x.stride(0) % 8 == 0 and x.storage_offset() % 8 == 0:
        ld, bs, batch = x.stride(1), x.stride(0), x.shape[0]  # e.g. a column slice of a fused q | k | v projection: no copy
    else:
        x = x.contiguous()
        ld, bs, batch = Cc, R * Cc, x.numel() // (R * Cc)
    out = torch.empty(*x.shape[:-2], Cc, Rp, dtype=x.dtype, device=x.device)
    check(lib.ur_transpose2d(x.data_ptr(), ld, bs, out.data_ptr(), Rp, Rp * Cc, R, Cc, batch, DT[x.dtype], _stream()),
          "ur_transpose2d")
    return out


class _TransposeDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ld_src", C.c_int64), ("bs_src", C.c_int64),
                ("ld_dst", C.c_int64), ("bs_dst", C.c_int64), ("R", C.c_int32), ("C", C.c_int32), ("batch", C.c_int32),
                ("rows_out", C.c_int32), ("colsum", C.c_void_p), ("colsum_ws", C.c_void_p), ("colsum_cnt", C.c_void_p)]


_lib.register_layout("ur_sizeof_transpose_desc", _TransposeDesc)  # checked in _lib.load(), not only in a test

# Bias gradients inside the transpose launch (ur_transpose_desc.colsum): correct and deterministic, removes ~650 launches per
# step, but every transposing workgroup then pays a memory-side store + counter round trip: 84.7 vs 84.6 ms per graphed
# step (tools/experiments/r03_run14.sh) -- no gain, so off by default.
FUSED_COLSUM = X.flag("fused_colsum", False)
# Column sums folded in the same launch by the last-arriving workgroup (ur_colsum_fused; identical bits, 476 launches fewer
# per step): measured 81.4 vs 81.1 ms per step (tools/experiments/r03_run35.sh) -- the agent-scope hand-off costs what the fold launch
# did -- so off by default.
COLSUM_ONE_LAUNCH = X.flag("colsum_one_launch", False)
_colsum_counters: dict = {}


def _colsum_counter(device) -> torch.Tensor:
    """Per-device counter block of the fused column sums (``ur_transpose_desc.colsum_cnt``): zero between launches by
    construction, shared by every call on the (single) training stream."""
    t = _colsum_counters.get(device)
    if t is None:
        t = _colsum_counters[device] = torch.zeros(4096, dtype=torch.int32, device=device)
    return t


MULTI_TRANSPOSE = X.flag("multi_transpose", True)
TRANSPOSE_MAX = 32
# W^T of the Linear weights for dx = dy . W, made in a few multi-tensor launches right after the batched cast of the weights
# (autograd_ops.CastParams) instead of one transpose launch per layer in the backward (426 launches, 3.1 ms per step):
# (data_ptr, shape) of the compute-dtype weight -> its transpose.  UR_EXPERIMENT=no_batch_wt: every linear_backward transposes its own.
BATCH_WT = X.flag("batch_wt", True)
class _DerivedWeights(dict):
    """(data_ptr, shape) of a compute-dtype weight -> a tensor derived from it (its transpose / its rotated form), valid only
    while the buffer the weight lives in is alive: entries carry a weak reference to that buffer, a lookup whose buffer is gone
    is a miss (a later tensor at the same address must not pick up a stale transpose -- ADVICE r4), and ``sweep`` drops the dead
    entries (called when the next network's weights are cast, so a forward that is never backpropagated leaks at most until
    the next differentiable forward)."""

    def put(self, key, owner: torch.Tensor, value: torch.Tensor):
        import weakref
        self[key] = (weakref.ref(owner), value)

    def lookup(self, key):
        ent = dict.get(self, key)
        if ent is None:
            return None
        if ent[0]() is None:
            dict.pop(self, key, None)
            return None
        return ent[1]

    def sweep(self):
        for k in [k for k, (r, _) in self.items() if r() is None]:
            dict.pop(self, k, None)


weight_t = _DerivedWeights()


def transpose2d_many(xs, colsum_of: Optional[int] = None, pad64=()):
    """``[transpose2d(x) for x in xs]`` in one launch per four tensors (``ur_transpose2d_multi``): same dtype, each
    [..., R, C] -> [..., C, ceil8(R)].  ``colsum_of = i``: also the fp32 column sums of the 2-D tensor xs[i], computed by
    the same launch from the tiles it reads anyway (the bias gradient of a linear / conv backward); returns
    (outs, sums).  ``pad64``: indices of the tensors whose transposed row length is zero-padded to a multiple of 64 by the
    launch itself (the contraction dimension of the dW GEMM) instead of by a zero fill + copy afterwards."""
    xs = list(xs)
    if pad64 and not MULTI_TRANSPOSE:
        outs = transpose2d_many(xs, colsum_of=colsum_of)
        o, rest = (outs if colsum_of is None else outs[0]), (None if colsum_of is None else outs[1])
        o = [_pad_rows64(t) if i in pad64 else t for i, t in enumerate(o)]
        return o if colsum_of is None else (o, rest)
    if colsum_of is not None and not (FUSED_COLSUM and MULTI_TRANSPOSE and xs[colsum_of].dim() == 2
                                      and xs[colsum_of].shape[1] <= 64 * 4096):
        return transpose2d_many(xs, pad64=pad64), colsum(xs[colsum_of])
    if colsum_of is None and not pad64 and (not MULTI_TRANSPOSE or len(xs) == 1):
        return [transpose2d(x) for x in xs]
    lib = _lib.load()
    outs, descs = [], []
    for i, x in enumerate(xs):
        _require_gpu(x)
        R, Cc = x.shape[-2:]
        Rp = (R + 63) // 64 * 64 if i in pad64 else (R + 7) // 8 * 8
        if x.dim() == 3 and x.stride(2) == 1 and x.stride(1) % 8 == 0 and x.stride(0) % 8 == 0 and x.storage_offset() % 8 == 0:
            ld, bs, batch = x.stride(1), x.stride(0), x.shape[0]
        else:
            x = x.contiguous()
            ld, bs, batch = Cc, R * Cc, x.numel() // (R * Cc)
        out = torch.empty(*x.shape[:-2], Cc, Rp, dtype=x.dtype, device=x.device)
        outs.append(out)
        descs.append((x, out, ld, bs, Rp, Rp * Cc, R, Cc, batch))
    if len({x.dtype for x in xs}) != 1:
        raise ValueError("transpose2d_many: one dtype per call")
    nmax = TRANSPOSE_MAX
    sums = ws = None
    for i in range(0, len(descs), nmax):
        part = descs[i:i + nmax]
        arr = (_TransposeDesc * len(part))()
        for k, (x, out, ld, bs, ldd, bsd, R, Cc, batch) in enumerate(part):
            arr[k].src, arr[k].dst = x.data_ptr(), out.data_ptr()
            arr[k].ld_src, arr[k].bs_src, arr[k].ld_dst, arr[k].bs_dst = ld, bs, ldd, bsd
            arr[k].R, arr[k].C, arr[k].batch = R, Cc, batch
            arr[k].rows_out = ldd if (i + k) in pad64 else 0
            if colsum_of is not None and i + k == colsum_of:
                sums = torch.empty(Cc, dtype=torch.float32, device=x.device)
                ws = torch.empty((R + 63) // 64 * Cc, dtype=torch.float32, device=x.device)
                arr[k].colsum, arr[k].colsum_ws = sums.data_ptr(), ws.data_ptr()
                arr[k].colsum_cnt = _colsum_counter(x.device).data_ptr()
        check(lib.ur_transpose2d_multi(arr, len(part), DT[xs[0].dtype], _stream()), "ur_transpose2d_multi")
    return outs if colsum_of is None else (outs, sums)


class _CastDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64)]


class GradSquares:
    """Sum of squares of the fp32 parameter gradients, collected where they are WRITTEN (the multi-tensor cast of the weight
    gradients, the conv-weight unpack) instead of by a second sweep over 7 GB at clipping time (train.py:1422).  Every
    contributing launch leaves per-workgroup partial sums; a parameter is *covered* when exactly one launch of the step
    produced its whole gradient.  If any parameter received two contributions (the cycle-consistency branch runs enc + unet
    twice) the partials do not describe the accumulated gradients and the step falls back to ``_foreach_norm``."""

    def __init__(self):
        self.begin(trusted=False)

    def begin(self, token=None, trusted: bool = True):
        """Start collecting for one backward.  ``token`` identifies whose gradients these are (the network triple of the
        step); ``trusted`` is False when the caller knows the backward ADDS to older gradients (accumulation micro-steps,
        a scaled loss, parameters that kept a ``.grad``): the partial sums would then not describe ``p.grad``."""
        self.parts, self.count = [], {}
        self.token, self.trusted = token, bool(trusted)

    def clear(self):
        self.begin(trusted=False)

    def add(self, partial: torch.Tensor, pids):
        self.parts.append(partial)
        for pid in pids:
            self.count[pid] = self.count.get(pid, 0) + 1

    def usable(self, token=None) -> bool:
        return (self.trusted and token == self.token and bool(self.parts)
                and all(c == 1 for c in self.count.values()))


grad_squares = GradSquares()
FUSED_GRADNORM = X.flag("fused_gradnorm", True)


class GradSink:
    """Where the fp32 parameter gradients of a backward are WRITTEN when they live in flat communication buckets
    (parallel.GradientBuckets): the kernels that produce them -- the multi-tensor cast of the Linear weight gradients, the conv
    weight unpack, the bias / norm-parameter sums -- store straight into the parameter's slice of its bucket and hand autograd
    that very view, which ``AccumulateGrad`` adopts as ``p.grad`` without a copy.  Before round 6 every gradient was produced
    in a fresh tensor and then ADDED into the (zeroed) bucket by autograd: 1430 extra launches and ~8 ms per step of cfg 4,
    plus the 7 GB of zeroing.

    ``take(pid)`` returns the destination view for the FIRST contribution to a parameter since ``begin()`` and None afterwards
    (a second contribution -- the cycle-consistency branch runs enc + unet twice, gradient-accumulation micro-steps -- takes the
    ordinary path: a fresh tensor, which autograd then adds in place to the view ``p.grad`` already is)."""

    def __init__(self):
        self.provider, self.written = None, set()

    def begin(self, provider):
        """``provider``: pid -> a NEW fp32 view object over the parameter's gradient storage (or None); None switches off."""
        self.provider, self.written = provider, set()

    def end(self):
        self.provider, self.written = None, set()

    def take(self, pid):
        if self.provider is None or pid in self.written or torch.is_anomaly_enabled():
            return None
        v = self.provider(pid)
        if v is not None:
            self.written.add(pid)
        return v


grad_sink = GradSink()


def cast_many(srcs, dtype, sumsq: bool = False, packed: bool = False, outs=None):
    """``[s.to(dtype) for s in srcs]`` for fp32 -> fp16 / bf16 or fp16 / bf16 -> fp32, 128 tensors per launch
    (``ur_cast_multi``).  ``sumsq`` (to fp32 only): also returns the per-workgroup sums of squares of everything written,
    one 1-D fp32 tensor (``ur_cast_multi_sumsq``).  ``outs``: write into these contiguous tensors (entries may be None:
    allocated here) instead of fresh ones -- the gradient views of a bucket (GradSink)."""
    lib = _lib.load()
    srcs = [s_.contiguous() for s_ in srcs]
    if outs is not None:
        outs = [o if o is not None else torch.empty_like(s_, dtype=dtype) for s_, o in zip(srcs, outs)]
        for s_, o in zip(srcs, outs):
            if o.dtype != dtype or o.numel() != s_.numel() or not o.is_contiguous():
                raise ValueError("cast_many(outs=...): contiguous tensors of the target dtype and the sources' sizes")
    elif packed and srcs:
        # one flat buffer, the copies back to back in the order given: consecutive tensors with the same trailing shape can
        # then be used as ONE matrix without a torch.cat (autograd_ops.cat_adjacent)
        flat = torch.empty(sum(s_.numel() for s_ in srcs), dtype=dtype, device=srcs[0].device)
        outs, off = [], 0
        for s_ in srcs:
            outs.append(flat[off: off + s_.numel()].view(s_.shape))
            off += s_.numel()
    else:
        outs = [torch.empty_like(s_, dtype=dtype) for s_ in srcs]
    if not srcs:
        return outs
    to_f32 = dtype == torch.float32
    low = srcs[0].dtype if to_f32 else dtype
    for s_ in srcs:
        _require_gpu(s_)
        if s_.dtype != (low if to_f32 else torch.float32):
            raise ValueError("cast_many: one source dtype per call (fp32 -> half or half -> fp32)")
    st = _stream()
    partials = []
    for i in range(0, len(srcs), 128):
        part = list(zip(srcs[i:i + 128], outs[i:i + 128]))
        part = [(a, b) for a, b in part if a.numel()]
        if not part:
            continue
        arr = (_CastDesc * len(part))()
        for k, (a, b) in enumerate(part):
            arr[k].src, arr[k].dst, arr[k].n = a.data_ptr(), b.data_ptr(), a.numel()
        if sumsq and to_f32:
            ps = torch.empty(int(lib.ur_cast_multi_blocks(arr, len(part))), dtype=torch.float32, device=srcs[0].device)
            check(lib.ur_cast_multi_sumsq(arr, len(part), 1, DT[low], ps.data_ptr(), st), "ur_cast_multi_sumsq")
            partials.append(ps)
        else:
            check(lib.ur_cast_multi(arr, len(part), int(to_f32), DT[low], st), "ur_cast_multi")
    if sumsq:
        return outs, (torch.cat(partials) if len(partials) > 1 else (partials[0] if partials else None))
    return outs


def colsum(x: torch.Tensor, rows_per_group: int = 0) -> torch.Tensor:
    """fp32 column sums of the 2-D view [M, N] of ``x``; with ``rows_per_group`` one sum per group of rows."""
    _require_gpu(x)
    lib = _lib.load()
    N = x.shape[-1]
    M = x.numel() // N
    groups = 1 if rows_per_group <= 0 else (M + rows_per_group - 1) // rows_per_group
    out = torch.empty(groups, N, dtype=torch.float32, device=x.device)
    dt = 2 if x.dtype == torch.float32 else DT[x.dtype]
    nws = lib.ur_colsum_workspace_floats(M, N, rows_per_group)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    cnt = _colsum_counter(x.device) if (COLSUM_ONE_LAUNCH and nws and lib.ur_colsum_counters(M, N, rows_per_group) <= 4096) else None
    check(lib.ur_colsum_fused(x.data_ptr(), N, M, N, rows_per_group, out.data_ptr(), ws.data_ptr() if nws else None,
                              cnt.data_ptr() if cnt is not None else None, dt, _stream()), "ur_colsum_fused")
    return out if rows_per_group > 0 else out[0]


class _WgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("partial", C.c_void_p),
                ("zero_page", C.c_void_p), ("lddy", C.c_int64), ("ldx", C.c_int64), ("lddw", C.c_int64),
                ("P", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("C", C.c_int32), ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32),
                ("Wout", C.c_int32), ("taps", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("splits", C.c_int32), ("tile", C.c_int32), ("zero_page_bytes", C.c_int32), ("dtype", C.c_int32)]


_lib.register_layout("ur_sizeof_wgrad_desc", _WgradDesc)

# dW (+ db) straight from dy and x as they lie in memory (ur_wgrad: LDS transpose reads) instead of transposed copies +
# the forward GEMM kernel.  UR_EXPERIMENT=no_wgrad restores the round-3 path (A/B: profiles/r04_wgrad_step_ab.txt).
WGRAD = X.flag("wgrad", True)
WGRAD_TILE = X.number("wgrad_tile", 0)
WGRAD_SPLITS = X.number("wgrad_splits", 0)
# measured (tile, slices) per problem "P,N,K,taps,stride" (tools/tune_wgrad.py); anything else takes the library's choice
WGRAD_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wgrad_tuning.json")
_wgrad_table: Optional[dict] = None
WGRAD_TRACE: Optional[dict] = None  # set to a dict to count the problems that pass through wgrad() (the tuner's input)


def wgrad_table() -> dict:
    global _wgrad_table
    if _wgrad_table is None:
        _wgrad_table = {}
        if X.flag("wgrad_table", True) and os.path.exists(WGRAD_TABLE_PATH):
            import json
            with open(WGRAD_TABLE_PATH) as f:
                _wgrad_table = {k: tuple(v) for k, v in json.load(f).items() if not k.startswith("_")}
    return _wgrad_table


def _pow2(v: int) -> bool:
    return v > 0 and (v & (v - 1)) == 0


def wgrad_ok(dy2: torch.Tensor, x: torch.Tensor, conv: Optional[Tuple[int, int]] = None) -> bool:
    """Shapes ``wgrad`` takes (include/ur_kernels.h: ur_wgrad): 16-byte aligned rows, N and K multiples of 8; a conv
    needs C % 64 == 0 and power-of-two output sizes."""
    if dy2.dtype not in DT or x.dtype != dy2.dtype or dy2.dim() != 2 or dy2.stride(1) != 1 or dy2.stride(0) % 8:
        return False
    if dy2.shape[1] % 8 or dy2.data_ptr() % 16 or x.data_ptr() % 16 or x.stride(-1) != 1:
        return False
    # the kernel's loaders form 32-bit byte offsets and pack column offsets into 24 bits (ur::wgrad_check): larger problems
    # must take the transposed-operand fallback here instead of failing with UR_E_UNSUPPORTED at launch (for deferred items
    # that would be at flush time, in the middle of a backward)
    P, N = dy2.shape
    if P * dy2.stride(0) * 2 >= 2 ** 32 or N * 2 >= 2 ** 24:
        return False
    if conv is None:
        if not (x.dim() == 2 and x.shape[1] % 8 == 0 and x.stride(0) % 8 == 0 and x.shape[0] == dy2.shape[0]):
            return False
        return P * x.stride(0) * 2 < 2 ** 32 and x.shape[1] * 2 < 2 ** 24
    Ho, Wo = conv[0], conv[1]
    if not (x.dim() == 4 and x.is_contiguous() and x.shape[3] % 64 == 0 and _pow2(Ho) and _pow2(Wo)):
        return False
    return x.numel() * 2 < 2 ** 32 and x.shape[3] * 2 < 2 ** 24


def wgrad(dy2: torch.Tensor, x: torch.Tensor, need_bias: bool = True, conv: Optional[Tuple[int, int, int]] = None,
          tile: int = 0, splits: int = 0) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dw[n][k] = sum_p dy2[p][n] * xcol[p][k], db[n] = sum_p dy2[p][n] (fp32) -- ``ur_wgrad``.
    dy2 [P, N] (row stride free); x [P, K] for a linear layer, or NHWC [B, H, W, C] with ``conv = (Ho, Wo, stride)``:
    xcol is then the implicit im2col of the 3x3 / pad 1 conv and dw comes out in the packed layout [N][(ky, kx, c)]."""
    _require_gpu(dy2)
    lib = _lib.load()
    P, N = dy2.shape
    d = _WgradDesc()
    if conv is None:
        K = x.shape[1]
        d.taps, d.ldx = 1, x.stride(0)
    else:
        Ho, Wo, stride = conv
        B, H, W, Cc = x.shape
        K = 9 * Cc
        d.taps, d.ldx, d.C, d.B, d.Hin, d.Win, d.Hout, d.Wout, d.stride, d.pad = 9, Cc, Cc, B, H, W, Ho, Wo, stride, 1
    dw = torch.empty(N, K, dtype=dy2.dtype, device=dy2.device)
    db = torch.empty(N, dtype=torch.float32, device=dy2.device) if need_bias else None
    zp = ops.zero_page(dy2.device)
    d.dy, d.x, d.dw, d.db = dy2.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if need_bias else None)
    d.zero_page, d.zero_page_bytes = zp.data_ptr(), ops.ZERO_PAGE_BYTES
    d.lddy, d.lddw, d.P, d.N, d.K = dy2.stride(0), K, P, N, K
    key = f"{P},{N},{K},{d.taps},{d.stride}"
    if WGRAD_TRACE is not None:
        WGRAD_TRACE[key] = WGRAD_TRACE.get(key, 0) + 1
    tuned = wgrad_table().get(key) if not (tile or splits or WGRAD_TILE or WGRAD_SPLITS) else None
    d.tile, d.dtype, d.splits = (tuned[0] if tuned else (tile or WGRAD_TILE)), DT[dy2.dtype], 1
    want = tuned[1] if tuned else (splits or WGRAD_SPLITS)
    if want:
        d.splits = max(1, min(want, (P + 31) // 32))
    else:
        ns, nf = C.c_int32(0), C.c_int64(0)
        check(lib.ur_wgrad_plan(C.byref(d), C.byref(ns), C.byref(nf)), "ur_wgrad_plan")
        d.splits = ns.value
    part = None
    if d.splits > 1:
        part = torch.empty(int(lib.ur_wgrad_partial_floats(C.byref(d))), dtype=torch.float32, device=dy2.device)
        d.partial = part.data_ptr()
    check(lib.ur_wgrad(C.byref(d), _stream()), "ur_wgrad")
    return dw, db


class _WgradPtrs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p)]


WGRAD_GROUP_MAX = 64


class WgradQueue:
    """Weight gradients of Linear layers, DEFERRED: ``add`` hands out uninitialised dw / db tensors and remembers (dy, x);
    ``flush`` computes everything pending, equally shaped problems together in one ``ur_wgrad_group`` launch.  The 16384 x 320 x
    320 projections of a UNet level are 30 us launches of 25 tiles x 16 slices each when run one by one
    (profiles/r04_tune_wgrad.txt); 56 of them in one launch fill the chip without slices, slabs or reduce passes.

    Safe only where nothing reads the tensors before the flush: autograd_ops.Linear uses it for weights that come from
    ``CastParams`` (whose backward -- the first reader -- flushes first) and biases that come from ``ParamBarrier``; a weight that
    appears a second time before a flush (autograd would ADD the two gradients right away) forces the flush."""

    def __init__(self):
        self.items, self.seen, self.keep = [], set(), []
        self.trace: Optional[dict] = None   # set to a dict to count (problem, group size) per flush (tools/tune_wgrad.py)
        # Every pending item keeps its (dy, x) alive until the flush, i.e. until the END of its network's backward: saved
        # inputs are not released layer by layer any more and every layer's output gradient is alive at once.  ``pending``
        # counts those bytes; past ``cap`` the queue flushes early (always safe: a flush only computes what is pending).
        # Default 24 GiB = never reached at cfg 4's per-GPU shape (B = 4: <= 11 GiB pending at the end of the UNet's
        # backward, tools/train_bench.py --mem prints the peaks); UR_WGRAD_PENDING_MB lowers it for larger batches.
        self.pending = 0
        self.peak_pending = 0
        self.early_flushes = 0
        self.cap = int(os.environ.get("UR_WGRAD_PENDING_MB", str(24 << 10))) << 20

    def add(self, dy2: torch.Tensor, x2: torch.Tensor, w_key, need_bias: bool, conv: Optional[Tuple[int, int, int]] = None):
        """``conv`` = (Ho, Wo, stride) with x2 the NHWC input of a 3x3 conv (dw then in the packed layout), else x2 [P, K]."""
        if isinstance(w_key, torch.Tensor):
            # the weight itself: kept alive until the flush so that its ADDRESS -- the identity `seen` goes by -- cannot be handed
            # to another weight in the meantime.  (Under gradient checkpointing the recomputed packed weights are short-lived
            # temporaries; their addresses were reused resnet after resnet, every reuse looked like a repeated weight and
            # flushed the queue early: smaller groups, other (tile, slices) plans, gradients a bf16 ulp off the plain step's.)
            self.keep.append(w_key)
            w_key = w_key.data_ptr()
        if w_key in self.seen or torch.is_anomaly_enabled():  # anomaly detection reads every node's outputs at once
            self.flush()
            return None
        self.seen.add(w_key)
        K = x2.shape[1] if conv is None else 9 * x2.shape[3]
        dw = torch.empty(dy2.shape[1], K, dtype=dy2.dtype, device=dy2.device)
        db = torch.empty(dy2.shape[1], dtype=torch.float32, device=dy2.device) if need_bias else None
        self.items.append((dy2, x2, dw, db, conv))
        self.pending += dy2.numel() * dy2.element_size() + x2.numel() * x2.element_size()
        self.peak_pending = max(self.peak_pending, self.pending)
        if self.pending > self.cap:
            self.early_flushes += 1
            self.flush(keep_seen=True)  # dw / db handed out above are filled now; a repeated weight is still recognised
        return dw, db

    def reset(self):
        """Drop whatever is pending (a backward that raised half-way leaves entries whose gradients nobody will read)."""
        self.items, self.seen, self.pending, self.keep = [], set(), 0, []
        norm_sums.reset()

    def flush(self, keep_seen: bool = False):
        # the deferred gamma / beta gradients ride on the same barriers; an EARLY flush (memory cap) must keep their `seen` set
        # too, or a tied / re-used gamma would be deferred a second time before its ParamBarrier
        norm_sums.flush(keep_seen=keep_seen)
        items, self.items, self.pending = self.items, [], 0
        if not keep_seen:
            self.seen, self.keep = set(), []
        groups: dict = {}
        for it in items:
            dy2, x2, conv = it[0], it[1], it[4]
            groups.setdefault((dy2.shape, x2.shape, dy2.stride(0), x2.stride(0), conv, dy2.dtype, dy2.device), []).append(it[:4])
        for key, its in groups.items():
            for i in range(0, len(its), WGRAD_GROUP_MAX):
                wgrad_group(its[i:i + WGRAD_GROUP_MAX], conv=key[4], trace=self.trace)


class _ColsumItem(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("pair", C.c_int32),
                ("reserved", C.c_int32)]


_lib.register_layout("ur_sizeof_colsum_item", _ColsumItem)
COLSUM_MULTI_MAX = 96


class NormSums:
    """gamma / beta gradients of LayerNorm / GroupNorm layers, DEFERRED like the weight gradients: ``add`` keeps the layer's
    partial sums ([M, N] fp32: per-wave rows of the LayerNorm backward, per-sample (channel, component) pairs of the GroupNorm
    backward) and hands out an uninitialised [N] result; ``flush`` sums up to 96 of them per ``ur_colsum_multi`` launch (one or
    two launches per layer otherwise: ~310 per training step).  Only for parameters behind ``autograd_ops.ParamBarrier``."""

    def __init__(self):
        self.items, self.seen = [], set()

    def fresh(self, gamma: torch.Tensor) -> bool:
        """May this layer's sums be deferred?  Not when the same affine parameters were already deferred since the last flush (a
        norm layer used twice, tied gamma / beta: autograd would ADD the two uninitialised results the moment the second
        arrives -- the repeat takes the immediate path instead), and not under anomaly detection (it inspects every node's
        outputs, which are uninitialised until the flush)."""
        if torch.is_anomaly_enabled():
            return False
        key = gamma.data_ptr()
        if key in self.seen:
            return False
        self.seen.add(key)
        return True

    def add(self, part: torch.Tensor, pair: bool) -> torch.Tensor:
        out = torch.empty(part.shape[1], dtype=torch.float32, device=part.device)
        self.items.append((part, out, pair))
        return out

    def reset(self):
        self.items, self.seen = [], set()

    def flush(self, keep_seen: bool = False):
        items, self.items = self.items, []
        if not keep_seen:
            self.seen = set()
        if not items:
            return
        lib = _lib.load()
        for i in range(0, len(items), COLSUM_MULTI_MAX):
            chunk = items[i:i + COLSUM_MULTI_MAX]
            arr = (_ColsumItem * len(chunk))()
            for k, (part, out, pair) in enumerate(chunk):
                arr[k].inp, arr[k].out, arr[k].M, arr[k].N, arr[k].pair = part.data_ptr(), out.data_ptr(), part.shape[0], part.shape[1], int(pair)
            check(lib.ur_colsum_multi(arr, len(chunk), _stream()), "ur_colsum_multi")


norm_sums = NormSums()
NORM_DEFER = X.flag("norm_defer", True)
wgrad_queue = WgradQueue()
# Deferred + grouped Linear weight gradients (WgradQueue).  UR_EXPERIMENT=no_wgrad_defer: every Linear computes its own at once.
WGRAD_DEFER = X.flag("wgrad_defer", True)


def wgrad_group(items, tile: int = 0, splits: int = 0, trace: Optional[dict] = None, conv: Optional[Tuple[int, int, int]] = None):
    """``ur_wgrad_group`` over ``items`` = [(dy2 [P, N], x, dw [N, K] out, db [N] fp32 out or None)], all of one shape and one
    pair of row strides; x [P, K], or NHWC [B, H, W, C] with ``conv`` = (Ho, Wo, stride) (K = 9 C, packed layout)."""
    lib = _lib.load()
    dy0, x0 = items[0][0], items[0][1]
    _require_gpu(dy0)
    P, N = dy0.shape
    n = len(items)
    d = _WgradDesc()
    zp = ops.zero_page(dy0.device)
    d.zero_page, d.zero_page_bytes = zp.data_ptr(), ops.ZERO_PAGE_BYTES
    if conv is None:
        K = x0.shape[1]
        d.taps, d.ldx = 1, x0.stride(0)
    else:
        Ho, Wo, stride = conv
        Bn, H, W, Cc = x0.shape
        K = 9 * Cc
        d.taps, d.ldx, d.C, d.B, d.Hin, d.Win, d.Hout, d.Wout, d.stride, d.pad = 9, Cc, Cc, Bn, H, W, Ho, Wo, stride, 1
    d.lddy, d.lddw, d.P, d.N, d.K, d.dtype = dy0.stride(0), K, P, N, K, DT[dy0.dtype]
    g = (_WgradPtrs * n)()
    for i, (dy2, x2, dw, db) in enumerate(items):
        g[i].dy, g[i].x, g[i].dw, g[i].db = dy2.data_ptr(), x2.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
    d.dy, d.x, d.dw = g[0].dy, g[0].x, g[0].dw
    key = f"{P},{N},{K},{d.taps},{d.stride}@{n}"
    if trace is not None:
        trace[key] = trace.get(key, 0) + 1
    tuned = wgrad_table().get(key) if not (tile or splits or WGRAD_TILE or WGRAD_SPLITS) else None
    d.tile, d.splits = (tuned[0] if tuned else (tile or WGRAD_TILE)), 1
    want = tuned[1] if tuned else (splits or WGRAD_SPLITS)
    if want:
        d.splits = max(1, min(want, (P + 31) // 32))
    else:
        ns, nf = C.c_int32(0), C.c_int64(0)
        check(lib.ur_wgrad_group_plan(C.byref(d), g, n, C.byref(ns), C.byref(nf)), "ur_wgrad_group_plan")
        d.splits = ns.value
    part = None
    if d.splits > 1:
        part = torch.empty(int(lib.ur_wgrad_partial_floats(C.byref(d))) * n, dtype=torch.float32, device=dy0.device)
        d.partial = part.data_ptr()
    check(lib.ur_wgrad_group(C.byref(d), g, n, _stream()), "ur_wgrad_group")


def _pad_rows64(t: torch.Tensor) -> torch.Tensor:
    """zero-pad the last (contraction) dim of a [R, M] matrix to a multiple of 64 (ur_igemm's K granularity)."""
    M = t.shape[-1]
    if M % 64 == 0:
        return t
    out = torch.zeros(*t.shape[:-1], (M + 63) // 64 * 64, dtype=t.dtype, device=t.device)
    out[..., :M] = t
    return out


def linear_backward(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, need_bias: bool = True, defer: bool = False
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """y = x @ w^T + b  (x [..., K], w [N, K] in the compute dtype, dy [..., N])  ->  (dx, dw, db).
    dx = dy @ w, dw = dy^T @ x (both through ``ur_igemm``), db = column sums of dy (fp32)."""
    K, N = x.shape[-1], w.shape[0]
    x2, dy2 = x.reshape(-1, K), dy.reshape(-1, N)
    if WGRAD and wgrad_ok(dy2, x2):
        wt = weight_t.lookup((w.data_ptr(), tuple(w.shape)))
        dx = ops.linear(dy2, wt if wt is not None else transpose2d(w)).view(x.shape)   # [M, N] @ [K, N]^T
        later = wgrad_queue.add(dy2, x2, w, need_bias) if (defer and WGRAD_DEFER) else None
        dw, db = later if later is not None else wgrad(dy2, x2, need_bias)
        return dx, dw, db
    dy2p = dy2  # transpose2d zero-pads the row count (M = batch rows in the time-embedding GEMMs) to a multiple of 8
    # [K, N], [N, Mp], [K, Mp]: one launch; the last two zero-padded to the 64-granularity of the dW contraction
    if need_bias:
        (wt, dyt, xt), db = transpose2d_many([w, dy2p, x2], colsum_of=1, pad64=(1, 2))
    else:
        (wt, dyt, xt), db = transpose2d_many([w, dy2p, x2], pad64=(1, 2)), None
    dx = ops.linear(dy2, wt).view(x.shape)                # [M, N] @ [K, N]^T
    dw = ops.linear(dyt, xt)                              # [N, M] @ [K, M]^T = [N, K]
    return dx, dw, db


def _rot_weights(w_packed: torch.Tensor, cin: int) -> torch.Tensor:
    """forward conv weights [N][(ky,kx,c)] -> the dgrad conv's [C][(ky',kx',n)] with ky' = 2-ky, kx' = 2-kx, i.e.
    out[c][t' * N + n] = w[n][(8 - t') * C + c]: nine [N][C] -> [C][N] transposes, ONE ``ur_transpose2d`` launch with
    the tap as batch index (source taps walked backwards: negative batch stride)."""
    N = w_packed.shape[0]
    if N % 8 or cin % 8 or not w_packed.is_contiguous() or w_packed.shape[1] != 9 * cin:
        w4 = w_packed.reshape(N, 3, 3, cin)
        return w4.flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * N).contiguous()
    lib = _lib.load()
    out = torch.empty(cin, 9 * N, dtype=w_packed.dtype, device=w_packed.device)
    esz = w_packed.element_size()
    check(lib.ur_transpose2d(w_packed.data_ptr() + 8 * cin * esz, 9 * cin, -cin, out.data_ptr(), 9 * N, N, N, cin, 9,
                             DT[w_packed.dtype], _stream()), "ur_transpose2d")
    return out


weight_rot = _DerivedWeights()  # (data_ptr, shape) of a packed conv weight -> its rotated / channel-transposed form (rot_weights_many)


def rot_weights_many(ws) -> list:
    """``[_rot_weights(w, cin) for w, cin in ws]`` with 32 weights per launch (``ur_transpose2d_multi``; every weight [N, 9 cin]
    contiguous with N and cin multiples of 8): the dgrad weights of all 3x3 convs of a network right after they are packed
    (autograd_ops.PackConvWeights) instead of one launch per conv in the backward."""
    lib = _lib.load()
    outs = [torch.empty(cin, 9 * w.shape[0], dtype=w.dtype, device=w.device) for w, cin in ws]
    for i in range(0, len(ws), TRANSPOSE_MAX):
        part = list(zip(ws[i:i + TRANSPOSE_MAX], outs[i:i + TRANSPOSE_MAX]))
        arr = (_TransposeDesc * len(part))()
        for k, ((w, cin), out) in enumerate(part):
            N = w.shape[0]
            if N % 8 or cin % 8 or not w.is_contiguous() or w.shape[1] != 9 * cin or w.dtype != ws[0][0].dtype:
                raise ValueError("rot_weights_many: packed [N, 9 cin] weights of one dtype with N, cin multiples of 8")
            arr[k].src, arr[k].dst = w.data_ptr() + 8 * cin * w.element_size(), out.data_ptr()
            arr[k].ld_src, arr[k].bs_src, arr[k].ld_dst, arr[k].bs_dst = 9 * cin, -cin, 9 * N, N
            arr[k].R, arr[k].C, arr[k].batch, arr[k].rows_out = N, cin, 9, 0
        check(lib.ur_transpose2d_multi(arr, len(part), DT[ws[0][0].dtype], _stream()), "ur_transpose2d_multi")
    return outs


def resample2x(x: torch.Tensor, mode: int) -> torch.Tensor:
    """NHWC 2x resampling glue (include/ur_kernels.h): 0 nearest upsample, 1 2x2 sum pooling, 2 zero insertion."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    Ho, Wo = (H // 2, W // 2) if mode == 1 else (2 * H, 2 * W)
    out = torch.empty(B, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    check(lib.ur_resample2x(x.contiguous().data_ptr(), out.data_ptr(), B, Ho, Wo, Cc, mode, DT[x.dtype], _stream()),
          "ur_resample2x")
    return out


def _pad_cols64(t: torch.Tensor) -> torch.Tensor:
    """zero-pad the channel (last) dim to a multiple of 64."""
    n = t.shape[-1]
    if n % 64 == 0:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], (n + 63) // 64 * 64, dtype=t.dtype, device=t.device)
    out[..., :n] = t
    return out


def conv3x3_backward(x: torch.Tensor, w_packed: torch.Tensor, dy: torch.Tensor, need_bias: bool = True, stride: int = 1,
                     need_dx: bool = True, defer: bool = False) -> Tuple[Optional[torch.Tensor], torch.Tensor, Optional[torch.Tensor]]:
    """3x3 / pad 1 / stride 1|2 conv over NHWC x [B,H,W,C] (C % 64 == 0) with packed weights [N][(ky,kx,c)],
    dy [B,Ho,Wo,N]  ->  (dx [B,H,W,C], dw [N][(ky,kx,c)], db [N] fp32).
    dx is the same implicit-GEMM conv applied to dy (zero-inserted for stride 2) with the rotated / channel-transposed
    weights; dw contracts dy^T with the transposed im2col of x over the B*Ho*Wo output pixels.  N is zero-padded to a
    multiple of 64 internally (conv_out has 4 / 28 channels)."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    N = w_packed.shape[0]
    Ho, Wo = dy.shape[1:3]
    if Cc % 64:
        raise RuntimeError("conv3x3_backward: input channels must be a multiple of 64")
    dyp = _pad_cols64(dy)                                     # [B,Ho,Wo,Np]
    Np = dyp.shape[-1]
    dx = None
    if need_dx:
        wpad = w_packed if Np == N else torch.cat([w_packed, w_packed.new_zeros(Np - N, w_packed.shape[1])], 0)
        src = dyp if stride == 1 else resample2x(dyp, 2)      # stride 2: zero insertion, then a stride-1 conv
        rot = weight_rot.lookup((w_packed.data_ptr(), tuple(w_packed.shape))) if Np == N else None
        dx = ops.conv3x3(src, rot if rot is not None else _rot_weights(wpad, Cc))
        if dx.shape[1] != H or dx.shape[2] != W:
            raise RuntimeError("conv3x3_backward: odd input sizes are not supported with stride 2")
    P = B * Ho * Wo
    xc = x.contiguous()
    if WGRAD and wgrad_ok(dyp.reshape(P, Np), xc, (Ho, Wo)):
        later = None
        if defer and WGRAD_DEFER and Np == N:
            later = wgrad_queue.add(dyp.reshape(P, Np), xc, w_packed, need_bias, conv=(Ho, Wo, stride))
        if later is not None:
            return dx, later[0], later[1]
        dw, db = wgrad(dyp.reshape(P, Np), xc, need_bias, conv=(Ho, Wo, stride))
        return dx, dw[:N], (db[:N].contiguous() if need_bias else None)
    Pp = (P + 63) // 64 * 64
    xcol_t = torch.empty(9 * Cc, Pp, dtype=x.dtype, device=x.device)
    check(lib.ur_im2col3x3_t(x.contiguous().data_ptr(), B, H, W, Cc, stride, xcol_t.data_ptr(), Pp, DT[x.dtype], _stream()),
          "ur_im2col3x3_t")
    if need_bias:
        (dyt,), db = transpose2d_many([dyp.reshape(P, Np)], colsum_of=0)
        db = db[:N].contiguous()
    else:
        dyt, db = transpose2d(dyp.reshape(P, Np)), None
    dyt = _pad_rows64(dyt)                                    # [Np, Pp]
    dw = ops.linear(dyt, xcol_t)[:N]                          # [Np, Pp] @ [9C, Pp]^T = [Np, 9C]
    return dx, dw, db


def silu_backward(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    dx = torch.empty_like(x)
    check(lib.ur_silu_backward(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), DT[x.dtype], _stream()),
          "ur_silu_backward")
    return dx


def geglu_forward(h: torch.Tensor) -> torch.Tensor:
    """h [..., 2D] = [value | gate] (the reference's chunk(2, -1)) -> value * gelu(gate)."""
    lib = _lib.load()
    D = h.shape[-1] // 2
    M = h.numel() // (2 * D)
    y = torch.empty(*h.shape[:-1], D, dtype=h.dtype, device=h.device)
    check(lib.ur_geglu_forward(h.data_ptr(), y.data_ptr(), M, D, DT[h.dtype], _stream()), "ur_geglu_forward")
    return y


def geglu_backward(h: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    D = h.shape[-1] // 2
    M = h.numel() // (2 * D)
    dh = torch.empty_like(h)
    check(lib.ur_geglu_backward(h.data_ptr(), dy.data_ptr(), dh.data_ptr(), M, D, DT[h.dtype], _stream()),
          "ur_geglu_backward")
    return dh


# GroupNorm backward of maps up to this many pixels per sample in ONE launch (ur_groupnorm_backward_fused) instead of
# statistics + channel partials + fold + dx; 0 disables (the A/B of tools/experiments/r04_run35.sh)
GN_BWD_FUSED_MAX_ROWS = X.number("gn_bwd_fused_max_rows", 1024)


def groupnorm_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                       groups: int = 32, silu: bool = False, stats: Optional[torch.Tensor] = None, defer: bool = False
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x, dy NHWC [B,H,W,C]; y = act(GN(x)*gamma + beta).  -> (dx, dgamma, dbeta) (the last two fp32).
    ``stats``: the forward pass's partial statistics when it kept them; otherwise they are recomputed with
    ``ur_groupnorm_stats`` (one more read of x)."""
    _require_gpu(x)
    lib = _lib.load()
    B, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (B * Cc)
    cpg = Cc // groups
    s = _stream()
    if (rows <= GN_BWD_FUSED_MAX_ROWS and cpg % 2 == 0 and cpg <= 128 and Cc % 8 == 0 and x.is_contiguous() and dy.is_contiguous()
            and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0):
        # small maps: one workgroup per (sample, group) does statistics, channel sums and dx in one launch
        chan_sum = torch.empty(B, Cc, 2, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        check(lib.ur_groupnorm_backward_fused(x.data_ptr(), dy.data_ptr(), Cc, B, rows, groups, gamma.data_ptr(), beta.data_ptr(),
                                              float(eps), int(silu), chan_sum.data_ptr(), dx.data_ptr(), DT[x.dtype], s),
              "ur_groupnorm_backward_fused")
        if defer and NORM_DEFER and norm_sums.fresh(gamma):
            sums = norm_sums.add(chan_sum.view(B, 2 * Cc), True).view(2, Cc)
            return dx, sums[1], sums[0]
        sums = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        check(lib.ur_pairsum_rows(chan_sum.data_ptr(), B, Cc, sums.data_ptr(), s), "ur_pairsum_rows")
        return dx, sums[1], sums[0]
    nstat, nchunks = ops._gn_chunks_bytes(B, rows, Cc, x.element_size())
    if stats is not None and stats.numel() == B * nstat * groups * 2:
        part = stats  # the forward's partial statistics (ops.groupnorm(return_stats=True)): same chunking, x not re-read
    else:
        part = torch.empty(B * nstat * groups * 2, dtype=torch.float32, device=x.device)
        check(lib.ur_groupnorm_stats(x.data_ptr(), None, None, None, Cc, 0, B, rows, groups, nstat, part.data_ptr(), DT[x.dtype],
                                     s), "ur_groupnorm_stats")
    nred = max(1, min(nchunks, 512 // B))  # the reduction pass needs ~512 workgroups, not one per 20 KB
    chan_part = torch.empty(B * nred, Cc, 2, dtype=torch.float32, device=x.device)
    chan_sum = torch.empty(B, Cc, 2, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    check(lib.ur_groupnorm_backward(x.data_ptr(), dy.data_ptr(), Cc, B, rows, groups, nstat, part.data_ptr(),
                                    gamma.data_ptr(), beta.data_ptr(), float(eps), int(silu), nred, chan_part.data_ptr(),
                                    chan_sum.data_ptr(), nchunks, dx.data_ptr(), DT[x.dtype], s), "ur_groupnorm_backward")
    if defer and NORM_DEFER and norm_sums.fresh(gamma):
        sums = norm_sums.add(chan_sum.view(B, 2 * Cc), True).view(2, Cc)
        return dx, sums[1], sums[0]
    sums = torch.empty(2, Cc, dtype=torch.float32, device=x.device)  # rows: sum dz (= dbeta), sum dz * xhat (= dgamma)
    check(lib.ur_pairsum_rows(chan_sum.data_ptr(), B, Cc, sums.data_ptr(), s), "ur_pairsum_rows")
    return dx, sums[1], sums[0]


def layernorm_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-5,
                       skip: Optional[torch.Tensor] = None, defer: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x, dy [..., C] -> (dx, dgamma, dbeta) (fp32 parameter gradients).  ``skip``: a gradient reaching x around the norm
    (same shape), added to dx by the kernel."""
    _require_gpu(x)
    lib = _lib.load()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    rpw = max(1, min(64, rows // 2048))  # rows per wave: enough waves to fill the chip, few partial rows
    waves = (rows + rpw - 1) // rpw
    part = torch.empty(waves, 2, Cc, dtype=torch.float32, device=x.device)  # every wave writes its whole row
    dx = torch.empty_like(x)
    check(lib.ur_layernorm_backward_skip(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), float(eps), rows, Cc, rpw, dx.data_ptr(),
                                         part.data_ptr(), skip.data_ptr() if skip is not None else None, DT[x.dtype], _stream()),
          "ur_layernorm_backward_skip")
    if defer and NORM_DEFER and norm_sums.fresh(gamma):
        sums = norm_sums.add(part.view(waves, 2 * Cc), False).view(2, Cc)
        return dx, sums[0], sums[1]
    sums = colsum(part.view(waves, 2 * Cc)).view(2, Cc)
    return dx, sums[0].contiguous(), sums[1].contiguous()


def _split_heads(x: torch.Tensor, H: int, d: int, Tp: int, dp: int, off: int = 0) -> torch.Tensor:
    lib = _lib.load()
    B, T = x.shape[0], x.shape[1]
    out = torch.empty(B * H, Tp, dp, dtype=x.dtype, device=x.device)
    check(lib.ur_split_heads(x.data_ptr(), x.stride(1), off, B, T, H, d, out.data_ptr(), Tp, dp, DT[x.dtype], _stream()),
          "ur_split_heads")
    return out


def _merge_heads(g: torch.Tensor, B: int, T: int, H: int, d: int, out: Optional[torch.Tensor] = None, off: int = 0
                 ) -> torch.Tensor:
    """[B*H, Tp, dp] -> columns off .. off + H*d of ``out`` [B, T, ld] (a new [B, T, H*d] tensor by default)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(B, T, H * d, dtype=g.dtype, device=g.device)
    check(lib.ur_merge_heads(g.data_ptr(), g.shape[1], g.shape[2], B, T, H, d, out.data_ptr(), out.stride(1), off,
                             DT[g.dtype], _stream()), "ur_merge_heads")
    return out


class _HeadsDesc(C.Structure):
    _fields_ = [("tok", C.c_void_p), ("heads", C.c_void_p), ("ld", C.c_int64), ("off", C.c_int32), ("T", C.c_int32),
                ("Tp", C.c_int32), ("reserved", C.c_int32)]


_lib.register_layout("ur_sizeof_heads_desc", _HeadsDesc)
# the per-head copies of the d = 40 flash backward (q, k, v, o, dO in; dq, dk, dv out) in one launch each way instead of 5 + 3
HEADS_MULTI = X.flag("heads_multi", True)


def split_heads_many(items, H: int, d: int, dp: int):
    """``[_split_heads(x, H, d, Tp, dp, off) for (x, Tp, off) in items]`` in one launch (``ur_split_heads_multi``)."""
    lib = _lib.load()
    B = items[0][0].shape[0]
    outs = [torch.empty(B * H, Tp, dp, dtype=x.dtype, device=x.device) for x, Tp, _ in items]
    arr = (_HeadsDesc * len(items))()
    for i, ((x, Tp, off), o) in enumerate(zip(items, outs)):
        if x.shape[0] != B or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
            raise RuntimeError("split_heads_many: [B, T, ld] operands with contiguous batches")
        arr[i].tok, arr[i].heads, arr[i].ld, arr[i].off, arr[i].T, arr[i].Tp = x.data_ptr(), o.data_ptr(), x.stride(1), off, x.shape[1], Tp
    check(lib.ur_split_heads_multi(arr, len(items), B, H, d, dp, DT[items[0][0].dtype], _stream()), "ur_split_heads_multi")
    return outs


def merge_heads_many(items, B: int, H: int, d: int):
    """items = [(g [B*H, Tp, dp], out [B, T, ld], off)]: columns off .. off + H*d of every ``out`` in one launch."""
    lib = _lib.load()
    arr = (_HeadsDesc * len(items))()
    for i, (g, out, off) in enumerate(items):
        arr[i].tok, arr[i].heads, arr[i].ld, arr[i].off, arr[i].T, arr[i].Tp = out.data_ptr(), g.data_ptr(), out.stride(1), off, out.shape[1], g.shape[1]
    check(lib.ur_merge_heads_multi(arr, len(items), B, H, d, items[0][0].shape[2], DT[items[0][0].dtype], _stream()), "ur_merge_heads_multi")


FLASH_BACKWARD = X.flag("flash_backward", True)  # 0: always the materialised-P path below
FORWARD_LSE = X.flag("forward_lse", True)        # 0: the dq kernel recomputes the row log-sum-exp


def flash_stats(B: int, H: int, Tq: int, Tk: int, d: int, device) -> Optional[torch.Tensor]:
    """The [2, B*H, Tq] fp32 statistics buffer of the flash backward if it will handle this shape (the forward kernel
    writes the row log-sum-exp into its first half: ``ops.attention(lse=stats[0])``), else None."""
    if not (FLASH_BACKWARD and FORWARD_LSE and _lib.load().ur_attention_backward_supported(Tq, Tk, d)):
        return None
    return torch.empty(2, B * H, Tq, dtype=torch.float32, device=device)


FLASH_DIRECT_MIN_D = X.number("flash_direct_min_d", 64)


def _flash_attention_backward(q, k, v, o, do, H, scale, fused_qkv, oq, ok, ov, Cc, d, stats=None):
    """``ur_attention_backward`` (csrc/attention_bwd.hip) -- P stays in registers; self-attention and the 77-key
    cross-attention.  Head dims >= 64 (d = 80 / 160): the kernels read q / k / v / o / dO and write dq / dk / dv in the
    [B, T, H*d] layouts (column offsets for the parts of a fused projection); the only prepared operands are the
    transposes of q, k and dO.  d = 40: per-head copies [B*H, T, 64] first (``ur_split_heads`` / ``ur_merge_heads``) --
    the same kernels with H = 1 and no bounds predicates -- which measured 0.4-0.5 ms per training step faster than
    reading the 80-byte head rows in place."""
    lib = _lib.load()
    B, Tq = q.shape[:2]
    Tk = k.shape[1]
    Tkp = (Tk + 63) // 64 * 64
    S, dp = B * H, (d + 31) // 32 * 32
    esz = q.element_size()
    direct = d >= FLASH_DIRECT_MIN_D
    # the transposed copies q^T, k^T, dO^T: only a library built with -DUR_ATTN_BWD_TRN=1 still reads them (ABI 9: the kernels
    # gather the transposed fragments from the row-major tiles with the LDS transpose read)
    need_t = bool(lib.ur_attention_backward_needs_transposes())
    has_lse = stats is not None and tuple(stats.shape) == (2, S, Tq)
    if not has_lse:
        stats = torch.empty(2, S, Tq, dtype=torch.float32, device=q.device)
    G = lib.ur_attention_backward_splits(S, Tq, Tkp, dp)
    part = torch.empty(2 * G * S * Tkp * dp, dtype=torch.float32, device=q.device) if G > 1 else None
    a = _lib.AttnBwdDesc()
    if direct:
        for t in (q, k, v, o, do):
            if t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
                raise RuntimeError("flash attention backward: [B, T, ld] operands with contiguous batches")
        qt = kt = dot_ = None
        if need_t:  # ABI <= 8 libraries: [B, C, T], one launch; the key side zero-padded to the 64-key tiles by the launch itself
            qt, kt, dot_ = transpose2d_many([q[..., oq:oq + Cc], k[..., ok:ok + Cc], do], pad64=((1,) if Tk % 64 else ()))
            if kt.shape[-1] != Tkp:
                kt = _pad_rows64(kt)
        if fused_qkv:
            g = torch.empty(B, Tq, 3 * Cc, dtype=q.dtype, device=q.device)
            outs, ldg, offs = (g, g, g), 3 * Cc, (oq, ok, ov)
        else:
            outs = (torch.empty(B, Tq, Cc, dtype=q.dtype, device=q.device), torch.empty(B, Tk, Cc, dtype=q.dtype, device=q.device),
                    torch.empty(B, Tk, Cc, dtype=q.dtype, device=q.device))
            ldg, offs = Cc, (0, 0, 0)
        a.q, a.k, a.v = q.data_ptr() + oq * esz, k.data_ptr() + ok * esz, v.data_ptr() + ov * esz
        a.o, a.dout = o.data_ptr(), do.data_ptr()
        a.ldq, a.ldk, a.ldv, a.ldo, a.lddo = q.stride(1), k.stride(1), v.stride(1), o.stride(1), do.stride(1)
        a.dq, a.dk, a.dv = (t.data_ptr() + off * esz for t, off in zip(outs, offs))
        a.lddq = a.lddk = a.lddv = ldg
        a.B, a.H, a.d, a.Tk_rows = B, H, d, Tk
    else:
        if HEADS_MULTI and all(t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) for t in (q, k, v, o, do)):
            qp, kp, vp, op, dop = split_heads_many([(q, Tq, oq), (k, Tkp, ok), (v, Tkp, ov), (o, Tq, 0), (do, Tq, 0)], H, d, dp)
        else:
            qp, kp, vp = _split_heads(q, H, d, Tq, dp, oq), _split_heads(k, H, d, Tkp, dp, ok), _split_heads(v, H, d, Tkp, dp, ov)
            op, dop = _split_heads(o, H, d, Tq, dp), _split_heads(do, H, d, Tq, dp)       # [S, T, dp]
        qt, kt, dot_ = transpose2d_many([qp, kp, dop]) if need_t else (None, None, None)  # [S, dp, T], one launch
        dQ, dK, dV = torch.empty_like(qp), torch.empty_like(kp), torch.empty_like(kp)
        a.q, a.k, a.v, a.o, a.dout = qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), op.data_ptr(), dop.data_ptr()
        a.ldq = a.ldk = a.ldv = a.ldo = a.lddo = a.lddq = a.lddk = a.lddv = dp
        a.dq, a.dk, a.dv = dQ.data_ptr(), dK.data_ptr(), dV.data_ptr()
        a.B, a.H, a.d, a.Tk_rows = S, 1, dp, Tkp
    if need_t:
        a.qt, a.kt, a.dot = qt.data_ptr(), kt.data_ptr(), dot_.data_ptr()
        a.ldqt, a.ldkt, a.lddot = qt.shape[-1], kt.shape[-1], dot_.shape[-1]
    a.stats = stats.data_ptr()
    a.part = part.data_ptr() if part is not None else None
    a.Tq, a.Tk, a.has_lse = Tq, Tk, int(has_lse)
    a.scale, a.dtype = scale, DT[q.dtype]
    check(lib.ur_attention_backward(C.byref(a), _stream()), "ur_attention_backward")
    if direct:
        return outs[0] if fused_qkv else outs
    if fused_qkv:
        g = torch.empty(B, Tq, 3 * Cc, dtype=q.dtype, device=q.device)
        if HEADS_MULTI:
            merge_heads_many([(dQ, g, oq), (dK, g, ok), (dV, g, ov)], B, H, d)
            return g
        for part_, off in ((dQ, oq), (dK, ok), (dV, ov)):
            _merge_heads(part_, B, Tq, H, d, out=g, off=off)
        return g
    if HEADS_MULTI:
        outs = (torch.empty(B, Tq, H * d, dtype=q.dtype, device=q.device), torch.empty(B, Tk, H * d, dtype=q.dtype, device=q.device),
                torch.empty(B, Tk, H * d, dtype=q.dtype, device=q.device))
        merge_heads_many([(dQ, outs[0], 0), (dK, outs[1], 0), (dV, outs[2], 0)], B, H, d)
        return outs
    return _merge_heads(dQ, B, Tq, H, d), _merge_heads(dK, B, Tk, H, d), _merge_heads(dV, B, Tk, H, d)


def attention_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, do: torch.Tensor, H: int,
                       scale: Optional[float] = None, fused_qkv: bool = False, o: Optional[torch.Tensor] = None,
                       stats: Optional[torch.Tensor] = None):
    """Gradients of o = softmax(q k^T * scale) v per head (q, do [B,Tq,H*d]; k, v [B,Tk,H*d]) -> (dq, dk, dv).
    The forward keeps nothing but q, k, v (flash kernel); here P is recomputed and materialised per (batch, head)
    ([B*H, Tq, Tk] in the compute dtype) and the five GEMMs run z-batched on ``ur_igemm``:
        S = Q K^T,  dV = P^T dO,  dP = dO V^T,  dS = P (dP - rowsum(dP P)) scale,  dQ = dS K,  dK = dS^T Q."""
    lib = _lib.load()
    B, Tq = q.shape[:2]
    Tk = k.shape[1]
    # fused_qkv: q is k is v = the [B, T, 3C] output of one q | k | v projection; heads are read at column offsets
    # 0 / C / 2C and the result is ONE [B, T, 3C] gradient (no slice copies, no concatenation)
    Cc = q.shape[2] // 3 if fused_qkv else q.shape[2]
    oq, ok, ov = (0, Cc, 2 * Cc) if fused_qkv else (0, 0, 0)
    d = Cc // H
    scale = float(d ** -0.5 if scale is None else scale)
    if FLASH_BACKWARD and o is not None and lib.ur_attention_backward_supported(Tq, Tk, d):
        return _flash_attention_backward(q, k, v, o, do, H, scale, fused_qkv, oq, ok, ov, Cc, d, stats)
    dp = (d + 63) // 64 * 64
    Tqp, Tkp = (Tq + 63) // 64 * 64, (Tk + 63) // 64 * 64
    S = B * H
    qp, dop = _split_heads(q, H, d, Tqp, dp, oq), _split_heads(do, H, d, Tqp, dp)   # [S, Tqp, dp]
    kp, vp = _split_heads(k, H, d, Tkp, dp, ok), _split_heads(v, H, d, Tkp, dp, ov) # [S, Tkp, dp]
    s_ = _stream()
    # P = softmax(Q K^T * scale): rows = queries (padded rows are all-zero q -> uniform rows, masked out below by dO = 0)
    P = ops.linear(qp, kp, out_scale=scale, streams=S)                              # [S*Tqp, Tkp] viewed [S, Tqp, Tkp]
    P = P.view(S, Tqp, Tkp)
    check(lib.ur_softmax_rows(P.data_ptr(), Tkp, S * Tqp, Tk, DT[P.dtype], s_), "ur_softmax_rows")
    dP = ops.linear(dop, vp, streams=S).view(S, Tqp, Tkp)                           # dO V^T
    Pt, dot_ = transpose2d(P), transpose2d(dop)                                     # [S, Tkp, Tqp], [S, dp, Tqp]
    dV = ops.linear(Pt.view(S * Tkp, Tqp), dot_, streams=S).view(S, Tkp, dp)        # P^T dO
    check(lib.ur_softmax_backward_rows(P.data_ptr(), dP.data_ptr(), Tkp, S * Tqp, Tk, scale, DT[P.dtype], s_),
          "ur_softmax_backward_rows")                                                 # dP now holds dS
    dS = dP
    kpt, qpt = transpose2d(kp), transpose2d(qp)                                     # [S, dp, Tkp], [S, dp, Tqp]
    dQ = ops.linear(dS.view(S * Tqp, Tkp), kpt, streams=S).view(S, Tqp, dp)         # dS K
    dSt = transpose2d(dS)                                                           # [S, Tkp, Tqp]
    dK = ops.linear(dSt.view(S * Tkp, Tqp), qpt, streams=S).view(S, Tkp, dp)        # dS^T Q
    if fused_qkv:
        g = torch.empty(B, Tq, 3 * Cc, dtype=q.dtype, device=q.device)
        for part, off in ((dQ, oq), (dK, ok), (dV, ov)):
            _merge_heads(part, B, Tq, H, d, out=g, off=off)
        return g
    return _merge_heads(dQ, B, Tq, H, d), _merge_heads(dK, B, Tk, H, d), _merge_heads(dV, B, Tk, H, d)
