"""torch.autograd Functions over the HIP ops: forward = the forward kernels of ``ops``, backward = the building blocks
of ``backward`` (device op 11).  They let the training step (train_step.py; reference
train/train.py:1258-1427) be written as an ordinary forward pass -- autograd only does the graph bookkeeping (skip
connections, concatenations, dtype casts of the fp32 master parameters); every gradient is computed by this package's
kernels.  All tensors are NHWC / token matrices in the compute dtype (bf16 or fp16); biases and norm parameters fp32.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib, backward as bw, ops
from ._lib import check
from .ops import DT, _stream


# Deferred weight gradients (backward.WgradQueue): a Linear may hand autograd UNINITIALISED dw / db tensors only if their first
# reader flushes the queue first.  That is the case for weights that are (views of) a CastParams output and biases that are a
# ParamBarrier output; both register here, by storage / data pointer, with a weak reference that dies with the autograd graph.
_deferred_w: dict = {}   # storage data_ptr of a CastParams buffer -> weakref(buffer)
_deferred_b: dict = {}   # data_ptr of a ParamBarrier output      -> weakref(output)
deferred_bias: dict = {}  # id(fp32 bias parameter) -> its ParamBarrier output for the current network forward (train_step)


def _alive(table: dict, ptr: int) -> bool:
    ref = table.get(ptr)
    if ref is None:
        return False
    t = ref()
    if t is None:
        del table[ptr]
        return False
    return True


def _can_defer(w, b) -> bool:
    if not bw.WGRAD_DEFER or not bw.WGRAD or not _alive(_deferred_w, w.untyped_storage().data_ptr()):
        return False
    if b is None:
        return True
    ref = _deferred_b.get(b.data_ptr())
    return ref is not None and ref() is b   # the barrier's output OBJECT, not merely a tensor at the parameter's address


class Linear(Function):
    """y = x @ w^T (+ b) (+ rowadd[m // rows_per_b]) (+ res).  x [..., K], w [N, K], b fp32 [N], rowadd [B, N]."""

    @staticmethod
    def forward(ctx, x, w, b, res, rowadd, rows_per_b):
        ctx.save_for_backward(x, w)
        ctx.flags = (b is not None, res is not None, rowadd is not None, int(rows_per_b))
        ctx.defer = _can_defer(w, b)
        return ops.linear(x, w, b, res=res, rowadd=rowadd, rows_per_b=rows_per_b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        has_b, has_res, has_row, rpb = ctx.flags
        dy = dy.contiguous()
        dx, dw, db = bw.linear_backward(x, w, dy, need_bias=has_b, defer=ctx.defer)
        drow = bw.colsum(dy, rows_per_group=rpb).to(dy.dtype) if has_row else None
        return dx, dw, db, (dy if has_res else None), drow, None


class ParamBarrier(Function):
    """Identity over many fp32 parameters (the Linear biases of a network): the outputs are new tensor objects over the
    parameters' own storage; the backward runs when the LAST of their gradients has arrived, flushes the deferred weight
    gradient queue (which fills the bias gradients handed out uninitialised) and passes them on."""

    @staticmethod
    def forward(ctx, *params):
        import weakref
        ctx.set_materialize_grads(False)
        ctx.pids = [id(p) for p in params]
        outs = tuple(torch.empty(0, dtype=p.dtype, device=p.device).set_(p.untyped_storage(), p.storage_offset(), p.shape, p.stride())
                     for p in params)
        for o in outs:
            _deferred_b[o.data_ptr()] = weakref.ref(o)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        bw.wgrad_queue.flush()
        if bw.grad_sink.provider is None:
            return grads
        # gradients living in communication buckets: ONE multi-tensor copy puts the (~1000 small) bias / norm gradients of
        # the network into their bucket slices; autograd receives those views and adopts them as p.grad (no add launches)
        res, src, dst = list(grads), [], []
        for i, (g, pid) in enumerate(zip(grads, ctx.pids)):
            if g is None or g.dtype != torch.float32:
                continue
            v = bw.grad_sink.take(pid)
            if v is not None and v.shape == g.shape:
                src.append(g)
                dst.append(v)
                res[i] = v
        if dst:
            torch._foreach_copy_(dst, src)
        return tuple(res)


class Conv3x3(Function):
    """3x3 / pad 1 conv over NHWC x with packed weights [N][(ky,kx,c)] (+ b) (+ rowadd per sample) (+ res)."""

    @staticmethod
    def forward(ctx, x, wp, b, res, rowadd, stride):
        ctx.save_for_backward(x, wp)
        ctx.flags = (b is not None, res is not None, rowadd is not None, int(stride))
        ctx.defer = _can_defer(wp, b)
        return ops.conv3x3(x, wp, b, stride=stride, rowadd=rowadd, res=res)

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        has_b, has_res, has_row, stride = ctx.flags
        dy = dy.contiguous()
        dx, dw, db = bw.conv3x3_backward(x, wp, dy, need_bias=has_b, stride=stride, need_dx=ctx.needs_input_grad[0],
                                         defer=ctx.defer)
        drow = None
        if has_row:
            drow = bw.colsum(bw._pad_cols64(dy), rows_per_group=dy.shape[1] * dy.shape[2])[:, : dy.shape[-1]].to(dy.dtype)
        return dx, dw, db, (dy if has_res else None), drow, None


class Up2x(Function):
    """nearest 2x upsample (F.interpolate of Upsample2D, unet_2d_blocks.py:2501); backward = 2x2 sum pooling."""

    @staticmethod
    def forward(ctx, x):
        return bw.resample2x(x, 0)

    @staticmethod
    def backward(ctx, dy):
        return bw.resample2x(dy.contiguous(), 1)


def _is_barrier(t) -> bool:
    """``t`` is the very output object of a live ParamBarrier (its gradient may be handed out uninitialised)."""
    ref = _deferred_b.get(t.data_ptr())
    return ref is not None and ref() is t


def _norm_params(gamma, beta):
    """the ParamBarrier outputs of a norm layer's parameters inside a batched-cast context (train_step), else themselves"""
    if deferred_bias:
        return deferred_bias.get(id(gamma), gamma), deferred_bias.get(id(beta), beta)
    return gamma, beta


def group_norm(x, gamma, beta, eps, groups, silu):
    gamma, beta = _norm_params(gamma, beta)
    return GroupNorm.apply(x, gamma, beta, eps, groups, silu)


def layer_norm_skip(x, gamma, beta, eps):
    gamma, beta = _norm_params(gamma, beta)
    return LayerNormSkip.apply(x, gamma, beta, eps)


class GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, groups, silu):
        ctx.save_for_backward(x, gamma, beta)
        ctx.defer = bw.NORM_DEFER and bw.WGRAD_DEFER and _is_barrier(gamma) and _is_barrier(beta)
        ctx.cfg = (float(eps), int(groups), bool(silu))
        y, ctx.stats = ops.groupnorm(x, gamma, beta, eps, groups=groups, silu=silu, return_stats=True)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        eps, groups, silu = ctx.cfg
        dx, dg, db = bw.groupnorm_backward(x, dy.contiguous(), gamma, beta, eps, groups=groups, silu=silu, stats=ctx.stats,
                                            defer=ctx.defer)
        return dx, dg, db, None, None, None


class LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = float(eps)
        return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = bw.layernorm_backward(x, dy.contiguous(), gamma, ctx.eps)
        return dx, dg, db, None


class LayerNormSkip(Function):
    """(x, LayerNorm(x)): the pre-norm residual pattern ``x + f(LN(x))`` as ONE autograd node, so that the gradient reaching x
    around the norm and the norm's own input gradient are added by the LayerNorm backward kernel
    (``ur_layernorm_backward_skip``) instead of by a separate accumulation launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = float(eps)
        ctx.defer = bw.NORM_DEFER and bw.WGRAD_DEFER and _is_barrier(gamma) and _is_barrier(beta)
        ctx.set_materialize_grads(False)
        return x.view_as(x), ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dskip, dy):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None
        skip = dskip.contiguous() if dskip is not None else None
        dx, dg, db = bw.layernorm_backward(x, dy.contiguous(), gamma, ctx.eps, skip=skip, defer=ctx.defer)
        return dx, dg, db, None


class Attention(Function):
    """softmax(q k^T / sqrt(d)) v per head; q [B,Tq,H*d], k / v [B,Tk,H*d].  Forward: the flash kernel (nothing but
    q, k, v is kept); backward: backward.attention_backward."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        B, Tq, Cc = q.shape
        Tk = k.shape[1]
        d = Cc // heads
        # column slices of a wider projection (the [B, 77, 2C] k | v of a cross-attention) are read in place: the kernels
        # take a leading dimension, transpose2d a strided source
        inplace = lambda t: (t.stride(-1) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0
                             and t.storage_offset() % 8 == 0)
        k_ = k if inplace(k) else k.contiguous()
        vt = bw.transpose2d_many([v if inplace(v) else v.contiguous()], pad64=(0,))[0]  # [B, C, Tk_pad]: padded by the launch
        stats = bw.flash_stats(B, heads, Tq, Tk, d, q.device)  # [2, B*H, T] when the flash backward will run, else None
        o = ops.attention(q.contiguous(), k_, vt, B=B, H=heads, Tq=Tq, Tk=Tk, d=d, ldq=Cc, ldk=k_.stride(1),
                          lse=None if stats is None else stats[0])
        ctx.save_for_backward(q, k, v, o)  # o: rowsum(do * o) of the flash backward (the out projection keeps it anyway)
        ctx.heads, ctx.stats = heads, stats
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        # column slices of a wider projection (the [B, 77, 2C] k | v of a cross-attention) are read in place
        q, k, v = (t if t.stride(-1) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0 else t.contiguous()
                   for t in (q, k, v))
        dq, dk, dv = bw.attention_backward(q, k, v, do.contiguous(), ctx.heads, o=o, stats=ctx.stats)
        return dq, dk, dv, None


class AttentionQKV(Function):
    """Self-attention over the [B, T, 3C] output of ONE q | k | v projection: the kernels read the three parts at their
    column offsets (leading dimension 3C) and the backward returns one [B, T, 3C] gradient."""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = qkv.contiguous()
        B, T, C3 = qkv.shape
        Cc = C3 // 3
        vt = bw.transpose2d_many([qkv[..., 2 * Cc:]], pad64=(0,))[0]  # [B, C, T_pad], read in place (strided), padded by the launch
        stats = bw.flash_stats(B, heads, T, T, Cc // heads, qkv.device)
        o = ops.attention(qkv, qkv, vt, B=B, H=heads, Tq=T, Tk=T, d=Cc // heads, ldq=C3, ldk=C3, q_off=0, k_off=Cc,
                          lse=None if stats is None else stats[0])
        ctx.save_for_backward(qkv, o)
        ctx.heads, ctx.stats = heads, stats
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o = ctx.saved_tensors
        return bw.attention_backward(qkv, qkv, qkv, do.contiguous(), ctx.heads, fused_qkv=True, o=o, stats=ctx.stats), None


class GEGLU(Function):
    """h = [value | gate] -> value * gelu(gate) (diffusers GEGLU)."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return bw.geglu_forward(h.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return bw.geglu_backward(h.contiguous(), dy.contiguous())


class SiLU(Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        ctx.save_for_backward(x)
        y = torch.empty_like(x)
        check(lib.ur_silu_forward(x.data_ptr(), y.data_ptr(), x.numel(), DT[x.dtype], _stream()), "ur_silu_forward")
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return bw.silu_backward(x, dy.contiguous())


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def linear(x, w, b=None, res=None, rowadd=None, rows_per_b=0):
    if b is not None and deferred_bias:
        b = deferred_bias.get(id(b), b)
    return Linear.apply(x, w, b, res, rowadd, rows_per_b)


def conv3x3(x, wp, b=None, res=None, rowadd=None, stride=1):
    if b is not None and deferred_bias:
        b = deferred_bias.get(id(b), b)
    return Conv3x3.apply(x, wp, b, res, rowadd, stride)


class CatAdjacent(Function):
    """``torch.cat(ts, 0)`` for matrices that already lie back to back in memory (the packed outputs of CastParams): the
    result is a view-like tensor over their storage, no copy; the backward hands every input its row slice of the gradient."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.rows = [t.shape[0] for t in ts]
        n = sum(ctx.rows)
        t0 = ts[0]
        # a NEW tensor object over the same storage (not an autograd view of an input: its gradient is defined below)
        return torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage(), t0.storage_offset(),
                                                                     (n,) + tuple(t0.shape[1:]), t0.stride())

    @staticmethod
    def backward(ctx, g):
        return tuple(torch.split(g, ctx.rows, 0))


def cat_adjacent(ts):
    """Row-wise concatenation of 2-D tensors; free when they are adjacent slices of one buffer, ``torch.cat`` otherwise."""
    ts = list(ts)
    ok = len(ts) > 1 and all(t.dim() == 2 and t.is_contiguous() and t.shape[1:] == ts[0].shape[1:] and t.dtype == ts[0].dtype
                             and t.untyped_storage().data_ptr() == ts[0].untyped_storage().data_ptr() for t in ts)
    if ok:
        off = ts[0].storage_offset()
        for t in ts:
            ok = ok and t.storage_offset() == off
            off += t.numel()
    if not ok:
        return torch.cat(ts, 0)
    return CatAdjacent.apply(*ts)


class PackConvWeight(Function):
    """fp32 master [Co, Ci, 3, 3] -> compute-dtype [Co][(ky, kx, ci_pad)] (ur_pack_conv_weight) and the packed weight
    gradient back to an fp32 [Co, Ci, 3, 3] gradient (ur_unpack_conv_weight_grad): one kernel each way."""

    @staticmethod
    def forward(ctx, weight, dtype, cin_pad):
        lib = _lib.load()
        co, ci = weight.shape[:2]
        cp = ci if cin_pad is None else cin_pad
        ctx.geom = (co, ci, cp)
        ctx.pid = id(weight)
        w = weight.detach().contiguous()
        out = torch.empty(co, 9 * cp, dtype=dtype, device=weight.device)
        check(lib.ur_pack_conv_weight(w.data_ptr(), out.data_ptr(), co, ci, cp, DT[dtype], _stream()), "ur_pack_conv_weight")
        return out

    @staticmethod
    def backward(ctx, dwp):
        lib = _lib.load()
        co, ci, cp = ctx.geom
        if dwp.stride(-1) != 1:
            dwp = dwp.contiguous()
        g = bw.grad_sink.take(ctx.pid)  # the parameter's slice of its communication bucket, or None
        if g is None or tuple(g.shape) != (co, ci, 3, 3):
            g = torch.empty(co, ci, 3, 3, dtype=torch.float32, device=dwp.device)
        if bw.FUSED_GRADNORM:
            part = torch.empty(int(lib.ur_unpack_conv_weight_grad_blocks(co, ci)), dtype=torch.float32, device=dwp.device)
            check(lib.ur_unpack_conv_weight_grad_sumsq(dwp.data_ptr(), dwp.stride(0), g.data_ptr(), co, ci, cp, part.data_ptr(),
                                                       DT[dwp.dtype], _stream()), "ur_unpack_conv_weight_grad_sumsq")
            bw.grad_squares.add(part, [ctx.pid])
        else:
            check(lib.ur_unpack_conv_weight_grad(dwp.data_ptr(), dwp.stride(0), g.data_ptr(), co, ci, cp, DT[dwp.dtype], _stream()),
                  "ur_unpack_conv_weight_grad")
        return g, None, None


class PackConvWeights(Function):
    """PackConvWeight for ALL 3x3 conv weights of a network as one autograd node: the packed copies are made up front, and the
    backward runs when the last packed gradient has arrived -- it flushes the deferred weight-gradient queue first
    (backward.WgradQueue: Conv3x3.backward hands out uninitialised dw for these weights), then unpacks every gradient."""

    @staticmethod
    def forward(ctx, dtype, cin_pads, *weights):
        import weakref
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        ctx.geoms, ctx.pids, outs = [], [id(w) for w in weights], []
        for w_, cpad in zip(weights, cin_pads):
            co, ci = w_.shape[:2]
            cp = ci if cpad is None else cpad
            ctx.geoms.append((co, ci, cp))
            out = torch.empty(co, 9 * cp, dtype=dtype, device=w_.device)
            check(lib.ur_pack_conv_weight(w_.detach().contiguous().data_ptr(), out.data_ptr(), co, ci, cp, DT[dtype], _stream()),
                  "ur_pack_conv_weight")
            _deferred_w[out.untyped_storage().data_ptr()] = weakref.ref(out)
            outs.append(out)
        # the dgrad form of every weight (taps rotated, channels transposed), 32 per launch (bw.weight_rot; dropped in the backward)
        ctx.rot_keys = []
        bw.weight_rot.sweep()  # entries of a forward that was never backpropagated, whose packed weights are gone
        if bw.BATCH_WT:
            elig = [(o, g[2]) for o, g in zip(outs, ctx.geoms) if g[0] % 64 == 0 and g[2] % 8 == 0]
            for (o, _), r in zip(elig, bw.rot_weights_many(elig) if elig else []):
                key = (o.data_ptr(), tuple(o.shape))
                bw.weight_rot.put(key, o, r)  # valid while the packed weight `o` is alive
                ctx.rot_keys.append(key)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        bw.wgrad_queue.flush()
        for key in ctx.rot_keys:
            bw.weight_rot.pop(key, None)
        res = []
        for dwp, (co, ci, cp), pid in zip(grads, ctx.geoms, ctx.pids):
            if dwp is None:
                res.append(None)
                continue
            if dwp.stride(-1) != 1:
                dwp = dwp.contiguous()
            g = bw.grad_sink.take(pid)  # the parameter's slice of its communication bucket, or None
            if g is None or tuple(g.shape) != (co, ci, 3, 3):
                g = torch.empty(co, ci, 3, 3, dtype=torch.float32, device=dwp.device)
            if bw.FUSED_GRADNORM:
                part = torch.empty(int(lib.ur_unpack_conv_weight_grad_blocks(co, ci)), dtype=torch.float32, device=dwp.device)
                check(lib.ur_unpack_conv_weight_grad_sumsq(dwp.data_ptr(), dwp.stride(0), g.data_ptr(), co, ci, cp, part.data_ptr(),
                                                           DT[dwp.dtype], _stream()), "ur_unpack_conv_weight_grad_sumsq")
                bw.grad_squares.add(part, [pid])
            else:
                check(lib.ur_unpack_conv_weight_grad(dwp.data_ptr(), dwp.stride(0), g.data_ptr(), co, ci, cp, DT[dwp.dtype], _stream()),
                      "ur_unpack_conv_weight_grad")
            res.append(g)
        return (None, None, *res)


packed_conv: dict = {}  # (id(fp32 conv weight), cin_pad) -> its PackConvWeights output for the current network forward (train_step)


def pack_conv_weight(weight: torch.Tensor, dtype, cin_pad=None) -> torch.Tensor:
    """differentiable version of layers.pack_conv3x3: [Co, Ci, 3, 3] fp32 master -> [Co][(ky,kx,ci_pad)] compute dtype."""
    if packed_conv:
        hit = packed_conv.get((id(weight), cin_pad))
        if hit is not None and hit.dtype == dtype:
            return hit
    if weight.is_cuda and weight.dtype == torch.float32 and dtype in DT and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3):
        return PackConvWeight.apply(weight, dtype, cin_pad)
    co, ci = weight.shape[:2]
    w = weight.to(dtype).permute(0, 2, 3, 1)  # cast first: the strided repack then moves 2-byte elements
    if cin_pad is not None and cin_pad != ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
    return w.reshape(co, -1).contiguous()

class CastParams(Function):
    """fp32 master parameters -> compute dtype for MANY tensors per launch (``ur_cast_multi``: 128 tensors each; torch's
    ``_foreach_copy_`` takes its per-tensor path for mixed dtypes), and their gradients back to fp32 the same way when the
    last of them has arrived.  Replaces one cast kernel per Linear
    weight in each direction (~370 + ~330 launches per training step)."""

    @staticmethod
    def forward(ctx, dtype, *params):
        import weakref
        ctx.set_materialize_grads(False)
        ctx.pids = [id(p) for p in params]
        outs = tuple(bw.cast_many([p.detach() for p in params], dtype, packed=True))
        ctx.wt_keys = []
        if outs and outs[0].is_cuda:
            flat = outs[0]._base if outs[0]._base is not None else outs[0]
            _deferred_w[flat.untyped_storage().data_ptr()] = weakref.ref(flat)
            bw.weight_t.sweep()  # entries of a forward that was never backpropagated, whose cast buffer is gone
            if bw.BATCH_WT and bw.WGRAD:
                # W^T for the dx GEMMs of the backward, 32 matrices per launch (bw.weight_t; dropped in the backward below)
                w2 = [o.reshape(o.shape[0], -1) for o in outs if o.dim() >= 2]
                w2 = [o for o in w2 if o.shape[0] % 8 == 0 and o.shape[1] % 8 == 0 and o.numel()]
                for o, t in zip(w2, bw.transpose2d_many(w2) if w2 else []):
                    key = (o.data_ptr(), tuple(o.shape))
                    bw.weight_t.put(key, flat, t)  # valid while the flat cast buffer is alive
                    ctx.wt_keys.append(key)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        bw.wgrad_queue.flush()  # the weight gradients below may have been handed out uninitialised (Linear.backward)
        for key in ctx.wt_keys:
            bw.weight_t.pop(key, None)
        idx = [i for i, g in enumerate(grads) if g is not None]
        res = [None] * len(grads)
        if idx:
            # destinations: the parameters' slices of their communication buckets where a sink is installed (bw.GradSink)
            sink = None
            if bw.grad_sink.provider is not None:
                sink = [bw.grad_sink.take(ctx.pids[i]) for i in idx]
                sink = [v if (v is not None and v.numel() == grads[i].numel()) else None for v, i in zip(sink, idx)]
            if bw.FUSED_GRADNORM:  # the sums of squares of what is written here feed the clipping norm (bw.GradSquares)
                outs, part = bw.cast_many([grads[i] for i in idx], torch.float32, sumsq=True, outs=sink)
                if part is not None:
                    bw.grad_squares.add(part, [ctx.pids[i] for i in idx if grads[i].numel()])
            else:
                outs = bw.cast_many([grads[i] for i in idx], torch.float32, outs=sink)
            for i, o_ in zip(idx, outs):
                res[i] = o_
        return (None, *res)
