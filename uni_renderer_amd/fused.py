"""Grouped ("two streams, one launch") execution of the dual-stream denoise step.

Uni-Renderer's step is three networks, but structurally it is TWO copies of one UNet running side by side:

    phase 1   AttributeEncoder.{conv_in, down, mid}   ||   UNet.{conv_in, down, mid}     (same shapes, different weights)
    phase 2   the 13 + 13 exchange 1x1 convs           (enc -> unet  and  unet -> dec, pairwise the same shapes)
    phase 3   UNet.{up, conv_out}                      ||   AttributeDecoder.{up, conv_out}

(`from_unet` guarantees the shape identity: controlnet.py:1437-1507, 2115-2192 of the reference.)  On a 256-CU
MI355X a batch-4 layer is too small to use large GEMM tiles -- throughput is bounded by the global->LDS fill rate
per CU, i.e. by the tile's arithmetic intensity -- so instead of launching the two copies one after the other (or
concurrently on two streams) every op of a pair is issued ONCE with `zbatch = 2`: activations are stacked
stream-major ([2B, H, W, C]), weights are stacked [2, N, K], and `ur_igemm` / norms select per-stream operands by
stride.  M doubles, tiles grow, launches halve.  The exchange uses one grouped 1x1 GEMM per skip whose residual
operand is the OTHER stream's tensor (negative per-z stride), which directly produces the stacked inputs of phase 3:
    [unet_skip + zc_i(enc_skip) ; enc_skip + cd_i(unet_skip)].

This is an execution strategy for the SAME arithmetic as the module-by-module path (controlnet.py classes); parity
between the two is tested bit-for-bit-close in tests/test_fused_gpu.py.  The module API remains the drop-in surface.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Sequence

import torch

from . import _experiments as X
from . import ops, tchain
from .controlnet import CIN_PAD, _compute_dtype
from .layers import (LOG2E, Attention, BasicTransformerBlock, ResnetBlock2D, Transformer2DModel, f32, geglu_perm,
                     pack_conv3x3, pack_matrix)


# q | k | v of a self-attention as ONE grouped GEMM with a transposed side output for V (ur_igemm_desc.out_vt, ABI 8);
# UR_EXPERIMENT=no_qkv_one_launch restores the q | k GEMM + transposed V projection pair (same-box A/B runs)
QKV_ONE_LAUNCH = X.flag("qkv_one_launch", True)
# the same for the prompt's K / V^T projections of a phase (one GEMM over [Wk; Wv] instead of a K GEMM + a V^T GEMM per stream)
CTXKV_ONE_LAUNCH = X.flag("ctxkv_one_launch", True)
# token rows per launch from which the 320-channel chain kernels (128 rows per workgroup) replace the GEMM-by-GEMM path:
# 32768 rows (grouped step at batch 4) and 16384 (hoisted step; grouped step at batch 2: 8.65 vs 8.72 ms) are wins, 8192 (hoisted step
# at batch 2: 6.47 vs 6.24 ms) and 4096 (cfg 2: 4.55 vs 4.14 ms) losses -- profiles/r06_tchain_rows_ab.txt, r05_hoist_ab.txt
TCHAIN_MIN_ROWS = X.number("tchain_min_rows", 16384)
# the chain kernels hand q / k to the d = 40 attention as head-major images; UR_EXPERIMENT=no_head_major_qk: token matrices (A/B)
HEAD_MAJOR_QK = X.flag("head_major_qk", True)


class _Packs:
    """Cache of stream-stacked packed tensors keyed by (name, module ids, dtype) + parameter versions."""

    def __init__(self):
        self._store = {}

    def get(self, name, mods, params, dtype, build):
        key = (name, tuple(id(m) for m in mods), dtype)
        ver = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = build()
        self._store[key] = (ver, val)
        return val


def _stk(ts):
    return torch.stack(list(ts), 0).contiguous()


class GroupedDualStreamStep:
    """enc + unet + dec step with every op of the two diffusion streams issued as one grouped kernel."""

    def __init__(self, unet, enc, dec, precise_residual: Optional[bool] = None):
        """precise_residual: carry the residual stream (block outputs, the x + f(x) adds, skips, exchange sums) as
        (hi, lo) fp16/bf16 pairs (include/ur_kernels.h) -- removes the random walk of the storage rounding along the
        residual path, which is 55 % of the step's fp16 error variance, for 2 extra bytes per
        element on those tensors.  Default: env UR_PRECISE_RESIDUAL (1 / 0), on."""
        self.unet, self.enc, self.dec = unet, enc, dec
        self.pk = _Packs()
        if precise_residual is None:
            precise_residual = os.environ.get("UR_PRECISE_RESIDUAL", "1") != "0"
        self.hilo = bool(precise_residual)
        # row-local chain kernels at the 320-channel level (tchain.py); UR_EXPERIMENT=no_tchain: the unfused GEMM / LayerNorm launches
        self.use_tchain = X.flag("tchain", True)
        # Independent work on a second HIP stream (= a parallel branch of the captured graph): the 13 exchange GEMMs
        # (each only needs its own skip pair, which phase 1 produces early), the up phase's time / prompt projections
        # (inputs only) and every self-attention's V^T projection (beside its q|k projection).  These launches are
        # small (a few hundred workgroups, 15-25 us each); serialised behind the dependent chain they cost their full
        # latency, on a sibling branch they fill the chip beside the chain's own launches.  Default: off.
        # MEASURED (r02, MI355X, cfg 3): 13.06 ms/step without, 13.53 ms with all three kinds of forks -- every
        # cross-stream edge of a hipGraph costs more in dependency signalling than the overlap returns -- so the default
        # is OFF; UR_EXPERIMENT=side_stream=1 all forks, 2 only the off-critical-path ones (exchange + up-phase context), 3 only V^T.
        self.side_level = X.number("side_stream", 0)
        self.use_side = self.side_level != 0
        self._side = None

    # ------------------------------------------------------------------ second stream (graph branch)
    class _Fork:
        """``with step._fork(deps) as f:`` runs the body on the side stream after everything enqueued on the current
        stream so far; ``f.join(*outs)`` makes the current stream wait for it.  Tensors crossing streams are recorded
        with the caching allocator (``deps`` are read on the side stream, ``outs`` on the main one)."""

        def __init__(self, owner, deps, kind):
            self.o, self.deps = owner, deps
            self.main = torch.cuda.current_stream()
            self.active = owner.side_level == 1 or (owner.side_level == 2 and kind != "vt") or (owner.side_level == 3 and kind == "vt")

        @staticmethod
        def _rec(t, stream):
            if torch.is_tensor(t):
                t.record_stream(stream)
                lo = ops.lo_of(t)
                if lo is not None:
                    lo.record_stream(stream)

        def __enter__(self):
            if self.active:
                if self.o._side is None:
                    self.o._side = torch.cuda.Stream()
                self.side = self.o._side
                self.side.wait_stream(self.main)
                for t in self.deps:
                    self._rec(t, self.side)
                self._ctx = torch.cuda.stream(self.side)
                self._ctx.__enter__()
            return self

        def __exit__(self, *exc):
            if self.active:
                self._ctx.__exit__(*exc)
            return False

        def join(self, *outs):
            if self.active:
                self.main.wait_stream(self.side)
                for t in outs:
                    self._rec(t, self.main)

    def _fork(self, *deps, kind="x"):
        return GroupedDualStreamStep._Fork(self, deps, kind)

    # ------------------------------------------------------------------ leaves (S streams in lockstep)
    def _resnet(self, rs: Sequence[ResnetBlock2D], x, temb, slice_, x1=None):
        S, pk, dt = len(rs), self.pk, x.dtype
        r0 = rs[0]
        g1 = pk.get("r.g1", rs, [r.norm1.weight for r in rs], dt, lambda: _stk(f32(r.norm1.weight) for r in rs))
        b1 = pk.get("r.b1", rs, [r.norm1.bias for r in rs], dt, lambda: _stk(f32(r.norm1.bias) for r in rs))
        g2 = pk.get("r.g2", rs, [r.norm2.weight for r in rs], dt, lambda: _stk(f32(r.norm2.weight) for r in rs))
        b2 = pk.get("r.b2", rs, [r.norm2.bias for r in rs], dt, lambda: _stk(f32(r.norm2.bias) for r in rs))
        w1 = pk.get("r.w1", rs, [r.conv1.weight for r in rs], dt, lambda: _stk(pack_conv3x3(r.conv1.weight, dt, cblock=ops.conv_cblock(r.conv1.weight.shape[1])) for r in rs))
        c1 = pk.get("r.c1", rs, [r.conv1.bias for r in rs], dt, lambda: _stk(f32(r.conv1.bias) for r in rs))
        w2 = pk.get("r.w2", rs, [r.conv2.weight for r in rs], dt, lambda: _stk(pack_conv3x3(r.conv2.weight, dt, cblock=ops.conv_cblock(r.conv2.weight.shape[1])) for r in rs))
        c2 = pk.get("r.c2", rs, [r.conv2.bias for r in rs], dt, lambda: _stk(f32(r.conv2.bias) for r in rs))
        lo, hi = slice_
        h = ops.groupnorm(x, g1, b1, r0.eps, x1=x1, groups=r0.groups, silu=True, streams=S)
        # conv1 -> norm2 -> SiLU: conv1's output has no other consumer, so where conv1 runs split-K on a small map (the
        # 16x16 / 8x8 levels) the GroupNorm is the split-K second pass and the conv output is never written (ops.conv3x3 gn=)
        h = ops.conv3x3(h, w1, c1, rowadd=temb[:, lo:hi], streams=S, cblock=ops.conv_cblock(h.shape[-1]),
                        ws=self._ws("r.w1ws", rs, [r.conv1.weight for r in rs], w1, h, streams=S),
                        gn=(g2, b2, r0.eps, r0.groups, True))
        if r0.conv_shortcut is not None and ops.FOLD_SHORTCUT:
            # the 1x1 conv_shortcut over (x | x1) rides in conv2's K loop (ur_igemm_desc.t0 / t1): one launch less and no
            # round trip of its output through memory
            w2s = pk.get("r.w2s", rs, [t for r in rs for t in (r.conv2.weight, r.conv_shortcut.weight)], dt,
                         lambda: _stk(torch.cat([pack_conv3x3(r.conv2.weight, dt, cblock=ops.conv_cblock(r.conv2.weight.shape[1])),
                                                 pack_matrix(r.conv_shortcut.weight, dt)], 1) for r in rs))
            c2s = pk.get("r.c2s", rs, [t for r in rs for t in (r.conv2.bias, r.conv_shortcut.bias)], dt,
                         lambda: _stk(f32(r.conv2.bias) + f32(r.conv_shortcut.bias) for r in rs))
            return ops.conv3x3(h, w2s, c2s, tail=(x, x1), out_scale=1.0 / r0.output_scale_factor, streams=S, hilo=self.hilo,
                               cblock=ops.conv_cblock(h.shape[-1]),
                               ws=self._ws("r.w2sws", rs, [t for r in rs for t in (r.conv2.weight, r.conv_shortcut.weight)], w2s, h,
                                           tail=(x, x1), streams=S))
        if r0.conv_shortcut is not None:
            ws = pk.get("r.ws", rs, [r.conv_shortcut.weight for r in rs], dt,
                        lambda: _stk(pack_matrix(r.conv_shortcut.weight, dt) for r in rs))
            bs = pk.get("r.bs", rs, [r.conv_shortcut.bias for r in rs], dt, lambda: _stk(f32(r.conv_shortcut.bias) for r in rs))
            sc = ops.linear(x, ws, bs, x1=x1, streams=S)
        else:
            sc = x
        return ops.conv3x3(h, w2, c2, res=sc, out_scale=1.0 / r0.output_scale_factor, streams=S, hilo=self.hilo,
                           cblock=ops.conv_cblock(h.shape[-1]),
                           ws=self._ws("r.w2ws", rs, [r.conv2.weight for r in rs], w2, h, streams=S))

    def _ws(self, name, mods, params, w, x, tail=None, streams=1):
        """Stage-image copy of the packed conv weights ``w`` [S, N, K] for the weight-streaming kernel, or None where the
        LDS-tiled build is used (ops.wsconv_prefer)."""
        N, K = w.shape[-2], w.shape[-1]
        if not ops.wsconv_prefer(x, N, K, tail=tail, streams=streams):
            return None
        return self.pk.get(name, mods, params, x.dtype, lambda: _stk(ops.wsconv_images(w[i]) for i in range(w.shape[0])))

    def _attn(self, as_: Sequence[Attention], xn, residual, kc, vtc, kv_slice):
        S, pk, dt = len(as_), self.pk, xn.dtype
        a0 = as_[0]
        Bt, T, _ = xn.shape
        H, d, C = a0.heads, a0.dim_head, a0.inner
        cs = d ** -0.5 * LOG2E
        wo = pk.get("a.wo", as_, [a.to_out[0].weight for a in as_], dt, lambda: _stk(pack_matrix(a.to_out[0].weight, dt) for a in as_))
        bo = pk.get("a.bo", as_, [a.to_out[0].bias for a in as_], dt, lambda: _stk(f32(a.to_out[0].bias) for a in as_))
        if not a0.is_cross:
            if QKV_ONE_LAUNCH and T % 64 == 0:
                # q | k | v as ONE grouped GEMM: the value columns leave the epilogue transposed (ur_igemm_desc.out_vt), which
                # retires the separate V^T projection launch of every self-attention of the 32x32 .. 8x8 levels
                wqkv = pk.get("a.wqkv", as_, [p for a in as_ for p in (a.to_q.weight, a.to_k.weight, a.to_v.weight)], dt,
                              lambda: _stk(torch.cat([pack_matrix(a.to_q.weight, dt), pack_matrix(a.to_k.weight, dt),
                                                      pack_matrix(a.to_v.weight, dt)], 0) for a in as_))
                ops.set_site("qkv")
                qk, vt = ops.linear(xn, wqkv, streams=S, out_scale=math.sqrt(cs), vt_cols=C, vt_tokens=T)
                ops.set_site(None)
            else:
                wqk = pk.get("a.wqk", as_, [p for a in as_ for p in (a.to_q.weight, a.to_k.weight)], dt,
                             lambda: _stk(torch.cat([pack_matrix(a.to_q.weight, dt), pack_matrix(a.to_k.weight, dt)], 0) for a in as_))
                wv = pk.get("a.wv", as_, [a.to_v.weight for a in as_], dt, lambda: _stk(pack_matrix(a.to_v.weight, dt) for a in as_))
                vt_first = X.flag("vt_first", True)
                ops.set_site("qk")
                if not vt_first:
                    qk = ops.linear(xn, wqk, streams=S, out_scale=math.sqrt(cs))
                with self._fork(xn, kind="vt") as f:  # V^T projection on the sibling branch, beside the q|k projection
                    ops.set_site("vt")
                    vt = ops.vt_proj(xn, wv, streams=S)
                    ops.set_site("qk")
                if vt_first:
                    qk = ops.linear(xn, wqk, streams=S, out_scale=math.sqrt(cs))  # scale folded in, see layers.Attention
                f.join(vt)
            o = ops.attention(qk, qk, vt, B=Bt, H=H, Tq=T, Tk=T, d=d, ldq=2 * C, ldk=2 * C, q_off=0, k_off=C, scale=0.0)
        else:
            wq = pk.get("a.wq", as_, [a.to_q.weight for a in as_], dt, lambda: _stk(pack_matrix(a.to_q.weight, dt) for a in as_))
            ops.set_site("q")
            q = ops.linear(xn, wq, streams=S, out_scale=cs)
            lo, hi = kv_slice
            o = ops.attention(q, kc[:, :, lo:hi], vtc[:, lo:hi], B=Bt, H=H, Tq=T, Tk=kc.shape[1], d=d, ldq=C,
                              ldk=kc.stride(1), scale=0.0)
        ops.set_site("co" if a0.is_cross else "ao")
        y = ops.linear(o, wo, bo, res=residual, streams=S, hilo=self.hilo)
        ops.set_site(None)
        return y

    def _tblock(self, bs: Sequence[BasicTransformerBlock], x, kc, vtc, kv_slice):
        S, pk, dt = len(bs), self.pk, x.dtype

        def ln(name, get):
            g = pk.get(name + ".g", bs, [get(b).weight for b in bs], dt, lambda: _stk(f32(get(b).weight) for b in bs))
            b_ = pk.get(name + ".b", bs, [get(b).bias for b in bs], dt, lambda: _stk(f32(get(b).bias) for b in bs))
            return g, b_

        g, b_ = ln("t.n1", lambda b: b.norm1)
        x = self._attn([b.attn1 for b in bs], ops.layernorm(x, g, b_, bs[0].norm1.eps, streams=S), x, None, None, None)
        g, b_ = ln("t.n2", lambda b: b.norm2)
        x = self._attn([b.attn2 for b in bs], ops.layernorm(x, g, b_, bs[0].norm2.eps, streams=S), x, kc, vtc, kv_slice)
        g, b_ = ln("t.n3", lambda b: b.norm3)
        xn = ops.layernorm(x, g, b_, bs[0].norm3.eps, streams=S)
        projs = [b.ff.net[0].proj for b in bs]
        outs = [b.ff.net[2] for b in bs]
        nh = projs[0].weight.shape[0] // 2

        def build_in():
            perm = geglu_perm(nh, projs[0].weight.device)
            return (_stk(pack_matrix(p.weight, dt)[perm] for p in projs), _stk(f32(p.bias)[perm] for p in projs))

        w_in, b_in = pk.get("t.ffi", bs, [p for pr in projs for p in (pr.weight, pr.bias)], dt, build_in)
        w_out = pk.get("t.ffo", bs, [o.weight for o in outs], dt, lambda: _stk(pack_matrix(o.weight, dt) for o in outs))
        b_out = pk.get("t.ffb", bs, [o.bias for o in outs], dt, lambda: _stk(f32(o.bias) for o in outs))
        ops.set_site("ffi")
        gg = ops.linear(xn, w_in, b_in, act=ops.ACT_GEGLU, streams=S)
        ops.set_site("ffo")
        y = ops.linear(gg, w_out, b_out, res=x, streams=S, hilo=self.hilo)
        ops.set_site(None)
        return y

    def _tblock_chain(self, bs: Sequence[BasicTransformerBlock], ts, x, blk_in, kc, vtc, kv_slice):
        """The 320-channel level: everything row-local around the two attentions runs as three ``ur_tchain`` launches
        (tchain.py) -- proj_in + LayerNorm1 + q / k / V^T projections; attn1 out-projection + residual + LayerNorm2 +
        cross-attention query projection; attn2 out-projection + residual + LayerNorm3 + GEGLU feed-forward + residual +
        proj_out + block input -- instead of ten GEMM and three LayerNorm launches.  ``x`` = the GroupNorm output, ``blk_in``
        the transformer's input.  Same arithmetic as ``_tblock`` + proj_in / proj_out; the activations round at the same points,
        with one difference on the weight side: the chain folds the q / k scale sqrt(d^-1/2 log2 e) into the projection WEIGHTS
        before their fp16 / bf16 cast (tchain.pack_*), where ``_attn`` applies ``out_scale`` to the fp32 accumulator."""
        S, pk, dt = len(bs), self.pk, x.dtype
        a1, a2 = [b.attn1 for b in bs], [b.attn2 for b in bs]
        a0 = a1[0]
        Bt, T, C = x.shape
        H, d = a0.heads, a0.dim_head
        cs = d ** -0.5 * LOG2E
        def build_pre():
            packs = [tchain.pack_chain_pre(t.proj_in.weight, t.proj_in.bias, b.norm1.weight, b.norm1.bias, b.attn1.to_q.weight,
                                           b.attn1.to_k.weight, b.attn1.to_v.weight, math.sqrt(cs), dt) for b, t in zip(bs, ts)]
            return _stk(p[0] for p in packs), _stk(p[1] for p in packs)

        wsp, csp = pk.get("tc.pre", bs, [p for b, t in zip(bs, ts) for p in (
            t.proj_in.weight, t.proj_in.bias, b.norm1.weight, b.norm1.bias, b.attn1.to_q.weight, b.attn1.to_k.weight,
            b.attn1.to_v.weight)], dt, build_pre)
        # x = the GroupNorm output: proj_in + LayerNorm1 + q / k / V^T projections in one launch
        # q / k leave the chain HEAD-MAJOR ([sample][head][token][40], round 6): a head's 64-key tile is one 5 KB run instead of 64
        # 80-byte slices of 640-byte token rows -- at the 128x128 level the slices made every XCD fetch 2.4 lines per line used and
        # a head's keys no longer fit its L2 (profiles/r06_pmc_attn_l2_cfg5.json)
        hm = HEAD_MAJOR_QK and H == tchain.HEADS and T % 32 == 0
        xr, q1, k1, vt = tchain.chain_pre(x.reshape(Bt * T, C), wsp, csp, bs[0].norm1.eps, tokens_per_sample=T, streams=S, head_major=hm)
        o1 = ops.attention(q1, k1, vt, B=Bt, H=H, Tq=T, Tk=T, d=d, ldq=C, ldk=C, scale=0.0, q_hstride=T * d if hm else 0,
                           k_hstride=T * d if hm else 0)
        x = xr.view(Bt, T, C)
        x.lo = xr.lo.view(Bt, T, C)

        def build_q():
            packs = [tchain.pack_chain_q(b.attn1.to_out[0].weight, b.attn1.to_out[0].bias, b.norm2.weight, b.norm2.bias,
                                         b.attn2.to_q.weight, cs, dt) for b in bs]
            return _stk(p[0] for p in packs), _stk(p[1] for p in packs)

        wsq, csq = pk.get("tc.q", bs, [p for b in bs for p in (b.attn1.to_out[0].weight, b.attn1.to_out[0].bias, b.norm2.weight,
                                                               b.norm2.bias, b.attn2.to_q.weight)], dt, build_q)
        y1, q2 = tchain.chain_q(o1.view(Bt * T, C), ops.view_hilo(x, Bt * T, C), wsq, csq, bs[0].norm2.eps, streams=S,
                                head_major_tokens=T if hm else 0)
        lo, hi = kv_slice
        o2 = ops.attention(q2.view(Bt, T, C), kc[:, :, lo:hi], vtc[:, lo:hi], B=Bt, H=H, Tq=T, Tk=kc.shape[1], d=d, ldq=C,
                           ldk=kc.stride(1), scale=0.0, q_hstride=T * d if hm else 0)

        def build_ff():
            packs = [tchain.pack_chain_ff(b.attn2.to_out[0].weight, b.attn2.to_out[0].bias, b.norm3.weight, b.norm3.bias,
                                          b.ff.net[0].proj.weight, b.ff.net[0].proj.bias, b.ff.net[2].weight, b.ff.net[2].bias,
                                          t.proj_out.weight, t.proj_out.bias, dt) for b, t in zip(bs, ts)]
            return _stk(p[0] for p in packs), _stk(p[1] for p in packs)

        wsf, csf = pk.get("tc.ff", bs, [p for b, t in zip(bs, ts) for p in (
            b.attn2.to_out[0].weight, b.attn2.to_out[0].bias, b.norm3.weight, b.norm3.bias, b.ff.net[0].proj.weight,
            b.ff.net[0].proj.bias, b.ff.net[2].weight, b.ff.net[2].bias, t.proj_out.weight, t.proj_out.bias)], dt, build_ff)
        out = tchain.chain_ff(o2.view(Bt * T, C), y1, ops.view_hilo(blk_in, Bt * T, C), wsf, csf, bs[0].norm3.eps, streams=S)
        return ops.view_hilo(out, Bt, T, C)

    def _transformer(self, ts: Sequence[Transformer2DModel], x, kc, vtc, kv_slices):
        S, pk, dt = len(ts), self.pk, x.dtype
        Bt, H, W, Cc = x.shape
        g = pk.get("x.g", ts, [t.norm.weight for t in ts], dt, lambda: _stk(f32(t.norm.weight) for t in ts))
        b_ = pk.get("x.b", ts, [t.norm.bias for t in ts], dt, lambda: _stk(f32(t.norm.bias) for t in ts))
        wi = pk.get("x.wi", ts, [t.proj_in.weight for t in ts], dt, lambda: _stk(pack_matrix(t.proj_in.weight, dt) for t in ts))
        bi = pk.get("x.bi", ts, [t.proj_in.bias for t in ts], dt, lambda: _stk(f32(t.proj_in.bias) for t in ts))
        wo = pk.get("x.wo", ts, [t.proj_out.weight for t in ts], dt, lambda: _stk(pack_matrix(t.proj_out.weight, dt) for t in ts))
        bo = pk.get("x.bo", ts, [t.proj_out.bias for t in ts], dt, lambda: _stk(f32(t.proj_out.bias) for t in ts))
        h = ops.groupnorm(x, g, b_, ts[0].norm.eps, groups=ts[0].groups, silu=False, streams=S)
        b0 = ts[0].transformer_blocks[0]
        # the chain kernels hard-code C = 320 (tchain.supported), 8 heads of 40, bias-free q / k / v and a feed-forward of
        # 2 x 1280 -> 320; anything else takes the GEMM-by-GEMM path
        # ... and one chain workgroup owns 128 token rows: below TCHAIN_MIN_ROWS rows per launch the chain leaves most of the 256 CUs
        # idle (cfg 2: 4096 rows = 32 workgroups, 109 us for the feed-forward chain at 75 TFLOP/s) and the tiled GEMMs win
        if (self.use_tchain and len(ts[0].transformer_blocks) == 1 and tchain.supported(h) and self.hilo
                and Bt * H * W >= TCHAIN_MIN_ROWS
                and (H * W) % 32 == 0 and b0.attn1.dim_head == 40 and b0.attn1.heads * 40 == Cc
                and b0.ff.net[2].weight.shape[1] == tchain.FF_HIDDEN and b0.ff.net[0].proj.weight.shape[0] == 2 * tchain.FF_HIDDEN
                and all(getattr(a, nm).bias is None for a in (b0.attn1, b0.attn2) for nm in ("to_q", "to_k", "to_v"))):
            return ops.view_hilo(self._tblock_chain([t.transformer_blocks[0] for t in ts], ts, h.view(Bt, H * W, Cc),
                                                    ops.view_hilo(x, Bt, H * W, Cc), kc, vtc, kv_slices[0]), Bt, H, W, Cc)
        ops.set_site("pi")
        h = ops.linear(h.view(Bt, H * W, Cc), wi, bi, streams=S, hilo=self.hilo)
        ops.set_site(None)
        for j in range(len(ts[0].transformer_blocks)):
            h = self._tblock([t.transformer_blocks[j] for t in ts], h, kc, vtc, kv_slices[j])
        ops.set_site("po")
        y = ops.linear(h, wo, bo, res=ops.view_hilo(x, Bt, H * W, Cc), streams=S, hilo=self.hilo)
        ops.set_site(None)
        return ops.view_hilo(y, Bt, H, W, Cc)

    def _conv(self, name, convs, x, stride=1, ups=False):
        S, pk, dt = len(convs), self.pk, x.dtype
        w = pk.get(name + ".w", convs, [c.weight for c in convs], dt, lambda: _stk(pack_conv3x3(c.weight, dt, cblock=ops.conv_cblock(c.weight.shape[1])) for c in convs))
        b = pk.get(name + ".b", convs, [c.bias for c in convs], dt, lambda: _stk(f32(c.bias) for c in convs))
        return ops.conv3x3(x, w, b, stride=stride, ups=ups, streams=S, hilo=self.hilo, cblock=ops.conv_cblock(x.shape[-1]))

    # ------------------------------------------------------------------ per-phase context (temb, prompt K / V^T)
    def _phase_ctx(self, nets, resnet_lists, cross_lists, semb, ehs, temb=True, kv=True):
        """Batched per-resnet time projections and per-cross-attention prompt K / V^T of ONE phase, grouped over the
        streams.  Column layouts are identical across streams because the module lists are.  ``temb`` / ``kv`` = False
        leave that half out (None): the prompt half depends on ``ehs`` only and the sampling loops compute it once per
        call (hoist.py), the time half once per step."""
        S, pk = len(nets), self.pk
        dt = semb.dtype if semb is not None else ehs.dtype
        tslices, off = {}, 0
        for r in resnet_lists[0]:
            tslices[id(r)] = (off, off + r.out_channels)
            off += r.out_channels
        if temb:
            key = tuple(id(r) for rl in resnet_lists for r in rl)
            wt = pk.get(("p.wt", key), nets, [r.time_emb_proj.weight for rl in resnet_lists for r in rl], dt,
                        lambda: _stk(torch.cat([pack_matrix(r.time_emb_proj.weight, dt) for r in rl], 0) for rl in resnet_lists))
            bt = pk.get(("p.bt", key), nets, [r.time_emb_proj.bias for rl in resnet_lists for r in rl], dt,
                        lambda: _stk(torch.cat([f32(r.time_emb_proj.bias) for r in rl], 0) for rl in resnet_lists))
            temb = ops.linear(semb, wt, bt, streams=S)  # [S*B, sum Cout]
        else:
            temb = None
        kc = vtc = None
        kslices = {}
        if cross_lists[0] and kv:
            ckey = tuple(id(a) for al in cross_lists for a in al)
            wk = pk.get(("p.wk", ckey), nets, [a.to_k.weight for al in cross_lists for a in al], dt,
                        lambda: _stk(torch.cat([pack_matrix(a.to_k.weight, dt) for a in al], 0) for al in cross_lists))
            wv = pk.get(("p.wv", ckey), nets, [a.to_v.weight for al in cross_lists for a in al], dt,
                        lambda: _stk(torch.cat([pack_matrix(a.to_v.weight, dt) for a in al], 0) for al in cross_lists))
            B, Tk, Cc = ehs.shape
            n = wk.shape[1]
            kc = torch.empty(S * B, Tk, n, dtype=dt, device=ehs.device)
            if CTXKV_ONE_LAUNCH and n % 16 == 0:
                # prompt K and V^T of every cross-attention of the phase from ONE grouped GEMM over [Wk; Wv]: the value
                # columns leave the epilogue transposed (ur_igemm_desc.out_vt) into a zero-filled [S*B, n, Tpad] tensor
                # (77 keys: the pad columns must be zero for the attention kernel) -- instead of a K GEMM plus one
                # transposed V projection per stream
                wkv = pk.get(("p.wkv", ckey), nets, [t for al in cross_lists for a in al for t in (a.to_k.weight, a.to_v.weight)], dt,
                             lambda: torch.cat([wk, wv], 1).contiguous())
                Tpad = (Tk + 63) // 64 * 64
                vtc = torch.zeros(S * B, n, Tpad, dtype=dt, device=ehs.device)
                ops.igemm(x0=ehs, w=wkv, out=kc, M=B * Tk, N=2 * n, K=Cc, c0=Cc, ldx0=Cc, ldw=Cc, ldc=n, n_store=n, zbatch=S, zx=0,
                          zw=wkv.stride(0), zout=B * Tk * n, out_vt=vtc, vt_n0=n, vt_rows=Tk, zvt=B * n * Tpad)
            else:
                ops.igemm(x0=ehs, w=wk, out=kc, M=B * Tk, N=n, K=Cc, c0=Cc, ldx0=Cc, ldw=Cc, ldc=n, zbatch=S, zx=0,
                          zw=wk.stride(0), zout=B * Tk * n)  # the prompt is shared by the streams (zx = 0)
                vtc = ops.vt_proj(ehs, wv, streams=S, shared_x=True)
            off = 0
            for a in cross_lists[0]:
                kslices[id(a)] = (off, off + a.inner)
                off += a.inner
        return temb, tslices, kc, vtc, kslices

    @staticmethod
    def _resnets_of(mods):
        return [m for mod in mods for m in mod.modules() if isinstance(m, ResnetBlock2D)]

    @staticmethod
    def _cross_of(mods):
        return [m for mod in mods for m in mod.modules() if isinstance(m, Attention) and m.is_cross]

    def _kvs(self, t0: Transformer2DModel, kslices):
        return [kslices[id(b.attn2)] for b in t0.transformer_blocks]

    # ------------------------------------------------------------------ building blocks of a step (S streams in lockstep)
    def _time_embed(self, nets, tvals, B, dt):
        """SiLU(time_embedding(Timesteps(t))) of ``nets`` as one grouped chain: rows [net 0 | net 1 | ...], ``tvals`` one [B]
        fp32 vector per network (controlnet.py:909-916 of the reference; the SiLU is the resnets' ``nonlinearity(temb)``)."""
        pk, S = self.pk, len(nets)
        c = self.unet.config
        ts = torch.cat(list(tvals)).contiguous() if S > 1 else tvals[0].contiguous()
        t_emb = ops.timestep_embedding(ts, S * B, c["block_out_channels"][0], c["flip_sin_to_cos"], c["freq_shift"], dt)
        tes = [n.time_embedding for n in nets]
        w1 = pk.get("te.w1", tes, [t.linear_1.weight for t in tes], dt, lambda: _stk(pack_matrix(t.linear_1.weight, dt) for t in tes))
        b1 = pk.get("te.b1", tes, [t.linear_1.bias for t in tes], dt, lambda: _stk(f32(t.linear_1.bias) for t in tes))
        w2 = pk.get("te.w2", tes, [t.linear_2.weight for t in tes], dt, lambda: _stk(pack_matrix(t.linear_2.weight, dt) for t in tes))
        b2 = pk.get("te.b2", tes, [t.linear_2.bias for t in tes], dt, lambda: _stk(f32(t.linear_2.bias) for t in tes))
        return ops.linear(ops.linear(t_emb, w1, b1, act=ops.ACT_SILU, streams=S), w2, b2, act=ops.ACT_SILU, streams=S)

    def _exchange(self, name, z_enc, z_dec, t, B, scale, dt):
        """t = [enc ; unet] stacked.  Returns [unet + zc(enc)*scale ; enc + cd(unet)] (z_dec given) or only the first half."""
        pk = self.pk
        half = t.numel() // 2
        Cc = t.shape[-1]
        if z_dec is not None:
            mods = [z_enc, z_dec]
            w = pk.get((name, "w", scale), mods, [m.weight for m in mods], dt,
                       lambda: _stk([pack_matrix(z_enc.weight, dt) * scale if scale != 1.0 else pack_matrix(z_enc.weight, dt),
                                     pack_matrix(z_dec.weight, dt)]))
            b = pk.get((name, "b", scale), mods, [m.bias for m in mods], dt,
                       lambda: _stk([f32(z_enc.bias) * scale, f32(z_dec.bias)]))
            tt = t.view(2, -1, Cc)
            tl = ops.lo_of(t)
            y = ops.linear(tt, w, b, res=tt[1], res_zstride=-half, streams=2, hilo=self.hilo,
                           res_lo=(tl.view(2, -1, Cc)[1] if tl is not None else None))
            return ops.view_hilo(y, *t.shape)
        w = pk.get((name, "w1", scale), [z_enc], [z_enc.weight], dt, lambda: (pack_matrix(z_enc.weight, dt) * scale).contiguous())
        b = pk.get((name, "b1", scale), [z_enc], [z_enc.bias], dt, lambda: f32(z_enc.bias) * scale)
        tl = ops.lo_of(t)
        return ops.linear(t[:B], w, b, res=t[B:], hilo=self.hilo, res_lo=(tl[B:] if tl is not None else None))

    def _conv_in(self, nets, x_in):
        pk, dt = self.pk, x_in.dtype
        cins = [n.conv_in for n in nets]
        wci = pk.get("cin.w", cins, [m.weight for m in cins], dt, lambda: _stk(pack_conv3x3(m.weight, dt, CIN_PAD) for m in cins))
        bci = pk.get("cin.b", cins, [m.bias for m in cins], dt, lambda: _stk(f32(m.bias) for m in cins))
        return ops.conv3x3(x_in, wci, bci, streams=len(nets), hilo=self.hilo)

    def _down_mid(self, nets, x, ctx, on_skip):
        """conv_in's output ``x`` through the down blocks and the mid block of ``nets`` (controlnet.py:1051-1115 / 1723-1748);
        ``on_skip(t)`` sees every skip tensor in the reference's order (conv_in output first)."""
        temb, tsl, kc, vtc, ksl = ctx
        on_skip(x)
        for bi_ in range(len(nets[0].down_blocks)):
            blks = [n.down_blocks[bi_] for n in nets]
            for li, r0 in enumerate(blks[0].resnets):
                x = self._resnet([b.resnets[li] for b in blks], x, temb, tsl[id(r0)])
                if getattr(blks[0], "has_cross_attention", False):
                    tsf = [b.attentions[li] for b in blks]
                    x = self._transformer(tsf, x, kc, vtc, self._kvs(tsf[0], ksl))
                on_skip(x)
            if blks[0].downsamplers is not None:
                x = self._conv("ds", [b.downsamplers[0].conv for b in blks], x, stride=2)
                on_skip(x)
        mids = [n.mid_block for n in nets]
        x = self._resnet([m.resnets[0] for m in mids], x, temb, tsl[id(mids[0].resnets[0])])
        for ai, a0 in enumerate(mids[0].attentions):
            tsf = [m.attentions[ai] for m in mids]
            x = self._transformer(tsf, x, kc, vtc, self._kvs(tsf[0], ksl))
            x = self._resnet([m.resnets[ai + 1] for m in mids], x, temb, tsl[id(mids[0].resnets[ai + 1])])
        return x

    def _up(self, nets, x, up_skips, ctx):
        """The up blocks of ``nets`` (controlnet.py:1119-1151 / 2480-2512); ``up_skips`` is consumed from its end."""
        temb, tsl, kc, vtc, ksl = ctx
        for bi_ in range(len(nets[0].up_blocks)):
            blks = [n.up_blocks[bi_] for n in nets]
            for li, r0 in enumerate(blks[0].resnets):
                s = up_skips.pop()
                x = self._resnet([b.resnets[li] for b in blks], x, temb, tsl[id(r0)], x1=s)
                if getattr(blks[0], "has_cross_attention", False):
                    tsf = [b.attentions[li] for b in blks]
                    x = self._transformer(tsf, x, kc, vtc, self._kvs(tsf[0], ksl))
            if blks[0].upsamplers is not None:
                ucs = [b.upsamplers[0].conv for b in blks]
                tgt = tuple(up_skips[-1].shape[1:3])  # the reference's upsample_size (controlnet.py:1129-1130)
                if tgt == (2 * x.shape[1], 2 * x.shape[2]):
                    x = self._conv("us", ucs, x, ups=True)
                else:  # latent side not a multiple of 8: general nearest resize, then the conv
                    x = self._conv("us", ucs, ops.resize_nearest(x, tgt))
        return x

    def _head(self, nets, x):
        """conv_norm_out -> SiLU -> conv_out of ``nets`` (output channels padded to the widest): [S*B, H, W, n_out]."""
        pk, dt, S = self.pk, x.dtype, len(nets)
        norms = [n.conv_norm_out for n in nets]
        g = pk.get("out.g", norms, [m.weight for m in norms], dt, lambda: _stk(f32(m.weight) for m in norms))
        b_ = pk.get("out.b", norms, [m.bias for m in norms], dt, lambda: _stk(f32(m.bias) for m in norms))
        h = ops.groupnorm(x, g, b_, norms[0].eps, groups=norms[0].num_groups, silu=True, streams=S)
        couts = [n.conv_out for n in nets]
        n_out = max(m.weight.shape[0] for m in couts)  # 4 (image) / 28 (attributes): pad to the widest

        def pad_rows(t, n):
            return t if t.shape[0] == n else torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))], 0)

        wco = pk.get("out.w", couts, [m.weight for m in couts], dt, lambda: _stk(pad_rows(pack_conv3x3(m.weight, dt), n_out) for m in couts))
        bco = pk.get("out.cb", couts, [m.bias for m in couts], dt, lambda: _stk(pad_rows(f32(m.bias), n_out) for m in couts))
        return ops.conv3x3(h, wco, bco, n_out=n_out, streams=S)

    def _ctx_of(self, nets, parts, semb, ehs, temb=True, kv=True):
        """``_phase_ctx`` of the sub-modules ``parts[i]`` of ``nets[i]``."""
        rl = [self._resnets_of(p) for p in parts]
        cl = [self._cross_of(p) for p in parts]
        return self._phase_ctx(nets, rl, cl, semb, ehs, temb=temb, kv=kv)

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def __call__(self, x_t, cond, ehs, t_img, t_attr, run_decoder: bool = True, conditioning_scale: float = 1.0
                 ) -> Dict[str, torch.Tensor]:
        unet, enc, dec = self.unet, self.enc, self.dec
        dt = _compute_dtype(unet.dtype, unet.compute_dtype)
        dev = x_t.device
        B, _, H, W = x_t.shape
        ehs = self._prep_ehs(ehs, B, dt)
        tvec = lambda t: self._tvec(t, B, dev)

        # --- time embeddings of the three networks in one grouped chain: rows [enc | unet | dec]
        nets3 = [enc, unet, dec] if run_decoder else [enc, unet]
        semb = self._time_embed(nets3, [tvec(t_attr), tvec(t_img)] + ([tvec(t_attr)] if run_decoder else []), B, dt)

        # ---- the exchange (phase 2): exchange i only needs skip pair i, so it is issued on the sibling branch as soon as
        # phase 1 has produced that pair
        scale = float(conditioning_scale)
        up_skips, forks = [], []
        late = []  # UR_EXPERIMENT=no_exchange_early: all exchange GEMMs after the mid block (round-1 order) instead of right behind
        early = X.flag("exchange_early", True)  # the kernel that produced their skip (input still in L2)

        def exchange_skip(t):
            if not early:
                late.append(t)
                return
            i = len(up_skips)
            with self._fork(t) as f:
                y = self._exchange(f"ex{i}", enc.controlnet_down_blocks[i], dec.control_down_blocks[i] if run_decoder else None,
                                   t, B, scale, dt)
            up_skips.append(y)
            forks.append((f, y))

        # ---- up-phase context (time projections, prompt K / V^T of the up blocks): depends on the inputs only
        pair3 = [unet, dec] if run_decoder else [unet]
        S = len(pair3)
        ctx3_early = X.flag("ctx3_early", True)
        with self._fork(semb, ehs) as f3:
            ctx3 = self._ctx_of(pair3, [[n.up_blocks] for n in pair3], semb[B: B + S * B], ehs) if ctx3_early else None

        # ================= phase 1: enc || unet : conv_in, down, mid =================
        pair = [enc, unet]
        ctx1 = self._ctx_of(pair, [[n.down_blocks, n.mid_block] for n in pair], semb[: 2 * B], ehs)
        x_in = torch.cat([ops.to_nhwc(cond, dt, CIN_PAD), ops.to_nhwc(x_t, dt, CIN_PAD)], 0)
        mid = self._down_mid(pair, self._conv_in(pair, x_in), ctx1, exchange_skip)  # [enc_mid ; unet_mid]

        # ================= phase 2: the mid exchange; join the sibling branch =================
        for i, t in enumerate(late):
            up_skips.append(self._exchange(f"ex{i}", enc.controlnet_down_blocks[i],
                                           dec.control_down_blocks[i] if run_decoder else None, t, B, scale, dt))
        x = self._exchange("exm", enc.controlnet_mid_block, dec.control_mid_block if run_decoder else None, mid, B, scale, dt)
        for f, y in forks:
            f.join(y)
        if ctx3 is None:
            ctx3 = self._ctx_of(pair3, [[n.up_blocks] for n in pair3], semb[B: B + S * B], ehs)
        f3.join(ctx3[0], ctx3[2], ctx3[3])

        # ================= phase 3: unet || dec : up path, conv_out =================
        y = self._head(pair3, self._up(pair3, x, up_skips, ctx3))  # [S*B, H, W, n_out]
        out = {"img_pred": ops.as_nchw_view(y[:B, :, :, : unet.conv_out.weight.shape[0]])}
        if run_decoder:
            out["attr_pred"] = ops.as_nchw_view(y[B:])
        return out

    @staticmethod
    def _prep_ehs(ehs, B, dt):
        ehs = ehs.to(dt).contiguous() if (ehs.dtype != dt or not ehs.is_contiguous()) else ehs
        if ehs.shape[0] == 1 and B > 1:
            ehs = ehs.expand(B, -1, -1).contiguous()
        return ehs

    @staticmethod
    def _tvec(t, B, dev):
        t = torch.as_tensor(t, device=dev, dtype=torch.float32).reshape(-1)
        return t.expand(B) if t.numel() == 1 else t
