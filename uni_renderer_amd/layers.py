"""L0 leaves of the hot path, MI355X-native.

The reference imports these from diffusers 0.24 (models/unet_2d_blocks.py:24-28, models/controlnet.py:24-34):
``ResnetBlock2D``, ``Transformer2DModel`` (-> ``BasicTransformerBlock`` -> ``Attention`` / ``FeedForward``),
``Downsample2D``, ``Upsample2D``, ``Timesteps``, ``TimestepEmbedding``.  Here each class is

  * a parameter holder whose ``state_dict`` keys and shapes are the diffusers ones (so SD-1.x /
    Uni-Renderer checkpoints load unchanged), and
  * a ``forward`` over NHWC tensors that only enqueues HIP kernels through ``ops`` (C ABI).  There is no
    PyTorch arithmetic on this path; parameter holders raise if someone calls their torch ``forward``.

Weights are re-laid-out ("packed") once per (dtype, parameter version) into the layouts the kernels read:
conv3x3 -> [Cout, (ky,kx,Cin)], q|k fused, GEGLU value/gate interleaved in groups of 4 columns.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import ops

_HOLDER_MSG = ("this nn.Module only holds parameters for the HIP path; its torch forward is disabled "
               "(no PyTorch fallback in uni_renderer_amd)")


class Conv2d(nn.Conv2d):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(_HOLDER_MSG)


class Linear(nn.Linear):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(_HOLDER_MSG)


class GroupNorm(nn.GroupNorm):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(_HOLDER_MSG)


class LayerNorm(nn.LayerNorm):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(_HOLDER_MSG)


def _ceil(v: int, m: int) -> int:
    return (v + m - 1) // m * m


LOG2E = 1.4426950408889634


class PackCache:
    """Per-module cache of packed tensors keyed by (name, dtype, device, parameter versions)."""

    def __init__(self):
        self._store = {}

    def get(self, name, params, dtype, build):
        key = (name, dtype)
        ver = tuple((p.data_ptr(), p._version, p.device) for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = build()
        self._store[key] = (ver, val)
        return val


def pack_conv3x3(weight: torch.Tensor, dtype, cin_pad: Optional[int] = None, cblock: int = 0) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> [Co, 9 * Ci_pad] with k = (ky*3 + kx) * Ci_pad + c (zero padded channels), or with
    ``cblock`` > 0 in the block-outer order k = (c // cblock) * 9 * cblock + (ky*3 + kx) * cblock + c % cblock
    (``ops.conv_cblock``; the launch must pass the same ``cblock``)."""
    co, ci = weight.shape[:2]
    cp = ci if cin_pad is None else cin_pad
    w = weight.detach().permute(0, 2, 3, 1)  # [Co, ky, kx, Ci]
    if cp != ci:
        w = torch.nn.functional.pad(w, (0, cp - ci))
    if cblock:
        w = w.reshape(co, 9, cp // cblock, cblock).permute(0, 2, 1, 3)  # [Co, block, tap, c]
    return w.reshape(co, 9 * cp).to(dtype).contiguous()


def pack_matrix(weight: torch.Tensor, dtype) -> torch.Tensor:
    """Linear [N, K] or 1x1 conv [N, K, 1, 1] -> [N, K]."""
    return weight.detach().reshape(weight.shape[0], -1).to(dtype).contiguous()


def geglu_perm(n_half: int, device) -> torch.Tensor:
    """Row order the GEGLU epilogue expects: packed row p -> value row (p//8)*4 + p%4 if p%8 < 4 else the
    matching gate row (include/ur_kernels.h, UR_ACT_GEGLU)."""
    p = torch.arange(2 * n_half, device=device)
    col = (p // 8) * 4 + (p % 4)
    return torch.where((p % 8) < 4, col, n_half + col)


def f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


class Ctx:
    """Per-forward context handed down the module tree."""

    __slots__ = ("dtype", "temb", "ehs", "B", "kc", "vtc")

    def __init__(self, dtype, B):
        self.dtype = dtype
        self.B = B
        self.temb = None  # [B, sum(Cout of all resnets)] = time_emb_proj(silu(emb)) of every resnet, batched
        self.ehs = None   # [B, 77, cross_dim] tokens in compute dtype
        self.kc = None    # [B, 77, sum(C)]   to_k(ehs) of every cross-attention of the network, one GEMM
        self.vtc = None   # [B, sum(C), 128]  to_v(ehs)^T of every cross-attention, one GEMM


# ---------------------------------------------------------------------------------------------------
class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding (ctor controlnet.py:289-295): linear_1 -> SiLU -> linear_2."""

    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)
        self._pk = PackCache()

    def forward(self, t_emb: torch.Tensor, silu_out: bool = True) -> torch.Tensor:
        """Returns SiLU(emb) when ``silu_out`` (every consumer of emb in this config is a resnet's
        ``time_emb_proj(SiLU(emb))``), computed in the GEMM epilogues."""
        dt = t_emb.dtype
        w1 = self._pk.get("w1", [self.linear_1.weight], dt, lambda: pack_matrix(self.linear_1.weight, dt))
        b1 = self._pk.get("b1", [self.linear_1.bias], dt, lambda: f32(self.linear_1.bias))
        w2 = self._pk.get("w2", [self.linear_2.weight], dt, lambda: pack_matrix(self.linear_2.weight, dt))
        b2 = self._pk.get("b2", [self.linear_2.bias], dt, lambda: f32(self.linear_2.bias))
        h = ops.linear(t_emb, w1, b1, act=ops.ACT_SILU)
        return ops.linear(h, w2, b2, act=ops.ACT_SILU if silu_out else ops.ACT_NONE)


class ResnetBlock2D(nn.Module):
    """GN+SiLU -> conv3x3 (+bias +temb) -> GN+SiLU -> conv3x3 (+bias, + shortcut(x), / output_scale_factor).
    ctor args as at unet_2d_blocks.py:1100-1111; time_embedding_norm="default", dropout 0."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.groups, self.eps, self.output_scale_factor = groups, eps, output_scale_factor
        self.norm1 = GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels)
        self.norm2 = GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.temb_slice = None  # (offset, offset + out_channels) into Ctx.temb, set by the owning network
        self._pk = PackCache()

    def forward(self, x, ctx: Ctx, x1=None, extra_res=None):
        """x (and optional x1, concatenated on channels) NHWC -> NHWC [B,H,W,out_channels]."""
        dt = x.dtype
        pk = self._pk
        g1, b1 = pk.get("n1", [self.norm1.weight, self.norm1.bias], dt, lambda: (f32(self.norm1.weight), f32(self.norm1.bias)))
        g2, b2 = pk.get("n2", [self.norm2.weight, self.norm2.bias], dt, lambda: (f32(self.norm2.weight), f32(self.norm2.bias)))
        w1 = pk.get("w1", [self.conv1.weight], dt,
                    lambda: pack_conv3x3(self.conv1.weight, dt, cblock=ops.conv_cblock(self.conv1.weight.shape[1])))
        cb1 = pk.get("cb1", [self.conv1.bias], dt, lambda: f32(self.conv1.bias))
        w2 = pk.get("w2", [self.conv2.weight], dt,
                    lambda: pack_conv3x3(self.conv2.weight, dt, cblock=ops.conv_cblock(self.conv2.weight.shape[1])))
        cb2 = pk.get("cb2", [self.conv2.bias], dt, lambda: f32(self.conv2.bias))
        lo, hi = self.temb_slice
        h = ops.groupnorm(x, g1, b1, self.eps, x1=x1, groups=self.groups, silu=True)
        h = ops.conv3x3(h, w1, cb1, rowadd=ctx.temb[:, lo:hi], cblock=ops.conv_cblock(h.shape[-1]))
        h = ops.groupnorm(h, g2, b2, self.eps, groups=self.groups, silu=True)
        if self.conv_shortcut is not None and ops.FOLD_SHORTCUT and extra_res is None:
            # conv_shortcut rides in conv2's K loop as its 1x1 tail (ops.conv3x3 ``tail``)
            w2s = pk.get("w2s", [self.conv2.weight, self.conv_shortcut.weight], dt,
                         lambda: torch.cat([pack_conv3x3(self.conv2.weight, dt, cblock=ops.conv_cblock(self.conv2.weight.shape[1])),
                                            pack_matrix(self.conv_shortcut.weight, dt)], 1).contiguous())
            cb2s = pk.get("cb2s", [self.conv2.bias, self.conv_shortcut.bias], dt,
                          lambda: f32(self.conv2.bias) + f32(self.conv_shortcut.bias))
            return ops.conv3x3(h, w2s, cb2s, tail=(x, x1), out_scale=1.0 / self.output_scale_factor, hilo=ops.PRECISE_RESIDUAL,
                               cblock=ops.conv_cblock(h.shape[-1]))
        if self.conv_shortcut is not None:
            ws = pk.get("ws", [self.conv_shortcut.weight], dt, lambda: pack_matrix(self.conv_shortcut.weight, dt))
            bs = pk.get("bs", [self.conv_shortcut.bias], dt, lambda: f32(self.conv_shortcut.bias))
            sc = ops.linear(x, ws, bs, x1=x1)
        else:
            if x1 is not None:
                raise RuntimeError("concat input requires a conv_shortcut (in_channels != out_channels)")
            sc = x
        return ops.conv3x3(h, w2, cb2, res=sc, out_scale=1.0 / self.output_scale_factor, hilo=ops.PRECISE_RESIDUAL,
                           cblock=ops.conv_cblock(h.shape[-1]))


class Attention(nn.Module):
    """diffusers Attention / AttnProcessor2_0: q,k,v without bias; softmax(QK^T/sqrt(d))V; out proj + bias."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.is_cross = cross_attention_dim is not None
        kv = cross_attention_dim if self.is_cross else query_dim
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(kv, inner, bias=False)
        self.to_v = Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([Linear(inner, query_dim), nn.Dropout(0.0)])
        self.kv_slice = None  # rows of the network-wide batched K / V^T projection (cross-attention only)
        self._pk = PackCache()

    def forward(self, xn, ctx: Ctx, residual):
        """xn: normalised tokens [B,T,C]; returns residual + to_out(attention)."""
        dt = xn.dtype
        pk = self._pk
        B, T, _ = xn.shape
        H, d, C = self.heads, self.dim_head, self.inner
        # The softmax scale and the change to base 2 are folded into the q (and, when q|k come from one GEMM, k)
        # projection epilogues in fp32 -- before the single rounding to fp16/bf16 -- so the attention kernel sees
        # scores in log2 units (ur_attention scale = 0) and spends no multiply-add per score.
        cs = d ** -0.5 * LOG2E
        wo = pk.get("wo", [self.to_out[0].weight], dt, lambda: pack_matrix(self.to_out[0].weight, dt))
        bo = pk.get("bo", [self.to_out[0].bias], dt, lambda: f32(self.to_out[0].bias))
        if self.is_cross and ctx.kc is not None and self.kv_slice is not None:
            # K / V^T of the prompt were projected once for the whole network (controlnet._begin)
            lo, hi = self.kv_slice
            wq = pk.get("wq", [self.to_q.weight], dt, lambda: pack_matrix(self.to_q.weight, dt))
            q = ops.linear(xn, wq, out_scale=cs)
            o = ops.attention(q, ctx.kc[:, :, lo:hi], ctx.vtc[:, lo:hi], B=B, H=H, Tq=T, Tk=ctx.kc.shape[1], d=d,
                              ldq=C, ldk=ctx.kc.stride(1), scale=0.0)
            return ops.linear(o, wo, bo, res=residual, hilo=ops.PRECISE_RESIDUAL)
        wv = pk.get("wv", [self.to_v.weight], dt, lambda: pack_matrix(self.to_v.weight, dt))
        if not self.is_cross:
            wqk = pk.get("wqk", [self.to_q.weight, self.to_k.weight], dt,
                         lambda: torch.cat([pack_matrix(self.to_q.weight, dt), pack_matrix(self.to_k.weight, dt)], 0))
            qk = ops.linear(xn, wqk, out_scale=math.sqrt(cs))  # [B,T,2C] = q | k, each carrying sqrt(cs)
            vt = ops.vt_proj(xn, wv)                       # [B,C,Tpad]
            o = ops.attention(qk, qk, vt, B=B, H=H, Tq=T, Tk=T, d=d, ldq=2 * C, ldk=2 * C, q_off=0, k_off=C, scale=0.0)
        else:
            wq = pk.get("wq", [self.to_q.weight], dt, lambda: pack_matrix(self.to_q.weight, dt))
            wk = pk.get("wk", [self.to_k.weight], dt, lambda: pack_matrix(self.to_k.weight, dt))
            ehs = ctx.ehs
            Tk = ehs.shape[1]
            q = ops.linear(xn, wq, out_scale=cs)
            k = ops.linear(ehs, wk)                        # [B,Tk,C]
            vt = ops.vt_proj(ehs, wv)                      # [B,C,ceil64(Tk)]
            o = ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, scale=0.0)
        return ops.linear(o, wo, bo, res=residual, hilo=ops.PRECISE_RESIDUAL)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """net.0.proj (GEGLU, erf GELU) -> net.2; keys ``ff.net.0.proj.*`` / ``ff.net.2.*`` as in diffusers."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), Linear(dim * mult, dim)])
        self._pk = PackCache()

    def forward(self, xn, residual):
        dt = xn.dtype
        pk = self._pk
        proj, out = self.net[0].proj, self.net[2]
        nh = proj.weight.shape[0] // 2

        def build_in():
            perm = geglu_perm(nh, proj.weight.device)
            return pack_matrix(proj.weight, dt)[perm].contiguous(), f32(proj.bias)[perm].contiguous()

        w_in, b_in = pk.get("in", [proj.weight, proj.bias], dt, build_in)
        w_out = pk.get("wout", [out.weight], dt, lambda: pack_matrix(out.weight, dt))
        b_out = pk.get("bout", [out.bias], dt, lambda: f32(out.bias))
        g = ops.linear(xn, w_in, b_in, act=ops.ACT_GEGLU)   # [B,T,4C]
        return ops.linear(g, w_out, b_out, res=residual, hilo=ops.PRECISE_RESIDUAL)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim=cross_attention_dim)
        self.norm3 = LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)
        self._pk = PackCache()

    def _ln(self, name, ln, dt):
        return self._pk.get(name, [ln.weight, ln.bias], dt, lambda: (f32(ln.weight), f32(ln.bias)))

    def forward(self, x, ctx: Ctx):
        dt = x.dtype
        g, b = self._ln("n1", self.norm1, dt)
        x = self.attn1(ops.layernorm(x, g, b, self.norm1.eps), ctx, x)
        g, b = self._ln("n2", self.norm2, dt)
        x = self.attn2(ops.layernorm(x, g, b, self.norm2.eps), ctx, x)
        g, b = self._ln("n3", self.norm3, dt)
        return self.ff(ops.layernorm(x, g, b, self.norm3.eps), x)


class Transformer2DModel(nn.Module):
    """GN(32, eps 1e-6) -> 1x1 proj_in -> transformer block(s) -> 1x1 proj_out -> + input
    (ctor unet_2d_blocks.py:1115-1126; use_linear_projection=False).  NHWC makes the NCHW<->token
    permutes of the reference disappear: [B,H,W,C] *is* [B,HW,C]."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32, num_layers=1):
        super().__init__()
        inner = heads * dim_head
        self.groups = norm_num_groups
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = Conv2d(inner, in_channels, 1)
        self._pk = PackCache()

    def forward(self, x, ctx: Ctx):
        dt = x.dtype
        pk = self._pk
        B, H, W, Cc = x.shape
        g, b = pk.get("n", [self.norm.weight, self.norm.bias], dt, lambda: (f32(self.norm.weight), f32(self.norm.bias)))
        wi = pk.get("wi", [self.proj_in.weight], dt, lambda: pack_matrix(self.proj_in.weight, dt))
        bi = pk.get("bi", [self.proj_in.bias], dt, lambda: f32(self.proj_in.bias))
        wo = pk.get("wo", [self.proj_out.weight], dt, lambda: pack_matrix(self.proj_out.weight, dt))
        bo = pk.get("bo", [self.proj_out.bias], dt, lambda: f32(self.proj_out.bias))
        h = ops.groupnorm(x, g, b, self.norm.eps, groups=self.groups, silu=False)
        h = ops.linear(h.view(B, H * W, Cc), wi, bi, hilo=ops.PRECISE_RESIDUAL)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        out = ops.linear(h, wo, bo, res=ops.view_hilo(x, B, H * W, Cc), hilo=ops.PRECISE_RESIDUAL)
        return ops.view_hilo(out, B, H, W, Cc)


class Downsample2D(nn.Module):
    """conv3x3 stride 2 pad 1 (ctor unet_2d_blocks.py:1143-1149); key ``downsamplers.0.conv``."""

    def __init__(self, channels, padding=1):
        super().__init__()
        if padding != 1:
            raise NotImplementedError("downsample_padding != 1")
        self.conv = Conv2d(channels, channels, 3, stride=2, padding=1)
        self._pk = PackCache()

    def forward(self, x):
        dt = x.dtype
        w = self._pk.get("w", [self.conv.weight], dt,
                         lambda: pack_conv3x3(self.conv.weight, dt, cblock=ops.conv_cblock(self.conv.weight.shape[1])))
        b = self._pk.get("b", [self.conv.bias], dt, lambda: f32(self.conv.bias))
        return ops.conv3x3(x, w, b, stride=2, hilo=ops.PRECISE_RESIDUAL, cblock=ops.conv_cblock(x.shape[-1]))


class Upsample2D(nn.Module):
    """nearest x2 (F.interpolate) then conv3x3 (unet_2d_blocks.py:2501), fused into one gather-conv."""

    def __init__(self, channels):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, padding=1)
        self._pk = PackCache()

    def forward(self, x, output_size=None):
        dt = x.dtype
        w = self._pk.get("w", [self.conv.weight], dt,
                         lambda: pack_conv3x3(self.conv.weight, dt, cblock=ops.conv_cblock(self.conv.weight.shape[1])))
        b = self._pk.get("b", [self.conv.bias], dt, lambda: f32(self.conv.bias))
        if output_size is not None and tuple(int(v) for v in output_size) != (2 * x.shape[1], 2 * x.shape[2]):
            # latent side not a multiple of 8 (controlnet.py:869-883, 1129-1130): F.interpolate(size=...), then the conv
            return ops.conv3x3(ops.resize_nearest(x, output_size), w, b, hilo=ops.PRECISE_RESIDUAL, cblock=ops.conv_cblock(x.shape[-1]))
        return ops.conv3x3(x, w, b, ups=True, hilo=ops.PRECISE_RESIDUAL, cblock=ops.conv_cblock(x.shape[-1]))


def zero_module(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        nn.init.zeros_(p)
    return m
