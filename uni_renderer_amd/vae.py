"""``AutoencoderKL`` (the SD-1.x VAE) on the same HIP kernels as the denoiser.

The reference encodes 8 images per training step (``vae.encode(x).latent_dist.sample() * vae.config.scaling_factor``,
train/train.py:1266-1304), 2 per sampling call (models/pipeline.py:2533-2538) and decodes 5 per inference
(``vae.decode(latents / scaling_factor, return_dict=False)[0]``, 2755-2769) with diffusers' ``AutoencoderKL``
(train.py:40, 953).  This module is that network -- same class name, ``encode`` / ``decode`` surface, ``config``,
diffusers ``state_dict`` key names (an SD-1.x ``vae/`` folder loads) -- with every layer running on ``ur_igemm`` (3x3
convs incl. the encoder's asymmetric-pad stride-2 downsample and the decoder's fused nearest-2x upsample, 1x1 convs,
the attention GEMMs), ``ur_groupnorm*`` and ``ur_softmax_rows``.  The single-head attention of the mid blocks has head
dim = 512 channels, outside the flash kernel's head dims, so it runs GEMM -> row softmax -> GEMM (the score matrix is
T x T = 32 MB per sample at 512x512, nothing next to 288 GB).  NHWC throughout; NCHW only at the two ends.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check
from .layers import Conv2d, GroupNorm, Linear, PackCache, f32, pack_conv3x3, pack_matrix
from .modeling_utils import ConfigModelMixin, register_to_config

CIN_PAD = 64


class _Resnet(nn.Module):
    """diffusers ResnetBlock2D with ``temb_channels=None`` (no time embedding), eps 1e-6."""

    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.groups, self.eps = groups, eps
        self.norm1 = GroupNorm(groups, cin, eps=eps)
        self.conv1 = Conv2d(cin, cout, 3, padding=1)
        self.norm2 = GroupNorm(groups, cout, eps=eps)
        self.conv2 = Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv2d(cin, cout, 1) if cin != cout else None
        self._pk = PackCache()

    def forward(self, x):
        dt, pk = x.dtype, self._pk
        g1, b1 = pk.get("n1", [self.norm1.weight, self.norm1.bias], dt, lambda: (f32(self.norm1.weight), f32(self.norm1.bias)))
        g2, b2 = pk.get("n2", [self.norm2.weight, self.norm2.bias], dt, lambda: (f32(self.norm2.weight), f32(self.norm2.bias)))
        w1 = pk.get("w1", [self.conv1.weight], dt, lambda: pack_conv3x3(self.conv1.weight, dt))
        c1 = pk.get("c1", [self.conv1.bias], dt, lambda: f32(self.conv1.bias))
        h = ops.groupnorm(x, g1, b1, self.eps, groups=self.groups, silu=True)
        h = ops.conv3x3(h, w1, c1)
        h = ops.groupnorm(h, g2, b2, self.eps, groups=self.groups, silu=True)
        if self.conv_shortcut is not None:  # 1x1 shortcut rides in conv2's K loop (ur_igemm_desc.t0)
            w2 = pk.get("w2s", [self.conv2.weight, self.conv_shortcut.weight], dt,
                        lambda: torch.cat([pack_conv3x3(self.conv2.weight, dt), pack_matrix(self.conv_shortcut.weight, dt)], 1).contiguous())
            c2 = pk.get("c2s", [self.conv2.bias, self.conv_shortcut.bias], dt, lambda: f32(self.conv2.bias) + f32(self.conv_shortcut.bias))
            return ops.conv3x3(h, w2, c2, tail=(x, None))
        w2 = pk.get("w2", [self.conv2.weight], dt, lambda: pack_conv3x3(self.conv2.weight, dt))
        c2 = pk.get("c2", [self.conv2.bias], dt, lambda: f32(self.conv2.bias))
        return ops.conv3x3(h, w2, c2, res=x)


class _Attention(nn.Module):
    """diffusers ``Attention(heads=1, dim_head=C, bias=True, residual_connection=True, norm_num_groups=32, eps=1e-6)``
    of ``UNetMidBlock2D``: GN -> q, k, v (Linear + bias) -> softmax(q k^T / sqrt(C)) v -> to_out -> + input."""

    def __init__(self, c, groups, eps=1e-6):
        super().__init__()
        self.c, self.groups, self.eps = c, groups, eps
        self.group_norm = GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = Linear(c, c), Linear(c, c), Linear(c, c)
        self.to_out = nn.ModuleList([Linear(c, c), nn.Dropout(0.0)])
        self._pk = PackCache()

    def forward(self, x):
        dt, pk, C = x.dtype, self._pk, self.c
        B, H, W, _ = x.shape
        T = H * W
        Tp = (T + 63) // 64 * 64
        g, b = pk.get("gn", [self.group_norm.weight, self.group_norm.bias], dt, lambda: (f32(self.group_norm.weight), f32(self.group_norm.bias)))
        wqk = pk.get("wqk", [self.to_q.weight, self.to_k.weight], dt,
                     lambda: torch.cat([pack_matrix(self.to_q.weight, dt), pack_matrix(self.to_k.weight, dt)], 0).contiguous())
        bqk = pk.get("bqk", [self.to_q.bias, self.to_k.bias], dt, lambda: torch.cat([f32(self.to_q.bias), f32(self.to_k.bias)]))
        wv = pk.get("wv", [self.to_v.weight], dt, lambda: pack_matrix(self.to_v.weight, dt))
        bv = pk.get("bv", [self.to_v.bias], dt, lambda: f32(self.to_v.bias))
        wo = pk.get("wo", [self.to_out[0].weight], dt, lambda: pack_matrix(self.to_out[0].weight, dt))
        bo = pk.get("bo", [self.to_out[0].bias], dt, lambda: f32(self.to_out[0].bias))
        xn = ops.groupnorm(x, g, b, self.eps, groups=self.groups, silu=False).view(B, T, C)
        qk = ops.linear(xn, wqk, bqk)                      # [B, T, 2C] = q | k
        vt = ops.vt_proj(xn, wv)                           # [B, C, Tp] = (x Wv^T)^T without the bias (added below)
        # scores: one GEMM per sample (z = B), q rows x k rows, scaled in the epilogue; rows padded to Tp columns
        P = torch.empty(B, T, Tp, dtype=dt, device=x.device)
        ops.igemm(x0=qk, w=qk[..., C:], out=P, M=T, N=T, K=C, c0=C, ldx0=2 * C, ldw=2 * C, ldc=Tp, n_store=Tp,
                  out_scale=C ** -0.5, zbatch=B, zx=T * 2 * C, zw=T * 2 * C, zout=T * Tp)
        check(_lib.load().ur_softmax_rows(P.data_ptr(), Tp, B * T, T, ops.DT[dt], ops._stream()), "ur_softmax_rows")
        # o = P v: rows of P sum to one, so the value bias is a plain column bias of this GEMM
        o = torch.empty(B, T, C, dtype=dt, device=x.device)
        ops.igemm(x0=P, w=vt, out=o, M=T, N=C, K=Tp, c0=Tp, ldx0=Tp, ldw=vt.stride(1), ldc=C, bias=bv, zbatch=B,
                  zx=T * Tp, zw=vt.stride(0), zout=T * C)
        return ops.linear(o, wo, bo, res=x.view(B, T, C)).view(B, H, W, C)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])
        self.attentions = nn.ModuleList([_Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Down(nn.Module):
    """encoder Downsample2D: F.pad(x, (0, 1, 0, 1)) + conv3x3 stride 2 padding 0 = ``ur_igemm`` with ``pad = 0``."""

    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, stride=2, padding=0)
        self._pk = PackCache()

    def forward(self, x):
        dt = x.dtype
        w = self._pk.get("w", [self.conv.weight], dt, lambda: pack_conv3x3(self.conv.weight, dt))
        b = self._pk.get("b", [self.conv.bias], dt, lambda: f32(self.conv.bias))
        return ops.conv3x3(x, w, b, stride=2, pad=0)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, padding=1)
        self._pk = PackCache()

    def forward(self, x):
        dt = x.dtype
        w = self._pk.get("w", [self.conv.weight], dt, lambda: pack_conv3x3(self.conv.weight, dt))
        b = self._pk.get("b", [self.conv.bias], dt, lambda: f32(self.conv.bias))
        return ops.conv3x3(x, w, b, ups=True)  # nearest-2x fused into the gather


class _Level(nn.Module):
    def __init__(self, cin, cout, n, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([_Down(cout)]) if down else None
        self.upsamplers = nn.ModuleList([_Up(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(_Level(c, co, layers, groups, down=i != len(boc) - 1))
            c = co
        self.mid_block = _Mid(c, groups)
        self.conv_norm_out = GroupNorm(groups, c, eps=1e-6)
        self.conv_out = Conv2d(c, 2 * latent_channels, 3, padding=1)


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, boc, layers, groups):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_Level(c, co, layers + 1, groups, up=i != len(boc) - 1))
            c = co
        self.conv_norm_out = GroupNorm(groups, c, eps=1e-6)
        self.conv_out = Conv2d(c, out_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """diffusers' posterior object: ``parameters`` [B, 2C, h, w] = mean | logvar (clamped to [-30, 20])."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class AutoencoderKL(ConfigModelMixin, nn.Module):
    @register_to_config
    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 down_block_types: Tuple[str, ...] = ("DownEncoderBlock2D",) * 4,
                 up_block_types: Tuple[str, ...] = ("UpDecoderBlock2D",) * 4,
                 block_out_channels: Tuple[int, ...] = (128, 256, 512, 512), layers_per_block: int = 2, act_fn: str = "silu",
                 latent_channels: int = 4, norm_num_groups: int = 32, sample_size: int = 512,
                 scaling_factor: float = 0.18215, force_upcast: bool = True):
        super().__init__()
        boc = tuple(block_out_channels)
        if (act_fn != "silu" or any(t != "DownEncoderBlock2D" for t in down_block_types[:len(boc)])
                or any(t != "UpDecoderBlock2D" for t in up_block_types[:len(boc)])):
            raise NotImplementedError("AutoencoderKL: only the SD-1.x encoder / decoder block types are implemented")
        if any(c % 64 for c in boc):
            raise NotImplementedError("AutoencoderKL: block_out_channels must be multiples of 64 (ur_igemm K granularity)")
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(out_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1)
        self.compute_dtype: Optional[torch.dtype] = None
        self._pk = PackCache()

    _DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}

    @classmethod
    def _convert_state_dict(cls, sd):
        """The ``vae/`` folders of SD 1.4 (the base the reference trains from), 1.5 and 2.x store the mid-block
        attention under the deprecated names ``query / key / value / proj_attn``; diffusers renames them while loading
        (``ModelMixin._convert_deprecated_attention_blocks``).  Same conversion here, plus the 1x1-conv weight shape
        ``[C, C, 1, 1]`` of still older exports."""
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if len(parts) >= 4 and parts[-2] in cls._DEPRECATED_ATTN and "attentions" in parts:
                k = ".".join(parts[:-2] + [cls._DEPRECATED_ATTN[parts[-2]], parts[-1]])
                if parts[-1] == "weight" and v.dim() == 4 and v.shape[2:] == (1, 1):
                    v = v[:, :, 0, 0].contiguous()
            out[k] = v
        return out

    def _dt(self):
        dt = self.compute_dtype or self.dtype
        if dt not in (torch.float16, torch.bfloat16):
            if torch.is_autocast_enabled():
                return torch.get_autocast_dtype("cuda")
            raise RuntimeError("the MI355X path computes in fp16/bf16: cast the VAE (.to(torch.float16)), run under "
                               "torch.autocast, or set vae.compute_dtype")
        return dt

    # ---- encode (train.py:1266-1304; pipeline.py:2533-2538) ---------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        dt, pk, e = self._dt(), self._pk, self.encoder
        w = pk.get("e.in", [e.conv_in.weight], dt, lambda: pack_conv3x3(e.conv_in.weight, dt, CIN_PAD))
        b = pk.get("e.inb", [e.conv_in.bias], dt, lambda: f32(e.conv_in.bias))
        h = ops.conv3x3(ops.to_nhwc(x, dt, CIN_PAD), w, b)
        for lvl in e.down_blocks:
            h = lvl(h)
        h = e.mid_block(h)
        g, gb = pk.get("e.no", [e.conv_norm_out.weight, e.conv_norm_out.bias], dt, lambda: (f32(e.conv_norm_out.weight), f32(e.conv_norm_out.bias)))
        h = ops.groupnorm(h, g, gb, e.conv_norm_out.eps, groups=e.conv_norm_out.num_groups, silu=True)
        # quant_conv (1x1, per pixel) composed with conv_out on the host in fp32: quant(conv(h)) = (Wq Wc) * h + (Wq bc + bq)
        q = self.quant_conv

        def compose():
            wq = q.weight.detach().float().reshape(q.weight.shape[0], -1)
            wc = torch.einsum("oi,icyx->ocyx", wq, e.conv_out.weight.detach().float())
            bc = wq @ e.conv_out.bias.detach().float() + q.bias.detach().float()
            return pack_conv3x3(wc, dt), bc.contiguous()

        wc, bc = pk.get("e.out", [e.conv_out.weight, e.conv_out.bias, q.weight, q.bias], dt, compose)
        moments = ops.to_nchw(ops.conv3x3(h, wc, bc, n_out=q.weight.shape[0]), x.dtype if x.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32)
        post = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)

    # ---- decode (pipeline.py:2755-2769) -----------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        dt, pk, d = self._dt(), self._pk, self.decoder
        pq = self.post_quant_conv
        B, Cz, H, W = z.shape
        # post_quant_conv: 1x1 over the (zero padded to 64) latent channels, output again 64 wide for conv_in
        wp = pk.get("d.pq", [pq.weight], dt, lambda: torch.nn.functional.pad(pack_matrix(pq.weight, dt), (0, CIN_PAD - Cz)).contiguous())
        bp = pk.get("d.pqb", [pq.bias], dt, lambda: f32(pq.bias))
        zin = ops.to_nhwc(z, dt, CIN_PAD)
        zq = torch.empty(B, H, W, CIN_PAD, dtype=dt, device=z.device)
        ops.igemm(x0=zin, w=wp, out=zq, M=B * H * W, N=pq.weight.shape[0], K=CIN_PAD, c0=CIN_PAD, ldx0=CIN_PAD, ldw=CIN_PAD,
                  ldc=CIN_PAD, n_store=CIN_PAD, bias=bp)
        w = pk.get("d.in", [d.conv_in.weight], dt, lambda: pack_conv3x3(d.conv_in.weight, dt, CIN_PAD))
        b = pk.get("d.inb", [d.conv_in.bias], dt, lambda: f32(d.conv_in.bias))
        h = d.mid_block(ops.conv3x3(zq, w, b))
        for lvl in d.up_blocks:
            h = lvl(h)
        g, gb = pk.get("d.no", [d.conv_norm_out.weight, d.conv_norm_out.bias], dt, lambda: (f32(d.conv_norm_out.weight), f32(d.conv_norm_out.bias)))
        h = ops.groupnorm(h, g, gb, d.conv_norm_out.eps, groups=d.conv_norm_out.num_groups, silu=True)
        wo = pk.get("d.out", [d.conv_out.weight], dt, lambda: pack_conv3x3(d.conv_out.weight, dt))
        bo = pk.get("d.outb", [d.conv_out.bias], dt, lambda: f32(d.conv_out.bias))
        img = ops.to_nchw(ops.conv3x3(h, wo, bo, n_out=d.conv_out.weight.shape[0]), z.dtype if z.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32)
        return DecoderOutput(sample=img) if return_dict else (img,)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True, generator=None):
        post = self.encode(sample).latent_dist
        z = post.sample(generator) if sample_posterior else post.mode()
        return self.decode(z, return_dict=return_dict)
