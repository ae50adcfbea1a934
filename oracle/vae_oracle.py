"""CPU fp32 ORACLE for the VAE (``AutoencoderKL``) on either side of the denoise loop.  TEST INFRASTRUCTURE ONLY
(same rules as unirenderer_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

PARITY UNPINNED.  The reference loads ``AutoencoderKL`` from ``diffusers==0.24.0.dev0`` (train/train.py:40, 953;
models/pipeline.py:10-27), which is not vendored and not installable here, so this file RESTATES the published SD-1.x
KL autoencoder from torch primitives: call sites ``vae.encode(x).latent_dist.sample() * vae.config.scaling_factor``
(train/train.py:1266-1304; models/pipeline.py:2533-2538) and ``vae.decode(latents / scaling_factor)``
(models/pipeline.py:2755-2769).  Structure (diffusers ``Encoder`` / ``Decoder`` / ``UNetMidBlock2D`` /
``DownEncoderBlock2D`` / ``UpDecoderBlock2D``, state-dict key names kept so an SD-1.x ``vae/`` checkpoint loads):

  encoder: conv_in 3->C0 | per level: ``layers_per_block`` ResnetBlock2D (no time embedding, GN eps 1e-6) then, except
           at the last level, Downsample2D = F.pad(x, (0,1,0,1)) + conv3x3 stride 2 padding 0 | mid: resnet, single-head
           attention (GN(32) -> q,k,v Linear with bias -> softmax(QK^T / sqrt(C)) V -> to_out -> + residual), resnet |
           GN + SiLU + conv_out -> 2 * latent_channels | quant_conv 1x1 -> (mean, logvar)
  decoder: post_quant_conv 1x1 | conv_in -> C_last | mid (same) | per level (reversed): ``layers_per_block + 1`` resnets
           then, except at the last, nearest-2x + conv3x3 | GN + SiLU + conv_out -> 3

Pinned by (tests/test_vae_cpu.py): the parameter count of the SD-1.x VAE (83,653,863), output shapes, op-level
identities (attention == F.scaled_dot_product_attention with one head, downsample == explicit zero-pad + conv) and
diffusers state-dict key names.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                     layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(64, 128),
                       layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers ``Attention(heads=1, dim_head=C, bias=True, residual_connection=True, norm_num_groups=32)``."""

    def __init__(self, c, groups, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax((q @ k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])
        self.attentions = nn.ModuleList([Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Level(nn.Module):
    def __init__(self, cin, cout, n, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

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
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(_Level(c, co, layers, groups, down=i != len(boc) - 1))
            c = co
        self.mid_block = MidBlock(c, groups)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, boc, layers, groups):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], groups)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_Level(c, co, layers + 1, groups, up=i != len(boc) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        self.cfg = dict(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                        block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                        norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @torch.no_grad()
    def encode_moments(self, x):
        """(mean, logvar clamped to [-30, 20]) of the posterior (DiagonalGaussianDistribution)."""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def build(config: dict, seed: int = 7):
    torch.manual_seed(seed)
    return AutoencoderKL(**config).eval()
