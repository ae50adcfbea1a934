"""L1 block library of the hot path (mirror of the reference's ``models/unet_2d_blocks.py``), NHWC + HIP.

Live blocks: ``CrossAttnDownBlock2D`` (ref 1063-1221), ``DownBlock2D`` (1224-1309),
``UNetMidBlock2DCrossAttn`` (670-813), ``UpBlock2D`` (2593-2704), ``CrossAttnUpBlock2D`` (2417-2590) and the
reference-added ``UpResBlock2D`` (2706-2822) / ``CrossAttnUpResBlock2D`` (2237-2415), plus the string-keyed
factories ``get_down_block`` (34-240) / ``get_up_block`` (243-505).  As in the reference the four Up* blocks
return ``(hidden, per-resnet outputs)``.  The other ~20 diffusers block types the reference file carries
are unreachable with the SD-1.x configuration and raise ``ValueError`` here, like an unknown name does
there (240, 505).

Differences that are purely MI355X-side: tensors are NHWC; ``torch.cat([hidden, skip], dim=1)`` is never
materialised -- the two tensors are handed to the resnet, whose GroupNorm and 1x1-shortcut kernels read
both sources; nearest-2x upsampling is fused into the following conv's gather.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .layers import Ctx, Downsample2D, ResnetBlock2D, Transformer2DModel, Upsample2D


class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, add_downsample=True, downsample_padding=1,
                 transformer_layers_per_block=1, output_scale_factor=1.0):
        super().__init__()
        self.num_attention_heads = num_attention_heads
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups,
                          resnet_eps, output_scale_factor) for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                               cross_attention_dim, resnet_groups, transformer_layers_per_block)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, downsample_padding)]) if add_downsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, ctx: Ctx, additional_residuals=None):
        output_states = ()
        n = len(self.resnets)
        for i, (resnet, attn) in enumerate(zip(self.resnets, self.attentions)):
            hidden_states = attn(resnet(hidden_states, ctx), ctx)
            if i == n - 1 and additional_residuals is not None:  # T2I-adapter hook (ref 1209-1211)
                hidden_states = ops.add(hidden_states, additional_residuals)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_downsample=True, downsample_padding=1, output_scale_factor=1.0):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups,
                          resnet_eps, output_scale_factor) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, downsample_padding)]) if add_downsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, ctx: Ctx):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, ctx)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, output_scale_factor=1.0,
                 transformer_layers_per_block=1):
        super().__init__()
        self.num_attention_heads = num_attention_heads
        resnets = [ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps, output_scale_factor)]
        attentions = []
        for _ in range(num_layers):
            attentions.append(Transformer2DModel(num_attention_heads, in_channels // num_attention_heads, in_channels,
                                                 cross_attention_dim, resnet_groups, transformer_layers_per_block))
            resnets.append(ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps,
                                         output_scale_factor))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.gradient_checkpointing = False

    def forward(self, hidden_states, ctx: Ctx):
        hidden_states = self.resnets[0](hidden_states, ctx)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = resnet(attn(hidden_states, ctx), ctx)
        return hidden_states


def _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, groups, eps, osf):
    rs = []
    for i in range(num_layers):
        res_skip = in_channels if i == num_layers - 1 else out_channels
        rin = prev_output_channel if i == 0 else out_channels
        r = ResnetBlock2D(rin + res_skip, out_channels, temb_channels, groups, eps, osf)
        r.split = (rin, res_skip)  # channels of (hidden, skip): the concat is never materialised
        rs.append(r)
    return nn.ModuleList(rs)


class UpBlock2D(nn.Module):
    """[cat(hidden, skip.pop()) -> resnet] x L (+ upsample); returns (hidden, per-resnet outputs)."""

    has_cross_attention = False
    adds_up_states = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, resolution_idx=None, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, add_upsample=True, output_scale_factor=1.0):
        super().__init__()
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_groups, resnet_eps, output_scale_factor)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self.resolution_idx = resolution_idx
        self.gradient_checkpointing = False

    def forward(self, hidden_states, res_hidden_states_tuple, ctx: Ctx, upsample_size=None,
                up_additional_states_tuple=None):
        output_states = ()
        for k, resnet in enumerate(self.resnets):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet(hidden_states, ctx, x1=res)
            if self.adds_up_states and up_additional_states_tuple is not None:  # UpResBlock2D (ref 2814)
                hidden_states = ops.add(hidden_states, up_additional_states_tuple[k])
            output_states += (hidden_states,)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states, output_states


class UpResBlock2D(UpBlock2D):
    adds_up_states = True


class CrossAttnUpBlock2D(nn.Module):
    """[cat -> resnet -> transformer] x L (+ upsample); returns (hidden, per-layer outputs)."""

    has_cross_attention = True
    adds_up_states = False

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, resolution_idx=None, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, num_attention_heads=1, cross_attention_dim=1280,
                 add_upsample=True, output_scale_factor=1.0, transformer_layers_per_block=1):
        super().__init__()
        self.num_attention_heads = num_attention_heads
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_groups, resnet_eps, output_scale_factor)
        self.attentions = nn.ModuleList([
            Transformer2DModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                               cross_attention_dim, resnet_groups, transformer_layers_per_block)
            for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self.resolution_idx = resolution_idx
        self.gradient_checkpointing = False

    def forward(self, hidden_states, res_hidden_states_tuple, ctx: Ctx, upsample_size=None,
                up_additional_states_tuple=None):
        output_states = ()
        for k, (resnet, attn) in enumerate(zip(self.resnets, self.attentions)):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = attn(resnet(hidden_states, ctx, x1=res), ctx)
            if self.adds_up_states and up_additional_states_tuple is not None:  # CrossAttnUpResBlock2D (ref 2408)
                hidden_states = ops.add(hidden_states, up_additional_states_tuple[k])
            output_states += (hidden_states,)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states, output_states


class CrossAttnUpResBlock2D(CrossAttnUpBlock2D):
    adds_up_states = True


def _strip(name: str) -> str:
    return name[7:] if name.startswith("UNetRes") else name  # ref 68, 278


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn="silu", transformer_layers_per_block=1, num_attention_heads=None, resnet_groups=None,
                   cross_attention_dim=None, downsample_padding=None, attention_head_dim=None, **unused):
    if resnet_act_fn not in ("silu", "swish"):
        raise NotImplementedError(f"act_fn {resnet_act_fn}")
    if attention_head_dim is None:
        attention_head_dim = num_attention_heads  # ref 62-66
    kind = _strip(down_block_type)
    if kind == "DownBlock2D":
        return DownBlock2D(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                           add_downsample, downsample_padding)
    if kind == "CrossAttnDownBlock2D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock2D")
        return CrossAttnDownBlock2D(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                                    num_attention_heads, cross_attention_dim, add_downsample, downsample_padding,
                                    transformer_layers_per_block)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_act_fn="silu", resolution_idx=None, transformer_layers_per_block=1,
                 num_attention_heads=None, resnet_groups=None, cross_attention_dim=None, attention_head_dim=None,
                 **unused):
    if resnet_act_fn not in ("silu", "swish"):
        raise NotImplementedError(f"act_fn {resnet_act_fn}")
    kind = _strip(up_block_type)
    plain = {"UpBlock2D": UpBlock2D, "UpResBlock2D": UpResBlock2D}
    cross = {"CrossAttnUpBlock2D": CrossAttnUpBlock2D, "CrossAttnUpResBlock2D": CrossAttnUpResBlock2D}
    if kind in plain:
        return plain[kind](in_channels, prev_output_channel, out_channels, temb_channels, resolution_idx, num_layers,
                           resnet_eps, resnet_groups, add_upsample)
    if kind in cross:
        if cross_attention_dim is None:
            raise ValueError(f"cross_attention_dim must be specified for {kind}")
        return cross[kind](in_channels, out_channels, prev_output_channel, temb_channels, resolution_idx, num_layers,
                           resnet_eps, resnet_groups, num_attention_heads, cross_attention_dim, add_upsample, 1.0,
                           transformer_layers_per_block)
    raise ValueError(f"{up_block_type} does not exist.")
