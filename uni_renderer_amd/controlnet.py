"""L2 networks of the dual-stream denoiser (mirror of the reference's ``models/controlnet.py``), MI355X-native.

  ``UNet2DConditionModel``   image stream            (ref 49-1166;  forward 781-1166)
  ``AttributeEncoderModel``  attribute stream, down  (ref 1170-1778; forward 1657-1778; from_unet 1437-1507)
  ``AttributeDecoderModel``  attribute stream, up    (ref 1781-2527; forward 2342-2527; from_unet 2115-2192)

Same class names, constructor keywords, ``forward`` signatures, return conventions and ``state_dict`` keys as
the reference, so ``train/train.py:1324-1354`` / ``models/pipeline.py:2660-2690`` call them unchanged.  Inside,
every tensor is NHWC fp16/bf16 and every op is a HIP kernel behind the C ABI (``ops``); tensors returned to the
caller are zero-copy NCHW *views* (channels-last strides) of those NHWC buffers, and such views are accepted
back without a copy -- the 13+13 exchange tensors never change layout between the three networks.

Configuration branches that the SD-1.x checkpoint family leaves inactive (class / addition embeddings,
encoder_hid_proj, GLIGEN, LoRA scale, attention masks) raise ``NotImplementedError``.
"""
from __future__ import annotations

import warnings

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .layers import Attention, Conv2d, Ctx, GroupNorm, PackCache, ResnetBlock2D, TimestepEmbedding, f32, pack_conv3x3, \
    pack_matrix, zero_module
from .modeling_utils import ConfigModelMixin, register_to_config
from .unet_2d_blocks import UNetMidBlock2DCrossAttn, get_down_block, get_up_block

CIN_PAD = 64  # conv_in reads its 4 / 28 input channels zero-padded to one 64-channel K chunk


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor = None


def _compute_dtype(param_dtype: torch.dtype, override: Optional[torch.dtype]) -> torch.dtype:
    if override is not None:
        return override
    if param_dtype in (torch.float16, torch.bfloat16):
        return param_dtype
    if torch.is_autocast_enabled():
        return torch.get_autocast_gpu_dtype()
    raise RuntimeError(
        "the MI355X path computes in fp16/bf16: cast the module (.to(torch.float16)), run under torch.autocast, "
        "or set module.compute_dtype")


def _as_tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _reject_inactive(**kw):
    for k, v in kw.items():
        if v is not None:
            raise NotImplementedError(f"{k} is not supported on the MI355X hot path (inactive in SD-1.x configs)")


class _DenoiserBase(ConfigModelMixin, nn.Module):
    """Shared plumbing: time embedding, batched per-resnet time projection, layout glue."""

    _supports_gradient_checkpointing = True
    compute_dtype: Optional[torch.dtype] = None

    def _finish_init(self):
        off = 0
        self._resnets: List[ResnetBlock2D] = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        for r in self._resnets:
            r.temb_slice = (off, off + r.out_channels)
            off += r.out_channels
        self._temb_total = off
        off = 0
        self._cross: List[Attention] = [m for m in self.modules() if isinstance(m, Attention) and m.is_cross]
        for a in self._cross:
            a.kv_slice = (off, off + a.inner)
            off += a.inner
        self._pk = PackCache()

    def _begin(self, B: int, timestep, device, encoder_hidden_states) -> Ctx:
        dt = _compute_dtype(self.dtype, self.compute_dtype)
        ctx = Ctx(dt, B)
        ctx.ehs = self._tokens(encoder_hidden_states, dt)
        if self._cross:  # every cross-attention K and V^T of the network in two GEMMs (they only see the prompt)
            wks = [a.to_k.weight for a in self._cross]
            wvs = [a.to_v.weight for a in self._cross]
            wk = self._pk.get("xk", wks, dt, lambda: torch.cat([pack_matrix(w, dt) for w in wks], 0).contiguous())
            wv = self._pk.get("xv", wvs, dt, lambda: torch.cat([pack_matrix(w, dt) for w in wvs], 0).contiguous())
            ctx.kc = ops.linear(ctx.ehs, wk)
            ctx.vtc = ops.vt_proj(ctx.ehs, wv)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=device)
        elif t.dim() == 0:
            t = t[None]
        t = t.to(device=device, dtype=torch.float32).contiguous()
        if t.numel() not in (1, B):
            raise ValueError(f"timestep has {t.numel()} entries for batch {B}")
        boc0 = self.config["block_out_channels"][0]
        t_emb = ops.timestep_embedding(t, B, boc0, self.config["flip_sin_to_cos"], self.config["freq_shift"], dt)
        semb = self.time_embedding(t_emb, silu_out=True)  # SiLU(emb), the only form the resnets consume
        ws = [r.time_emb_proj.weight for r in self._resnets]
        bs = [r.time_emb_proj.bias for r in self._resnets]
        wcat = self._pk.get("temb_w", ws, dt, lambda: torch.cat([pack_matrix(w, dt) for w in ws], 0).contiguous())
        bcat = self._pk.get("temb_b", bs, dt, lambda: torch.cat([f32(b) for b in bs], 0).contiguous())
        ctx.temb = ops.linear(semb, wcat, bcat)  # [B, sum Cout]
        return ctx

    def _autograd_mode(self, *tensors) -> bool:
        """True when this call must be differentiable: autograd is recording and a parameter or an input wants a
        gradient -- the situation of train/train.py:1324-1354 (modules called, then ``accelerator.backward(loss)``).
        The forwards then run over ``autograd_ops`` (train_step.py: HIP forward AND backward kernels, same return
        tuples); under ``torch.no_grad()`` or with frozen parameters and no input that requires a gradient they take the
        fused inference path.  ``train()`` / ``eval()`` does not enter the decision (as in torch)."""
        if not torch.is_grad_enabled():
            return False
        for t in tensors:
            if torch.is_tensor(t):
                if t.requires_grad:
                    return True
            elif isinstance(t, (list, tuple)) and any(torch.is_tensor(u) and u.requires_grad for u in t):
                return True
        # torch semantics: eval() never disables autograd.  A module whose parameters require a gradient, called with
        # autograd recording, must return outputs with a grad_fn whatever its train() / eval() flag -- otherwise a
        # fine-tuning set-up that leaves a network in eval() (frozen-statistics modes, partial training) would get NO
        # gradients for it and no error.  The common inference mistake -- calling a from_pretrained() module
        # (eval(), requires_grad parameters) outside torch.no_grad() -- therefore takes the slower differentiable path;
        # it is told so once.  (The reference's sampling methods are all @torch.no_grad(), pipeline.py Appendix B.)
        if not any(p.requires_grad for p in self.parameters()):
            return False
        if not self.training and not getattr(type(self), "_warned_eval_autograd", False):
            type(self)._warned_eval_autograd = True
            warnings.warn(f"{type(self).__name__} is in eval() mode but was called with autograd recording and parameters "
                          "that require gradients: taking the differentiable path (activations are saved).  For "
                          "inference wrap the call in torch.no_grad() or call .requires_grad_(False).", stacklevel=3)
        return True

    @staticmethod
    def _tokens(ehs: torch.Tensor, dt) -> torch.Tensor:
        if ehs.dtype != dt or not ehs.is_contiguous():
            ehs = ehs.to(dt).contiguous()  # boundary cast of the [B,77,768] prompt embedding (plumbing)
        return ehs

    def _conv_in(self, x_nchw: torch.Tensor, ctx: Ctx) -> torch.Tensor:
        dt = ctx.dtype
        w = self._pk.get("cin_w", [self.conv_in.weight], dt, lambda: pack_conv3x3(self.conv_in.weight, dt, CIN_PAD))
        b = self._pk.get("cin_b", [self.conv_in.bias], dt, lambda: f32(self.conv_in.bias))
        if x_nchw.shape[1] > CIN_PAD:
            raise NotImplementedError("conv_in with more than 64 input channels")
        return ops.conv3x3(ops.to_nhwc(x_nchw, dt, CIN_PAD), w, b, hilo=ops.PRECISE_RESIDUAL)

    def _conv_out(self, x: torch.Tensor, ctx: Ctx) -> torch.Tensor:
        dt = ctx.dtype
        g, b = self._pk.get("no", [self.conv_norm_out.weight, self.conv_norm_out.bias], dt,
                            lambda: (f32(self.conv_norm_out.weight), f32(self.conv_norm_out.bias)))
        w = self._pk.get("co_w", [self.conv_out.weight], dt, lambda: pack_conv3x3(self.conv_out.weight, dt))
        cb = self._pk.get("co_b", [self.conv_out.bias], dt, lambda: f32(self.conv_out.bias))
        h = ops.groupnorm(x, g, b, self.conv_norm_out.eps, groups=self.conv_norm_out.num_groups, silu=True)
        return ops.conv3x3(h, w, cb, n_out=self.conv_out.weight.shape[0])

    def _zero_conv(self, name: str, conv: Conv2d, x: torch.Tensor, ctx: Ctx, res=None, scale: float = 1.0):
        dt = ctx.dtype
        w = self._pk.get(name + "_w", [conv.weight], dt, lambda: pack_matrix(conv.weight, dt))
        b = self._pk.get(name + "_b", [conv.bias], dt, lambda: f32(conv.bias))
        return ops.linear(x, w, b, res=res, out_scale=scale, hilo=ops.PRECISE_RESIDUAL and res is not None)


def _exchange_channels(block_out_channels, layers_per_block):
    ch = [block_out_channels[0]]
    for i, c in enumerate(block_out_channels):
        ch += [c] * layers_per_block[i]
        if i != len(block_out_channels) - 1:
            ch.append(c)
    return ch


# =====================================================================================================
class UNet2DConditionModel(_DenoiserBase):
    """Image stream.  ``forward(..., return_dict=False)`` returns
    ``(sample, raw_down_block_res_samples[12], raw_mid_block_sample, up_block_res_samples[13])`` (ref 1163-1164)."""

    @register_to_config
    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 8,
        out_channels: int = 8,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
        mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
        up_block_types: Tuple[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
        only_cross_attention: Union[bool, Tuple[bool]] = False,
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: Union[int, Tuple[int]] = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        dropout: float = 0.0,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: Union[int, Tuple[int]] = 1280,
        transformer_layers_per_block: Union[int, Tuple[int]] = 1,
        reverse_transformer_layers_per_block=None,
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        num_attention_heads: Optional[Union[int, Tuple[int]]] = None,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        addition_embed_type: Optional[str] = None,
        addition_time_embed_dim: Optional[int] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        resnet_skip_time_act: bool = False,
        resnet_out_scale_factor: int = 1.0,
        time_embedding_type: str = "positional",
        time_embedding_dim: Optional[int] = None,
        time_embedding_act_fn: Optional[str] = None,
        timestep_post_act: Optional[str] = None,
        time_cond_proj_dim: Optional[int] = None,
        conv_in_kernel: int = 3,
        conv_out_kernel: int = 3,
        projection_class_embeddings_input_dim: Optional[int] = None,
        attention_type: str = "default",
        class_embeddings_concat: bool = False,
        mid_block_only_cross_attention: Optional[bool] = None,
        cross_attention_norm: Optional[str] = None,
        addition_embed_type_num_heads=64,
    ):
        super().__init__()
        self.sample_size = sample_size
        if num_attention_heads is not None:
            raise ValueError("num_attention_heads cannot be passed (ref controlnet.py:206-209); use attention_head_dim")
        num_attention_heads = attention_head_dim  # ref 216-222: the SD-1.x "attention_head_dim" is the head COUNT
        n = len(down_block_types)
        if len(up_block_types) != n or len(block_out_channels) != n:
            raise ValueError("down_block_types, up_block_types and block_out_channels must have the same length")
        _reject_inactive(encoder_hid_dim=encoder_hid_dim, encoder_hid_dim_type=encoder_hid_dim_type,
                         class_embed_type=class_embed_type, addition_embed_type=addition_embed_type,
                         num_class_embeds=num_class_embeds, time_embedding_act_fn=time_embedding_act_fn,
                         timestep_post_act=timestep_post_act, time_cond_proj_dim=time_cond_proj_dim,
                         cross_attention_norm=cross_attention_norm)
        if (dual_cross_attention or use_linear_projection or upcast_attention or resnet_skip_time_act
                or class_embeddings_concat or only_cross_attention not in (False, (False,) * n, [False] * n)
                or resnet_time_scale_shift != "default" or time_embedding_type != "positional"
                or attention_type != "default" or conv_in_kernel != 3 or conv_out_kernel != 3
                or mid_block_type != "UNetMidBlock2DCrossAttn" or norm_num_groups is None or dropout != 0.0
                or center_input_sample):
            raise NotImplementedError("configuration outside the SD-1.x family the MI355X hot path implements")
        heads = _as_tuple(num_attention_heads, n)
        cross = _as_tuple(cross_attention_dim, n)
        layers = list(_as_tuple(layers_per_block, n))
        tlayers = list(_as_tuple(transformer_layers_per_block, n))
        boc = tuple(block_out_channels)
        temb_c = time_embedding_dim or boc[0] * 4

        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)

        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(get_down_block(
                kind, num_layers=layers[i], transformer_layers_per_block=tlayers[i], in_channels=in_c,
                out_channels=out_c, temb_channels=temb_c, add_downsample=i != n - 1, resnet_eps=norm_eps,
                resnet_act_fn=act_fn, resnet_groups=norm_num_groups, cross_attention_dim=cross[i],
                num_attention_heads=heads[i], downsample_padding=downsample_padding))
        self.mid_block = UNetMidBlock2DCrossAttn(
            boc[-1], temb_c, resnet_eps=norm_eps, resnet_groups=norm_num_groups, num_attention_heads=heads[-1],
            cross_attention_dim=cross[-1], output_scale_factor=mid_block_scale_factor,
            transformer_layers_per_block=tlayers[-1])

        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rboc, rheads, rlayers, rcross = boc[::-1], heads[::-1], layers[::-1], cross[::-1]
        rtl = tlayers[::-1] if reverse_transformer_layers_per_block is None else reverse_transformer_layers_per_block
        out_c = rboc[0]
        for i, kind in enumerate(up_block_types):
            prev_c, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, n - 1)]
            add_up = i != n - 1
            self.num_upsamplers += int(add_up)
            self.up_blocks.append(get_up_block(
                kind, num_layers=rlayers[i] + 1, transformer_layers_per_block=rtl[i], in_channels=in_c,
                out_channels=out_c, prev_output_channel=prev_c, temb_channels=temb_c, add_upsample=add_up,
                resnet_eps=norm_eps, resnet_act_fn=act_fn, resolution_idx=i, resnet_groups=norm_num_groups,
                cross_attention_dim=rcross[i], num_attention_heads=rheads[i]))
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)
        self._finish_init()

    def forward(
        self,
        sample: torch.Tensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        class_labels: Optional[torch.Tensor] = None,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residual: Optional[torch.Tensor] = None,
        down_intrablock_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        encoder_attention_mask: Optional[torch.Tensor] = None,
        return_dict: bool = True,
    ):
        _reject_inactive(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                         cross_attention_kwargs=cross_attention_kwargs, added_cond_kwargs=added_cond_kwargs,
                         down_intrablock_additional_residuals=down_intrablock_additional_residuals,
                         encoder_attention_mask=encoder_attention_mask)
        if self._autograd_mode(sample, encoder_hidden_states, down_block_additional_residuals, mid_block_additional_residual):
            return self._forward_autograd(sample, timestep, encoder_hidden_states, down_block_additional_residuals,
                                          mid_block_additional_residual, return_dict)
        state = self.forward_down_mid(sample, timestep, encoder_hidden_states)
        return self.forward_up(state, down_block_additional_residuals, mid_block_additional_residual, return_dict)

    def _forward_autograd(self, sample, timestep, ehs, down_res, mid_res, return_dict):
        """The differentiable forward (train_step.unet_forward) behind the reference's signature and return tuple."""
        from . import train_step as TS

        dt = _compute_dtype(self.dtype, self.compute_dtype)
        if (down_res is None) != (mid_res is None):
            raise NotImplementedError("T2I-adapter style down_block_additional_residuals without a mid residual")
        res = [TS.to_nhwc_grad(r, dt) for r in down_res] if down_res is not None else None
        mid = TS.to_nhwc_grad(mid_res, dt) if mid_res is not None else None
        img, raw_down, raw_mid, ups = TS.unet_forward(self, TS.to_nhwc_grad(sample, dt, CIN_PAD),
                                                      TS._prompt(ehs, sample.shape[0], dt), timestep, dt, res, mid,
                                                      collect_up=True)
        v = lambda t: t.permute(0, 3, 1, 2)
        if not return_dict:
            return v(img), tuple(v(t) for t in raw_down), v(raw_mid), tuple(v(t) for t in ups)
        return UNet2DConditionOutput(sample=v(img))

    # The forward is split at the point where the other stream's features enter (ref 1078-1087): the down path +
    # mid block do not depend on the encoder, so `graph.GraphedDualStreamStep` runs them concurrently with it.
    def forward_down_mid(self, sample, timestep, encoder_hidden_states):
        B, _, H, W = sample.shape
        ctx = self._begin(B, timestep, sample.device, encoder_hidden_states)
        x = self._conv_in(sample, ctx)
        skips = (x,)
        for blk in self.down_blocks:
            x, st = blk(x, ctx)
            skips += st
        raw_mid = self.mid_block(x, ctx)
        return dict(ctx=ctx, raw_down=skips, raw_mid=raw_mid, hw=(H, W))

    def forward_up(self, state, down_block_additional_residuals=None, mid_block_additional_residual=None,
                   return_dict: bool = True):
        ctx, raw_down, raw_mid = state["ctx"], state["raw_down"], state["raw_mid"]
        H, W = state["hw"]
        dt = ctx.dtype
        f = 2 ** self.num_upsamplers
        forward_upsample_size = (H % f != 0) or (W % f != 0)  # ref 869-883
        skips, x = raw_down, raw_mid
        is_controlnet = mid_block_additional_residual is not None and down_block_additional_residuals is not None
        if down_block_additional_residuals is not None and not is_controlnet:
            raise NotImplementedError("T2I-adapter style down_block_additional_residuals without a mid residual")
        if is_controlnet:  # ref 1078-1087: 12 exchange adds; ref 1114-1115: mid
            hl = ops.PRECISE_RESIDUAL
            skips = tuple(ops.add(s, ops.to_nhwc(e, dt), hilo=hl) for s, e in zip(skips, down_block_additional_residuals))
            x = ops.add(x, ops.to_nhwc(mid_block_additional_residual, dt), hilo=hl)
        up_res = (x,)
        for i, blk in enumerate(self.up_blocks):
            nres = len(blk.resnets)
            res, skips = skips[-nres:], skips[:-nres]
            upsample_size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = skips[-1].shape[1:3]
            x, ups = blk(x, res, ctx, upsample_size=upsample_size)
            up_res += ups
        out = ops.as_nchw_view(self._conv_out(x, ctx))
        if not return_dict:
            v = ops.as_nchw_view
            return out, tuple(v(s) for s in raw_down), v(raw_mid), tuple(v(s) for s in up_res)
        return UNet2DConditionOutput(sample=out)


# =====================================================================================================
class AttributeEncoderModel(_DenoiserBase):
    """Attribute stream, encoder half: conv_in + down + mid of a UNet plus 12+1 zero-initialised 1x1 convs.
    ``forward`` ignores ``sample`` (ref 1716-1720) and ALWAYS returns the 4-tuple
    ``(down_block_res_samples list[12], mid_block_res_sample, raw_down tuple[12], raw_mid)`` (ref 1778)."""

    @register_to_config
    def __init__(
        self,
        in_channels: int = 4,
        conditioning_channels: int = 3,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
        only_cross_attention: Union[bool, Tuple[bool]] = False,
        block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1280,
        transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1,
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        attention_head_dim: Union[int, Tuple[int, ...]] = 8,
        num_attention_heads: Optional[Union[int, Tuple[int, ...]]] = None,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        addition_embed_type: Optional[str] = None,
        addition_time_embed_dim: Optional[int] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        projection_class_embeddings_input_dim: Optional[int] = None,
        controlnet_conditioning_channel_order: str = "rgb",
        conditioning_embedding_out_channels: Optional[Tuple[int, ...]] = (16, 32, 96, 256),
        global_pool_conditions: bool = False,
        addition_embed_type_num_heads: int = 64,
        len_t: int = 1,
    ):
        super().__init__()
        self.len_t = len_t
        num_attention_heads = num_attention_heads or attention_head_dim
        n = len(down_block_types)
        if len(block_out_channels) != n:
            raise ValueError("block_out_channels and down_block_types must have the same length")
        _reject_inactive(encoder_hid_dim=encoder_hid_dim, encoder_hid_dim_type=encoder_hid_dim_type,
                         class_embed_type=class_embed_type, addition_embed_type=addition_embed_type,
                         num_class_embeds=num_class_embeds)
        if (use_linear_projection or upcast_attention or resnet_time_scale_shift != "default"
                or global_pool_conditions or norm_num_groups is None
                or only_cross_attention not in (False, (False,) * n, [False] * n)):
            raise NotImplementedError("configuration outside the SD-1.x family the MI355X hot path implements")
        heads = _as_tuple(num_attention_heads, n)
        tlayers = list(_as_tuple(transformer_layers_per_block, n))
        boc = tuple(block_out_channels)
        temb_c = boc[0] * 4

        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList()
        out_c = boc[0]
        self.controlnet_down_blocks.append(zero_module(Conv2d(out_c, out_c, 1)))
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(get_down_block(
                kind, num_layers=layers_per_block, transformer_layers_per_block=tlayers[i], in_channels=in_c,
                out_channels=out_c, temb_channels=temb_c, add_downsample=i != n - 1, resnet_eps=norm_eps,
                resnet_act_fn=act_fn, resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                num_attention_heads=heads[i], downsample_padding=downsample_padding))
            for _ in range(layers_per_block):
                self.controlnet_down_blocks.append(zero_module(Conv2d(out_c, out_c, 1)))
            if i != n - 1:
                self.controlnet_down_blocks.append(zero_module(Conv2d(out_c, out_c, 1)))
        self.controlnet_mid_block = zero_module(Conv2d(boc[-1], boc[-1], 1))
        self.mid_block = UNetMidBlock2DCrossAttn(
            boc[-1], temb_c, resnet_eps=norm_eps, resnet_groups=norm_num_groups, num_attention_heads=heads[-1],
            cross_attention_dim=cross_attention_dim, output_scale_factor=mid_block_scale_factor,
            transformer_layers_per_block=tlayers[-1])
        self._finish_init()

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, controlnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels=(16, 32, 96, 256), load_weights_from_unet: bool = True,
                  len_t: int = 1):
        c = unet.config
        m = cls(
            transformer_layers_per_block=c.get("transformer_layers_per_block", 1), in_channels=c["in_channels"],
            flip_sin_to_cos=c["flip_sin_to_cos"], freq_shift=c["freq_shift"], down_block_types=c["down_block_types"],
            only_cross_attention=c["only_cross_attention"], block_out_channels=c["block_out_channels"],
            layers_per_block=c["layers_per_block"], downsample_padding=c["downsample_padding"],
            mid_block_scale_factor=c["mid_block_scale_factor"], act_fn=c["act_fn"],
            norm_num_groups=c["norm_num_groups"], norm_eps=c["norm_eps"], cross_attention_dim=c["cross_attention_dim"],
            attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"],
            use_linear_projection=c["use_linear_projection"], upcast_attention=c["upcast_attention"],
            resnet_time_scale_shift=c["resnet_time_scale_shift"],
            controlnet_conditioning_channel_order=controlnet_conditioning_channel_order,
            conditioning_embedding_out_channels=conditioning_embedding_out_channels, len_t=1)  # ref 1492
        if load_weights_from_unet:  # ref 1496-1505
            m.conv_in.load_state_dict(unet.conv_in.state_dict())
            m.time_embedding.load_state_dict(unet.time_embedding.state_dict())
            m.down_blocks.load_state_dict(unet.down_blocks.state_dict())
            m.mid_block.load_state_dict(unet.mid_block.state_dict())
        return m

    def forward(
        self,
        sample: torch.Tensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        controlnet_cond: torch.Tensor,
        conditioning_scale: float = 1.0,
        class_labels: Optional[torch.Tensor] = None,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        return_dict: bool = True,
    ):
        _reject_inactive(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                         added_cond_kwargs=added_cond_kwargs, cross_attention_kwargs=cross_attention_kwargs)
        B = controlnet_cond.shape[0]
        if self._autograd_mode(controlnet_cond, encoder_hidden_states):
            from . import train_step as TS

            dt = _compute_dtype(self.dtype, self.compute_dtype)
            res, mid, raw_down, raw_mid = TS.encoder_forward(
                self, TS.to_nhwc_grad(controlnet_cond, dt, CIN_PAD), TS._prompt(encoder_hidden_states, B, dt), timestep, dt,
                conditioning_scale=float(conditioning_scale))
            v = lambda t: t.permute(0, 3, 1, 2)
            return [v(t) for t in res], v(mid), tuple(v(t) for t in raw_down), v(raw_mid)
        ctx = self._begin(B, timestep, controlnet_cond.device, encoder_hidden_states)
        x = self._conv_in(controlnet_cond, ctx)  # `sample` is ignored, as in the reference
        skips = (x,)
        for blk in self.down_blocks:
            x, st = blk(x, ctx)
            skips += st
        raw_down = skips
        x = self.mid_block(x, ctx)
        raw_mid = x
        s = float(conditioning_scale)
        res = [self._zero_conv(f"z{i}", z, t, ctx, scale=s) for i, (t, z) in enumerate(zip(skips, self.controlnet_down_blocks))]
        mid = self._zero_conv("zmid", self.controlnet_mid_block, x, ctx, scale=s)
        v = ops.as_nchw_view
        return [v(t) for t in res], v(mid), tuple(v(t) for t in raw_down), v(raw_mid)


# =====================================================================================================
class AttributeDecoderModel(_DenoiserBase):
    """Attribute stream, decoder half: the up path of a UNet fed by the encoder's raw skips, with the image
    stream's raw skips mixed in through 12+1 zero-initialised 1x1 convs -- fused here as the residual epilogue of
    the 1x1-conv GEMM: ``skip_enc[i] + conv1x1_i(skip_unet[i])`` (ref 2446-2461, 2476-2477)."""

    @register_to_config
    def __init__(
        self,
        out_channels: int = 4,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        up_block_types: Tuple[str] = ("UpResBlock2D", "CrossAttnUpResBlock2D", "CrossAttnUpResBlock2D", "CrossAttnUpResBlock2D"),
        only_cross_attention: Union[bool, Tuple[bool]] = False,
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: Union[int, Tuple[int]] = 2,
        dropout: float = 0.0,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: Union[int, Tuple[int]] = 1280,
        transformer_layers_per_block: Union[int, Tuple[int]] = 1,
        reverse_transformer_layers_per_block=None,
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        num_attention_heads: Optional[Union[int, Tuple[int]]] = None,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        addition_embed_type: Optional[str] = None,
        addition_time_embed_dim: Optional[int] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        resnet_skip_time_act: bool = False,
        resnet_out_scale_factor: int = 1.0,
        conv_out_kernel: int = 3,
        projection_class_embeddings_input_dim: Optional[int] = None,
        attention_type: str = "default",
        class_embeddings_concat: bool = False,
        cross_attention_norm: Optional[str] = None,
        addition_embed_type_num_heads=64,
        len_t: int = 1,
    ):
        super().__init__()
        self.len_t = len_t
        num_attention_heads = num_attention_heads or attention_head_dim
        n = len(up_block_types)
        if len(block_out_channels) != n:
            raise ValueError("block_out_channels and up_block_types must have the same length")
        _reject_inactive(encoder_hid_dim=encoder_hid_dim, encoder_hid_dim_type=encoder_hid_dim_type,
                         class_embed_type=class_embed_type, addition_embed_type=addition_embed_type,
                         num_class_embeds=num_class_embeds, cross_attention_norm=cross_attention_norm)
        if (dual_cross_attention or use_linear_projection or upcast_attention or resnet_skip_time_act
                or class_embeddings_concat or resnet_time_scale_shift != "default" or attention_type != "default"
                or conv_out_kernel != 3 or norm_num_groups is None or dropout != 0.0
                or only_cross_attention not in (False, (False,) * n, [False] * n)):
            raise NotImplementedError("configuration outside the SD-1.x family the MI355X hot path implements")
        heads = _as_tuple(num_attention_heads, n)
        cross = _as_tuple(cross_attention_dim, n)
        layers = list(_as_tuple(layers_per_block, n))
        tlayers = list(_as_tuple(transformer_layers_per_block, n))
        boc = tuple(block_out_channels)
        temb_c = boc[0] * 4
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.num_upsamplers = 0
        self.control_down_blocks = nn.ModuleList(
            [zero_module(Conv2d(c, c, 1)) for c in _exchange_channels(boc, layers)])
        self.control_mid_block = zero_module(Conv2d(boc[-1], boc[-1], 1))
        self.up_blocks = nn.ModuleList()
        rboc, rheads, rlayers, rcross = boc[::-1], heads[::-1], layers[::-1], cross[::-1]
        rtl = tlayers[::-1] if reverse_transformer_layers_per_block is None else reverse_transformer_layers_per_block
        out_c = rboc[0]
        for i, kind in enumerate(up_block_types):
            prev_c, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, n - 1)]
            add_up = i != n - 1
            self.num_upsamplers += int(add_up)
            self.up_blocks.append(get_up_block(
                kind, num_layers=rlayers[i] + 1, transformer_layers_per_block=rtl[i], in_channels=in_c,
                out_channels=out_c, prev_output_channel=prev_c, temb_channels=temb_c, add_upsample=add_up,
                resnet_eps=norm_eps, resnet_act_fn=act_fn, resolution_idx=i, resnet_groups=norm_num_groups,
                cross_attention_dim=rcross[i], num_attention_heads=rheads[i]))
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)
        self._finish_init()

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, load_weights_from_unet: bool = True, len_t: int = 1):
        c = unet.config
        m = cls(
            out_channels=c["out_channels"], flip_sin_to_cos=c["flip_sin_to_cos"], freq_shift=c["freq_shift"],
            up_block_types=c["up_block_types"],  # ref 2150: the UNet's plain Up blocks, not the UpRes defaults
            only_cross_attention=c["only_cross_attention"], block_out_channels=c["block_out_channels"],
            layers_per_block=c["layers_per_block"], dropout=c["dropout"], act_fn=c["act_fn"],
            norm_num_groups=c["norm_num_groups"], norm_eps=c["norm_eps"], cross_attention_dim=c["cross_attention_dim"],
            transformer_layers_per_block=c.get("transformer_layers_per_block", 1),
            reverse_transformer_layers_per_block=c.get("reverse_transformer_layers_per_block"),
            attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"],
            dual_cross_attention=c["dual_cross_attention"], use_linear_projection=c["use_linear_projection"],
            upcast_attention=c["upcast_attention"], resnet_time_scale_shift=c["resnet_time_scale_shift"],
            resnet_skip_time_act=c["resnet_skip_time_act"], resnet_out_scale_factor=c["resnet_out_scale_factor"],
            conv_out_kernel=c["conv_out_kernel"], len_t=len_t)
        if load_weights_from_unet:  # ref 2182-2190
            m.conv_out.load_state_dict(unet.conv_out.state_dict())
            m.time_embedding.load_state_dict(unet.time_embedding.state_dict())
            m.up_blocks.load_state_dict(unet.up_blocks.state_dict())
        return m

    def forward(
        self,
        sample: torch.Tensor,
        down_block_res_samples: Tuple[torch.Tensor],
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        up_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residual: Optional[torch.Tensor] = None,
        class_labels: Optional[torch.Tensor] = None,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        return_dict: bool = True,
    ):
        _reject_inactive(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                         added_cond_kwargs=added_cond_kwargs, cross_attention_kwargs=cross_attention_kwargs)
        if mid_block_additional_residual is None:
            raise ValueError("mid_block_additional_residual is mandatory (the reference crashes on None, ref 2476)")
        B = sample.shape[0]
        if self._autograd_mode(sample, encoder_hidden_states, down_block_res_samples, down_block_additional_residuals,
                               up_block_additional_residuals, mid_block_additional_residual):
            from . import train_step as TS

            dt = _compute_dtype(self.dtype, self.compute_dtype)
            g = lambda t: TS.to_nhwc_grad(t, dt)
            extras = [g(u) for u in up_block_additional_residuals[1:]] if up_block_additional_residuals is not None else None
            out = TS.decoder_forward(
                self, g(sample), [g(t) for t in down_block_res_samples], TS._prompt(encoder_hidden_states, B, dt), timestep, dt,
                raw_unet=([g(t) for t in down_block_additional_residuals] if down_block_additional_residuals is not None else None),
                raw_mid_unet=g(mid_block_additional_residual), extras=extras).permute(0, 3, 1, 2)
            return out if not return_dict else UNet2DConditionOutput(sample=out)
        H0, W0 = down_block_res_samples[0].shape[-2:]
        f = 2 ** self.num_upsamplers
        forward_upsample_size = (H0 % f != 0) or (W0 % f != 0)
        ctx = self._begin(B, timestep, sample.device, encoder_hidden_states)
        dt = ctx.dtype
        skips = tuple(ops.to_nhwc(s, dt) for s in down_block_res_samples)
        if down_block_additional_residuals is not None:  # ref 2446-2461
            skips = tuple(
                self._zero_conv(f"c{i}", z, ops.to_nhwc(e, dt), ctx, res=s)
                for i, (s, e, z) in enumerate(zip(skips, down_block_additional_residuals, self.control_down_blocks)))
        x = self._zero_conv("cmid", self.control_mid_block, ops.to_nhwc(mid_block_additional_residual, dt), ctx,
                            res=ops.to_nhwc(sample, dt))  # ref 2476-2477
        ups = None
        if up_block_additional_residuals is not None:  # only consumed by UpRes* block types
            ups = [ops.to_nhwc(u, dt) for u in up_block_additional_residuals]
        k = 1
        for i, blk in enumerate(self.up_blocks):
            nres = len(blk.resnets)
            res, skips = skips[-nres:], skips[:-nres]
            upsample_size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = skips[-1].shape[1:3]
            extra = tuple(ups[k:k + nres]) if ups is not None else None
            k += nres
            x, _ = blk(x, res, ctx, upsample_size=upsample_size, up_additional_states_tuple=extra)
        out = ops.as_nchw_view(self._conv_out(x, ctx))
        if not return_dict:
            return out
        return UNet2DConditionOutput(sample=out)
