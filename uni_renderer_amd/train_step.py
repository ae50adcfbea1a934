"""Training step of the dual-stream denoiser on the MI355X (SURVEY section 8a device op 11, cfg 4; reference
train/train.py:1258-1427): the forward pass of AttributeEncoderModel -> UNet2DConditionModel -> AttributeDecoderModel
written over the autograd Functions of ``autograd_ops`` (every forward AND backward kernel is this package's HIP code;
torch.autograd only walks the graph), the reference's x0-prediction MSE losses, and the data-parallel gradient
all-reduce of ``parallel.GradientBuckets`` (RCCL over xGMI, one flat bucket list for the three networks instead of the
reference's three DDP wrappers, train.py:1140-1142).

The modules hold the fp32 master parameters (as under the reference's autocast); they are cast to the compute dtype
inside the graph, so gradients arrive in fp32 on ``param.grad``.  This path trades the inference path's fusions for
differentiability: GEGLU is its own kernel, the q / k / v projections are separate GEMMs, the up-path concatenation is
materialised, attention recomputes and materialises P in the backward.  It is a functional first version -- parity of
loss and gradients against the CPU restatement of the reference under autograd is tested (tests/test_train_gpu.py); throughput work comes next.
"""
from __future__ import annotations


from typing import Dict, List, Optional, Sequence

import torch

from . import _experiments as X
from . import autograd_ops as A
from . import backward as B
from . import ops
from .controlnet import CIN_PAD
from .layers import BasicTransformerBlock, ResnetBlock2D


BATCH_CASTS = X.flag("batch_casts", True)
_cast: Dict[int, torch.Tensor] = {}          # id(fp32 weight) -> its compute-dtype copy, valid inside one network forward
_castable: Dict[int, tuple] = {}             # id(network) -> (network, its Linear / 1x1-conv weights)


class _batched_casts:
    """All Linear / 1x1-conv weights of ``net`` cast to ``dt`` by ONE multi-tensor launch (autograd_ops.CastParams) for
    the duration of a network forward; ``_wc`` / ``_w2`` pick the copies up.  Their gradients return to fp32 in one
    launch when the network's backward has produced the last of them."""

    def __init__(self, net, dt):
        self.net, self.dt = net, dt

    def __enter__(self):
        if not (BATCH_CASTS and torch.is_grad_enabled()):
            return self
        hit = _castable.get(id(self.net))
        if hit is None or hit[0] is not self.net:
            ws = [m.weight for m in self.net.modules()
                  if isinstance(m, torch.nn.Linear) or (isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1))]
            # order = memory order of the packed copies (CastParams): the weights this file concatenates into one GEMM
            # operand come first and in the order of their use, so the "concatenation" is a view (A.cat_adjacent):
            # all time_emb_proj, then the to_k / to_v pairs of the cross-attentions; q | k | v of a self-attention are
            # neighbours in module order already
            temb = [m.time_emb_proj.weight for m in self.net.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
            ctxw = [w for m in self.net.modules() if isinstance(m, BasicTransformerBlock) and getattr(m, "attn2", None) is not None
                    for w in (m.attn2.to_k.weight, m.attn2.to_v.weight)]
            first = {id(w) for w in temb + ctxw}
            ws = temb + ctxw + [w for w in ws if id(w) not in first]
            bs = [m.bias for m in self.net.modules()
                  if (isinstance(m, torch.nn.Linear) or (isinstance(m, torch.nn.Conv2d) and m.kernel_size in ((1, 1), (3, 3))))
                  and m.bias is not None and m.bias.dtype == torch.float32]
            # gamma / beta of the norm layers too: their gradients are summed in one launch at the barrier (backward.NormSums)
            bs += [p_ for m in self.net.modules() if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)) and m.weight is not None
                   for p_ in (m.weight, m.bias) if p_ is not None and p_.dtype == torch.float32]
            conv_in = getattr(self.net, "conv_in", None)
            c3 = [(m.weight, CIN_PAD if m is conv_in else None) for m in self.net.modules()
                  if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.weight.dtype == torch.float32]
            hit = _castable[id(self.net)] = (self.net, [w for w in ws if w.dtype == torch.float32], bs, c3)
        ws = [w for w in hit[1] if w.requires_grad]
        if ws:
            for w, c in zip(ws, A.CastParams.apply(self.dt, *ws)):
                _cast[id(w)] = c
        self.keys = [id(w) for w in ws]
        # the biases of the same layers behind one barrier node: their gradients may then be deferred with the weights'
        # (autograd_ops.ParamBarrier; A.linear looks them up in A.deferred_bias)
        bs = [b for b in hit[2] if b.requires_grad] if (ws and B.WGRAD_DEFER and B.WGRAD) else []
        if bs:
            for b, o in zip(bs, A.ParamBarrier.apply(*bs)):
                A.deferred_bias[id(b)] = o
        self.bkeys = [id(b) for b in bs]
        # ... and the 3x3 conv weights packed up front by one node (autograd_ops.PackConvWeights), same reason
        c3 = [(w, cp) for w, cp in hit[3] if w.requires_grad and w.is_cuda] if (B.WGRAD_DEFER and B.WGRAD) else []
        if c3:
            outs = A.PackConvWeights.apply(self.dt, tuple(cp for _, cp in c3), *[w for w, _ in c3])
            for (w, cp), o in zip(c3, outs):
                A.packed_conv[(id(w), cp)] = o
        self.ckeys = [(id(w), cp) for w, cp in c3]
        return self

    def __exit__(self, *exc):
        for k in getattr(self, "keys", ()):
            _cast.pop(k, None)
        for k in getattr(self, "bkeys", ()):
            A.deferred_bias.pop(k, None)
        for k in getattr(self, "ckeys", ()):
            A.packed_conv.pop(k, None)
        return False


def _wc(w, dt):
    """compute-dtype copy of the weight ``w`` (from the network's batched cast when there is one)."""
    c = _cast.get(id(w))
    return c if c is not None and c.dtype == dt else w.to(dt)


def _w2(conv_or_lin, dt):
    """[N, K] compute-dtype view of a Linear / 1x1-conv weight (differentiable)."""
    w = _wc(conv_or_lin.weight, dt)
    return w.reshape(w.shape[0], -1)


def _temb_projections(resnets, temb_act, dt):
    """All ``time_emb_proj`` of a network phase as ONE GEMM (what the inference path does too): the M = batch-size
    linears, their two backward GEMMs, three transposes and the bias column sum per resnet are launch-bound, ~25
    launches per resnet.  Returns {id(resnet): [B, C_out] column slice}; autograd splits the gradients back."""
    w = A.cat_adjacent([_wc(r.time_emb_proj.weight, dt) for r in resnets])
    b = torch.cat([r.time_emb_proj.bias for r in resnets], 0)
    t_all = A.linear(temb_act, w, b)
    # torch.split, not per-resnet slicing: its backward is ONE cat of the slice gradients, where every SliceBackward
    # allocates a zero-filled full-width tensor, copies its slice in and adds it to the running sum (3 launches each)
    parts = torch.split(t_all, [r.time_emb_proj.weight.shape[0] for r in resnets], dim=1)
    return {id(r): part for r, part in zip(resnets, parts)}


def _resnets_of(blocks, mid=None):
    rs = [r for blk in blocks for r in blk.resnets]
    return rs + (list(mid.resnets) if mid is not None else [])


def _resnet(r, x, temb, dt, x1=None):
    """models/unet_2d_blocks.py:1100-1111 (ResnetBlock2D, time_embedding_norm='default'); ``temb``: the dict of
    _temb_projections."""
    xin = torch.cat([x, x1], -1) if x1 is not None else x
    h = A.group_norm(xin, r.norm1.weight, r.norm1.bias, r.eps, r.groups, True)
    t = temb[id(r)]
    h = A.conv3x3(h, A.pack_conv_weight(r.conv1.weight, dt), r.conv1.bias, rowadd=t)
    h = A.group_norm(h, r.norm2.weight, r.norm2.bias, r.eps, r.groups, True)
    sc = xin if r.conv_shortcut is None else A.linear(xin, _w2(r.conv_shortcut, dt), r.conv_shortcut.bias)
    if r.output_scale_factor != 1.0:
        raise NotImplementedError("output_scale_factor != 1 in the training path")
    return A.conv3x3(h, A.pack_conv_weight(r.conv2.weight, dt), r.conv2.bias, res=sc)


def _resnet_ckpt(blk, r, x, temb, dt, x1=None):
    """``_resnet``, recomputed in the backward when the block asks for it (unet_2d_blocks.py:1172-1197: training mode and
    ``gradient_checkpointing``; ``use_reentrant=False`` like the reference).  The recompute runs the same kernels on the same
    inputs, so the saved tensors -- and with them every gradient -- are bit-identical to the ones the plain forward keeps."""
    if getattr(blk, "gradient_checkpointing", False) and torch.is_grad_enabled() and blk.training:
        from torch.utils.checkpoint import checkpoint

        t = temb[id(r)]
        fn = (lambda x_, t_, x1_: _resnet(r, x_, {id(r): t_}, dt, x1=x1_)) if x1 is not None else \
             (lambda x_, t_: _resnet(r, x_, {id(r): t_}, dt))
        args = (x, t, x1) if x1 is not None else (x, t)
        return checkpoint(fn, *args, use_reentrant=False)
    return _resnet(r, x, temb, dt, x1=x1)


def _self_attn(a, xn, res, dt):
    """q | k | v as ONE projection (one forward and two backward GEMMs instead of three of each)."""
    Cc = a.to_q.weight.shape[0]
    qkv = A.linear(xn, A.cat_adjacent([_wc(a.to_q.weight, dt), _wc(a.to_k.weight, dt), _wc(a.to_v.weight, dt)]))
    o = A.AttentionQKV.apply(qkv, a.heads)
    return A.linear(o, _w2(a.to_out[0], dt), a.to_out[0].bias, res=res)


def _cross_attn(a, xn, kv, res, dt):
    """``kv`` [B, 77, 2C]: this block's columns of the phase-wide prompt projection (_context_projections)."""
    Cc = a.to_q.weight.shape[0]
    q = A.linear(xn, _w2(a.to_q, dt))
    k_, v_ = torch.split(kv, Cc, dim=-1)
    o = A.Attention.apply(q, k_, v_, a.heads)
    return A.linear(o, _w2(a.to_out[0], dt), a.to_out[0].bias, res=res)


def _context_projections(blocks, ehs, dt):
    """to_k | to_v of every cross-attention of a network phase applied to the prompt embedding as ONE GEMM (M = 77 B
    rows: per block these projections and their backward are launch-bound).  {id(block): [B, 77, 2C] column slice}."""
    tbs = [tb for blk in blocks for t in getattr(blk, "attentions", []) for tb in t.transformer_blocks]
    if not tbs:
        return {}
    w = A.cat_adjacent([_wc(w_, dt) for tb in tbs for w_ in (tb.attn2.to_k.weight, tb.attn2.to_v.weight)])
    kv_all = A.linear(ehs, w)
    parts = torch.split(kv_all, [2 * tb.attn2.to_k.weight.shape[0] for tb in tbs], dim=-1)  # one cat in the backward
    return {id(tb): part for tb, part in zip(tbs, parts)}


def _tblock(b, x, kvs, dt):
    # (x, LN(x)) as one autograd node: the residual's gradient is added inside the LayerNorm backward kernel
    xs, xn = A.layer_norm_skip(x, b.norm1.weight, b.norm1.bias, b.norm1.eps)
    x = _self_attn(b.attn1, xn, xs, dt)
    xs, xn = A.layer_norm_skip(x, b.norm2.weight, b.norm2.bias, b.norm2.eps)
    x = _cross_attn(b.attn2, xn, kvs[id(b)], xs, dt)
    proj, out = b.ff.net[0].proj, b.ff.net[2]
    xs, xn = A.layer_norm_skip(x, b.norm3.weight, b.norm3.bias, b.norm3.eps)
    h = A.linear(xn, _w2(proj, dt), proj.bias)
    return A.linear(A.GEGLU.apply(h), _w2(out, dt), out.bias, res=xs)


def _transformer(t, x, ehs, dt):
    B, H, W, Cc = x.shape
    h = A.group_norm(x, t.norm.weight, t.norm.bias, t.norm.eps, t.groups, False)
    h = A.linear(h.view(B, H * W, Cc), _w2(t.proj_in, dt), t.proj_in.bias)
    for blk in t.transformer_blocks:
        h = _tblock(blk, h, ehs, dt)
    return A.linear(h, _w2(t.proj_out, dt), t.proj_out.bias, res=x.view(B, H * W, Cc)).view(B, H, W, Cc)


def _conv(m, x, dt, stride=1, cin_pad=None):
    return A.conv3x3(x, A.pack_conv_weight(m.weight, dt, cin_pad), m.bias, stride=stride)


def _time(net, timesteps, B, dt, dev):
    t = torch.as_tensor(timesteps, device=dev, dtype=torch.float32).reshape(-1)
    t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
    c = net.config
    t_emb = ops.timestep_embedding(t, B, c["block_out_channels"][0], c["flip_sin_to_cos"], c["freq_shift"], dt)
    te = net.time_embedding
    emb = A.linear(A.SiLU.apply(A.linear(t_emb, _w2(te.linear_1, dt), te.linear_1.bias)), _w2(te.linear_2, dt),
                   te.linear_2.bias)
    return A.SiLU.apply(emb)  # every resnet applies SiLU to the embedding before its projection


def _down_mid(net, x, temb_act, ehs, dt):
    temb = _temb_projections(_resnets_of(net.down_blocks, net.mid_block), temb_act, dt)
    ehs = _context_projections(list(net.down_blocks) + [net.mid_block], ehs, dt)  # from here on: per-block K | V
    skips = [x]
    for blk in net.down_blocks:
        for i, r in enumerate(blk.resnets):
            x = _resnet_ckpt(blk, r, x, temb, dt)
            if getattr(blk, "has_cross_attention", False):
                x = _transformer(blk.attentions[i], x, ehs, dt)
            skips.append(x)
        if blk.downsamplers is not None:
            x = _conv(blk.downsamplers[0].conv, x, dt, stride=2)
            skips.append(x)
    m = net.mid_block
    x = _resnet(m.resnets[0], x, temb, dt)
    for a, r in zip(m.attentions, m.resnets[1:]):
        x = _resnet(r, _transformer(a, x, ehs, dt), temb, dt)
    return x, skips


def _up_out(net, x, skips: List[torch.Tensor], temb_act, ehs, dt, extras=None, collect=None):
    """Up path + conv_out.  ``extras``: per-resnet tensors added after each resnet(/transformer) of an ``UpRes*`` block
    (unet_2d_blocks.py:2408, 2814); ``collect``: list that receives the per-layer outputs (the reference-modified
    blocks return them, 2584-2590 / 2697-2704)."""
    temb = _temb_projections(_resnets_of(net.up_blocks), temb_act, dt)
    ehs = _context_projections(net.up_blocks, ehs, dt)
    skips = list(skips)
    k = 0
    for blk in net.up_blocks:
        for i, r in enumerate(blk.resnets):
            x = _resnet_ckpt(blk, r, x, temb, dt, x1=skips.pop())
            if getattr(blk, "has_cross_attention", False):
                x = _transformer(blk.attentions[i], x, ehs, dt)
            if extras is not None and getattr(blk, "adds_up_states", False):
                x = A.Add.apply(x, extras[k])
            k += 1
            if collect is not None:
                collect.append(x)
        if blk.upsamplers is not None:
            tgt = tuple(skips[-1].shape[1:3])  # the reference's upsample_size (controlnet.py:1129-1130)
            if tgt != (2 * x.shape[1], 2 * x.shape[2]):
                raise NotImplementedError("training path: latent sides must be multiples of 2**num_upsamplers")
            x = _conv(blk.upsamplers[0].conv, A.Up2x.apply(x), dt)
    n = net.conv_norm_out
    h = A.group_norm(x, n.weight, n.bias, n.eps, n.num_groups, True)
    return _conv(net.conv_out, h, dt)


def to_nhwc_grad(t: torch.Tensor, dt, cpad: Optional[int] = None) -> torch.Tensor:
    """NCHW (any strides / dtype, may carry a grad_fn) -> contiguous NHWC in the compute dtype, channels zero padded to
    ``cpad``; plain differentiable torch glue (a zero-copy view when ``t`` is one of this package's NCHW views)."""
    v = t.permute(0, 2, 3, 1)
    if v.dtype != dt:
        v = v.to(dt)
    if cpad is not None and v.shape[-1] != cpad:
        v = torch.nn.functional.pad(v, (0, cpad - v.shape[-1]))
    return v.contiguous()


def _prompt(ehs, B, dt):
    ehs = ehs.to(dt).contiguous()
    if ehs.shape[0] == 1 and B > 1:
        ehs = ehs.expand(B, -1, -1).contiguous()
    return ehs


# The three networks as differentiable functions over NHWC tensors.  ``controlnet.py``'s module ``forward``s route here
# whenever autograd is recording (train/train.py:1324-1354 calls the modules and then ``accelerator.backward(loss)``),
# and ``dual_stream_forward`` below composes them without the NCHW views in between.
def encoder_forward(enc, cond_nhwc, ehs, t_attr, dt, conditioning_scale: float = 1.0):
    """AttributeEncoderModel (controlnet.py:1657-1778).  ``cond_nhwc`` [B,H,W,CIN_PAD].  Returns
    (res[12], mid_res, raw_down[12], raw_mid), NHWC."""
    B, dev = cond_nhwc.shape[0], cond_nhwc.device
    with _batched_casts(enc, dt):
        te = _time(enc, t_attr, B, dt, dev)
        xe = _conv(enc.conv_in, cond_nhwc, dt, cin_pad=CIN_PAD)
        raw_mid_enc, raw_enc = _down_mid(enc, xe, te, ehs, dt)
        res = [A.linear(s, _w2(z, dt), z.bias) for s, z in zip(raw_enc, enc.controlnet_down_blocks)]
        mid_res = A.linear(raw_mid_enc, _w2(enc.controlnet_mid_block, dt), enc.controlnet_mid_block.bias)
    if conditioning_scale != 1.0:  # ref 1773-1775
        res = [r * conditioning_scale for r in res]
        mid_res = mid_res * conditioning_scale
    return res, mid_res, raw_enc, raw_mid_enc


def unet_forward(unet, x_nhwc, ehs, t_img, dt, res=None, mid_res=None, collect_up: bool = False):
    """UNet2DConditionModel (controlnet.py:781-1166).  ``x_nhwc`` [B,H,W,CIN_PAD]; ``res`` / ``mid_res``: the encoder's
    residuals (NHWC) or None.  Returns (img_pred, raw_down[12], raw_mid, up_res[13] | None), NHWC."""
    B, dev = x_nhwc.shape[0], x_nhwc.device
    with _batched_casts(unet, dt):
        tu = _time(unet, t_img, B, dt, dev)
        xu = _conv(unet.conv_in, x_nhwc, dt, cin_pad=CIN_PAD)
        raw_mid_unet, raw_unet = _down_mid(unet, xu, tu, ehs, dt)
        skips, mid = raw_unet, raw_mid_unet
        if res is not None:  # ref 1078-1087, 1114-1115
            skips = [A.Add.apply(s, r) for s, r in zip(raw_unet, res)]
            mid = A.Add.apply(raw_mid_unet, mid_res)
        ups = [mid] if collect_up else None
        img = _up_out(unet, mid, skips, tu, ehs, dt, collect=ups)
    return img, raw_unet, raw_mid_unet, ups


def decoder_forward(dec, raw_mid_enc, raw_enc, ehs, t_attr, dt, raw_unet=None, raw_mid_unet=None, extras=None):
    """AttributeDecoderModel (controlnet.py:2342-2527): exchange skip_enc + conv1x1(skip_unet) (2446-2461, 2476-2477),
    up path on its own weights.  All NHWC; returns attr_pred [B,H,W,out_channels]."""
    B, dev = raw_mid_enc.shape[0], raw_mid_enc.device
    with _batched_casts(dec, dt):
        td = _time(dec, t_attr, B, dt, dev)
        dskips = list(raw_enc)
        if raw_unet is not None:
            dskips = [A.linear(u, _w2(z, dt), z.bias, res=e) for u, e, z in zip(raw_unet, raw_enc, dec.control_down_blocks)]
        xd = A.linear(raw_mid_unet, _w2(dec.control_mid_block, dt), dec.control_mid_block.bias, res=raw_mid_enc)
        return _up_out(dec, xd, dskips, td, ehs, dt, extras=extras)


def dual_stream_forward(unet, enc, dec, x_t, cond, ehs, t_img, t_attr, dtype=torch.bfloat16, run_decoder: bool = True,
                        cond_nhwc: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Differentiable dual-stream step (the call pattern of train.py:1324-1354): NCHW inputs, NHWC predictions
    ``img_pred`` [B,H,W,4] / ``attr_pred`` [B,H,W,28] in the compute dtype.  ``cond_nhwc`` ([B,H,W,28], compute dtype,
    may require grad) replaces ``cond``: the cycle-consistency pass feeds the decoder's own prediction back in."""
    B = x_t.shape[0]
    dt = dtype
    ehs = _prompt(ehs, B, dt)
    if cond_nhwc is not None:
        cin = torch.nn.functional.pad(cond_nhwc, (0, CIN_PAD - cond_nhwc.shape[-1])).contiguous()
    else:
        cin = ops.to_nhwc(cond, dt, CIN_PAD)
    res, mid_res, raw_enc, raw_mid_enc = encoder_forward(enc, cin, ehs, t_attr, dt)
    img, raw_unet, raw_mid_unet, _ = unet_forward(unet, ops.to_nhwc(x_t, dt, CIN_PAD), ehs, t_img, dt, res, mid_res)
    out = {"img_pred": img}
    if run_decoder:
        out["attr_pred"] = decoder_forward(dec, raw_mid_enc, raw_enc, ehs, t_attr, dt, raw_unet, raw_mid_unet)
    return out


def mse_losses(out: Dict[str, torch.Tensor], target_img: torch.Tensor, target_attr: Optional[torch.Tensor]) -> torch.Tensor:
    """The x0-prediction losses of train.py:1356-1365: mean squared error of both predictions against the clean
    latents (targets NCHW fp32)."""
    loss = torch.mean((out["img_pred"].float() - target_img.permute(0, 2, 3, 1).float()) ** 2)
    if "attr_pred" in out and target_attr is not None:
        loss = loss + torch.mean((out["attr_pred"].float() - target_attr.permute(0, 2, 3, 1).float()) ** 2)
    return loss


def reference_losses(nets: Sequence[torch.nn.Module], batch: Dict[str, torch.Tensor], dtype=torch.bfloat16,
                     inverse: bool = True) -> Dict[str, torch.Tensor]:
    """The losses of train/train.py:1356-1413 on one micro-batch.

    batch: ``x_t`` [B,4,h,w] noisy image latent, ``cond`` [B,28,h,w] = clean mask latent (4) | noisy attribute latents
    (24), ``ehs``, ``t_img``, ``t_attr``, ``target_img`` [B,4,h,w], ``target_attr`` [B,24,h,w]; for the inverse-rendering
    branch also ``x_t_c`` / ``t_img_c`` (a fresh noising of the image latent).

    loss = mse(img) + 10 mse(attr) + 0.01 contrastive (albedo of samples 0 / 1 positive; material, specular negative), or
    -- inverse rendering -- mse(img) + mse(attr) + 0.8 mse(img_c), where img_c is the UNet's prediction when the encoder
    is conditioned on [mask | the decoder's OWN attribute prediction] at t_attr = 0 (gradient flows through it)."""
    unet, enc, dec = nets
    F = torch.nn.functional
    out = dual_stream_forward(unet, enc, dec, batch["x_t"], batch["cond"], batch["ehs"], batch["t_img"], batch["t_attr"],
                              dtype=dtype)
    nhwc = lambda t: t.permute(0, 2, 3, 1).float()
    mask_pred = out["attr_pred"][..., 4:]  # rule out the clean mask group (train.py:1354)
    loss_img = F.mse_loss(out["img_pred"].float(), nhwc(batch["target_img"]))
    loss_mask = F.mse_loss(mask_pred.float(), nhwc(batch["target_attr"]))
    res = {"loss_img": loss_img, "loss_mask": loss_mask}
    if not inverse:
        cos = lambda a: F.cosine_similarity(a[0].reshape(-1).float(), a[1].reshape(-1).float(), dim=0) / 0.1
        m_dis, a_dis, s_dis = cos(mask_pred[..., :4]), cos(mask_pred[..., 8:12]), cos(mask_pred[..., 12:16])
        pos = torch.exp(a_dis)
        res["contrastive"] = -torch.log(pos / (pos + torch.exp(m_dis) + torch.exp(s_dis)))
        res["loss"] = loss_img + loss_mask * 10.0 + res["contrastive"] * 0.01
        return res
    mask_lat = batch["cond"][:, :4].permute(0, 2, 3, 1).to(dtype)
    cond_c = torch.cat((mask_lat, mask_pred), dim=-1)
    B = batch["x_t"].shape[0]
    out_c = dual_stream_forward(unet, enc, dec, batch["x_t_c"], None, batch["ehs"], batch["t_img_c"],
                                torch.zeros(B, device=batch["x_t"].device), dtype=dtype, run_decoder=False, cond_nhwc=cond_c)
    res["loss_c"] = F.mse_loss(out_c["img_pred"].float(), nhwc(batch["target_img"]))
    res["loss"] = loss_img + loss_mask + 0.8 * res["loss_c"]
    return res


def _forward_backward(nets, batch, optimizer, buckets, dtype, inverse, grad_accum=None, loss_scale=None, scaler=None):
    """forward, losses, backward; gradients end in ``buckets``' flat buffers (collectives of complete buckets already in
    flight when the buckets were built with ``overlap=True``) or in fresh ``p.grad`` tensors.
    ``grad_accum`` = (k, n): micro-step k of n of one optimisation step (train.py:1236 ``accelerator.accumulate``):
    gradients are zeroed only at k = 0, the loss is divided by n (accelerate's ``backward``), and with ``buckets`` every
    micro-step but the last runs under ``no_sync()``.  ``loss_scale``: a factor (device scalar or float) the loss is
    multiplied with before ``backward()``; ``scaler``: a ``torch.amp.GradScaler`` whose ``scale(loss)`` is used instead
    (train.sh:21 ``--mixed_precision="fp16"``: accelerate's ``backward`` does exactly that)."""
    unet, enc, dec = nets
    k, n = grad_accum if grad_accum is not None else (0, 1)
    if inverse is None:
        out = dual_stream_forward(unet, enc, dec, batch["x_t"], batch["cond"], batch["ehs"], batch["t_img"],
                                  batch["t_attr"], dtype=dtype)
        loss = mse_losses(out, batch["target_img"], batch["target_attr"])
    else:
        loss = reference_losses(nets, batch, dtype=dtype, inverse=inverse)["loss"]
    if k == 0:
        if buckets is not None:
            buckets.zero_grad()  # gradients live in the buckets' flat buffers (views): zero in place, keep the views
        elif optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
    # per-launch sums of squares of the fp32 gradients written by this backward (clipping norm).  They describe the
    # gradients only if this backward produces them from scratch: one micro-step, no loss scaling, and no parameter of
    # these networks holding an older gradient (an optimizer over a subset of the parameters zeroes only its own)
    fresh = (n == 1 and loss_scale is None and scaler is None and buckets is None
             and all(p.grad is None for p in _parameters(nets)))
    B.grad_squares.begin(token=_nets_token(nets), trusted=fresh)
    B.wgrad_queue.reset()
    report = loss.detach()
    if n > 1:
        loss = loss / n
    if loss_scale is not None:
        loss = loss * loss_scale
    if scaler is not None:
        loss = scaler.scale(loss)
    if buckets is not None and k + 1 < n:
        with buckets.no_sync():
            loss.backward()
    else:
        loss.backward()  # with ``buckets``: complete buckets are all-reduced (async) while the backward is still running
    B.wgrad_queue.flush()  # nothing pending after a complete backward (CastParams / ParamBarrier flush); a partial graph may leave some
    if buckets is not None and getattr(buckets, "direct_write", False) and not buckets._hooks:
        # no hooks adopted stray gradients during the backward (overlap=False / gloo): do it here, inside a capture if there is one
        buckets.adopt_all()
    return report


_param_cache: Dict[tuple, tuple] = {}


def _nets_token(nets) -> tuple:
    return tuple(id(n) for n in nets)


def _parameters(nets) -> list:
    """All parameters of the networks, cached per network triple: walking ``Module.parameters()`` of ~700 tensors costs
    ~7 ms of host time, at the tail of a host-bound eager step every time it is done."""
    key = tuple(id(n) for n in nets)
    hit = _param_cache.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[0], nets)):  # (the networks themselves, their parameters)
        hit = _param_cache[key] = (tuple(nets), [p for n in nets for p in n.parameters()])
    return hit[1]


def _clip_and_update(nets, optimizer, buckets, max_grad_norm, stats, scaler=None):
    """train.py:1422-1425: clip_grad_norm_ + optimizer.step(), the clipping folded into the optimizer pass when it can.
    With ``scaler`` (fp16 AMP, train.sh:21): the reference's order -- ``unscale_`` (which also looks for inf / nan),
    clip, ``scaler.step`` (skips the update when an overflow was found: parameters, moments and the step counter stay),
    ``scaler.update``."""
    if scaler is not None:
        if optimizer is None:
            raise ValueError("a GradScaler needs the optimizer whose gradients it unscales")
        scaler.unscale_(optimizer)
        if max_grad_norm is not None:
            if buckets is not None:
                stats["grad_norm"] = buckets.clip_grad_norm_(max_grad_norm)
            else:
                params = [p for p in _parameters(nets) if p.grad is not None]
                stats["grad_norm"] = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
        scaler.step(optimizer)
        scaler.update()
        return
    folded = False
    if max_grad_norm is not None:
        # torch's fused Adam / AdamW kernels (and optim.FusedAdamW) divide every gradient by ``optimizer.grad_scale`` on
        # the fly (the hook GradScaler uses): handing them 1 / clip_coefficient applies train.py:1422-1424's clipping
        # inside the optimizer pass instead of a separate read-modify-write sweep over 7 GB of gradients
        fold = (optimizer is not None and getattr(optimizer, "_step_supports_amp_scaling", False)
                and all(g.get("fused") for g in optimizer.param_groups))
        if fold:
            if buckets is not None and getattr(buckets, "sumsq_parts", None):
                # finish() converted the reduced half-precision buckets back to fp32 and summed the squares in the same pass
                norm = torch.cat(buckets.sumsq_parts).sum().sqrt()
                buckets.sumsq_parts = None
            elif buckets is not None:
                # _foreach_norm, not vector_norm per bucket: a CAPTURED vector_norm over a 32 MB tensor returns wrong
                # values on replay in this torch / ROCm build (tools/graph_norm_repro.py)
                norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(list(buckets.flat))))
            elif B.FUSED_GRADNORM and B.grad_squares.usable(_nets_token(nets)):
                # the big gradients' sums of squares came out of the kernels that wrote them; only the parameters those
                # launches do not cover (norm / bias vectors, a few small matrices) are swept here
                cov = B.grad_squares.count
                rest = [p.grad for p in _parameters(nets) if p.grad is not None and id(p) not in cov]
                sq = torch.cat(B.grad_squares.parts).sum()
                if rest:
                    sq = sq + torch.stack(torch._foreach_norm(rest)).square().sum()
                norm = sq.sqrt()
                B.grad_squares.clear()  # consumed: stale partial sums must not describe a later step
            else:
                grads = [p.grad for p in _parameters(nets) if p.grad is not None]
                norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
            stats["grad_norm"] = norm
            optimizer.grad_scale = torch.clamp((norm + 1e-6) / max_grad_norm, min=1.0).to(torch.float32).reshape(())
            optimizer.found_inf = torch.zeros((), dtype=torch.float32, device=norm.device)
            folded = True
        elif buckets is not None:
            stats["grad_norm"] = buckets.clip_grad_norm_(max_grad_norm)
        else:
            params = [p for p in _parameters(nets) if p.grad is not None]
            stats["grad_norm"] = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    if optimizer is not None:
        optimizer.step()
        if folded:
            del optimizer.grad_scale, optimizer.found_inf


def train_step(nets: Sequence[torch.nn.Module], batch: Dict[str, torch.Tensor], optimizer=None, buckets=None,
               dtype=torch.bfloat16, max_grad_norm: Optional[float] = 1.0, inverse: Optional[bool] = None,
               as_tensors: bool = False, grad_accum: Optional[tuple] = None, scaler=None) -> Dict[str, float]:
    """One optimisation step: forward, losses, backward (HIP kernels), gradient all-reduce (``buckets``: a
    parallel.GradientBuckets, no-op on one rank), clipping (train.py:1422-1424), optimizer step.
    ``grad_accum`` = (k, n): micro-step k of n (``accelerator.accumulate``, train.py:1236): gradients are zeroed at
    k = 0 only, the loss is divided by n, the collectives / clipping / update happen at k = n - 1 only.
    ``scaler``: a ``torch.amp.GradScaler`` for the reference's fp16 AMP recipe (train.sh:21; pass ``dtype=torch.float16``).
    ``inverse``: None = the plain two-stream MSE objective (mse_losses); True / False = the reference's inverse-rendering
    (cycle consistency) / rendering (contrastive) objectives (reference_losses).  Ranks may pick different branches
    (compute_t, train.py:445): parameters without a gradient contribute zeros to the buckets.
    ``as_tensors``: return the statistics as device tensors (no host synchronisation: the step can then be captured
    into a HIP graph, tools/train_bench.py --graph)."""
    stats = {"loss": _forward_backward(nets, batch, optimizer, buckets, dtype, inverse, grad_accum=grad_accum, scaler=scaler)}
    last = grad_accum is None or grad_accum[0] + 1 == grad_accum[1]
    if buckets is not None:
        if last:
            buckets.finish()
        else:
            with buckets.no_sync():
                buckets.finish()  # an accumulation micro-step: only adopts gradients re-created outside the buckets
    if last:
        _clip_and_update(nets, optimizer, buckets, max_grad_norm, stats, scaler=scaler)
    return stats if as_tensors else {k: float(v) for k, v in stats.items()}


class GraphedTrainStep:
    """The training step as HIP-graph replays, also on several GPUs.

    Eagerly the step is host-bound (~9700 launches issued from Python: 118 ms against 90 ms of GPU work at cfg 4's
    per-GPU shape), which is also what bounds the data-parallel step when the gradient collectives are launched from
    autograd hooks.  Here forward + losses + backward are ONE captured graph writing the gradients into the flat bucket
    buffers of ``buckets`` (built with ``overlap=False``: no hooks, nothing but kernels inside the capture); the
    collectives run eagerly on those few flat buffers (``buckets.finish()``: one all-reduce or reduce-scatter +
    all-gather per 256 MB bucket, bf16 on the wire if asked), and clipping + the optimizer step are a second graph.
    Without ``buckets`` (one GPU) the whole step is a single graph.  One instance per objective branch (``inverse``):
    the branch is a host decision (train.py:445), so a training loop keeps one instance per branch over the same
    networks and optimizer.  ``warmup`` eager steps run first (REAL optimisation steps on ``batch``: allocator pools, lazy
    optimizer state); 0 is allowed with optimizers whose state can be created up front (optim.FusedAdamW).
    The optimizer's lr / betas / eps / weight_decay are captured by value; ``step()`` notices when a scheduler or a resumed
    checkpoint changed them and captures again."""

    def __init__(self, nets, batch: Dict[str, torch.Tensor], optimizer, buckets=None, dtype=torch.bfloat16,
                 max_grad_norm: Optional[float] = 1.0, inverse: Optional[bool] = None, warmup: int = 2,
                 capture_collectives: Optional[bool] = None):
        """``capture_collectives`` (RCCL only; default: on when ``buckets`` were built with ``overlap=True`` on the nccl
        backend): the bucket collectives are captured INTO the step graph.  The post-accumulate hooks of ``buckets`` enqueue
        each bucket's all-reduce / reduce-scatter + all-gather on RCCL's stream the moment the bucket is complete -- inside a
        capture that is a fork: the collective becomes a parallel branch of the graph that runs beside the rest of the
        backward, joined by ``buckets.finish()`` in front of clipping + update.  With the weight gradients deferred to the end
        of each network's backward (backward.WgradQueue) the buckets complete network by network: the decoder's collectives
        overlap the UNet's backward, the UNet's the encoder's -- DDP's overlap at train/train.py:1421, with the whole step
        ONE replay and no host in the loop.  Without it (gloo, or ``overlap=False`` buckets): forward + backward graph,
        eager collectives, update graph, serially."""
        import torch.distributed as dist
        hooks = bool(buckets is not None and getattr(buckets, "_hooks", None))
        nccl = bool(buckets is not None and dist.is_initialized() and dist.get_backend(getattr(buckets, "group", None)) == "nccl")
        if capture_collectives is None:
            capture_collectives = hooks and nccl
        if capture_collectives and not (hooks and nccl):
            raise ValueError("capture_collectives needs GradientBuckets(..., overlap=True) on the nccl (RCCL) backend")
        if hooks and not capture_collectives:
            raise ValueError("GraphedTrainStep without captured collectives needs GradientBuckets(..., overlap=False): "
                             "hooks launching eager collectives cannot run inside a capture")
        self.capture_collectives = bool(capture_collectives)
        if self.capture_collectives:  # RCCL creates its communicator on first use: that must not happen inside a capture
            t = torch.zeros(8, device=buckets.flat[0].device)
            dist.all_reduce(t, group=buckets.group)
            torch.cuda.synchronize()
        self.nets, self.optimizer, self.buckets = nets, optimizer, buckets
        # gloo (the CPU-side test transport) stages device tensors through the host; handing it a tensor whose producer
        # graph is still replaying was measured at 50-140 s per step with two ranks on one GPU (0.03 s after a stream
        # synchronise).  RCCL collectives are stream-ordered and need no host wait.
        self._host_sync_before_collectives = bool(buckets is not None and dist.is_initialized()
                                                  and dist.get_backend(getattr(buckets, "group", None)) == "gloo")
        self.batch = {k: v.clone() for k, v in batch.items()}
        self.stats: Dict[str, torch.Tensor] = {}
        kw = dict(dtype=dtype, inverse=inverse)
        if warmup == 0:
            # nothing may be created lazily inside the capture (a zero-fill captured there would re-run on every replay)
            init = getattr(optimizer, "_init_state", None)
            if init is None:
                raise ValueError("warmup=0 needs an optimizer that can create its state eagerly (optim.FusedAdamW)")
            for grp in optimizer.param_groups:
                for p in grp["params"]:
                    if p.requires_grad:
                        init(p)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # allocator pools, lazy optimizer state, the tuning-table lookups
                self._eager(kw, max_grad_norm)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.max_grad_norm = max_grad_norm
        self._kw = kw
        self._capture()

    def _hyper(self):
        """The optimizer hyper-parameters the captured update holds BY VALUE (kernel arguments of ``ur_adamw_multi``).
        lr and weight_decay are NOT among them when the optimizer keeps them in device memory (optim.FusedAdamW.sync_hyper):
        a scheduler step then only needs that copy refreshed before the replay."""
        dev_lr = hasattr(self.optimizer, "sync_hyper")
        # ... and the ADDRESSES it holds: the device (lr, weight_decay) pair and the optimizer's ``generation`` (bumped when
        # load_state_dict / add_param_group replaced state tensors a captured update points at)
        return (getattr(self.optimizer, "generation", 0),) + tuple(
            ((None if dev_lr else float(g["lr"])), tuple(float(b) for b in g["betas"]), float(g["eps"]),
             (None if dev_lr else float(g["weight_decay"])),
             (g["_ur_hyper"][0].data_ptr() if g.get("_ur_hyper") is not None else None)) for g in self.optimizer.param_groups)

    def _capture(self):
        nets, optimizer, buckets, kw, max_grad_norm = self.nets, self.optimizer, self.buckets, self._kw, self.max_grad_norm
        if hasattr(optimizer, "sync_hyper"):
            optimizer.sync_hyper()
        self._captured_hyper = self._hyper()
        self.g_fb = torch.cuda.CUDAGraph()
        self.g_up = None
        if self.capture_collectives:
            buckets._reset()
            before = buckets.launched_from_hooks
            with torch.cuda.graph(self.g_fb):
                # the hooks fire inside backward(): every complete bucket's collective forks onto RCCL's stream here
                self.stats["loss"] = _forward_backward(nets, self.batch, None, buckets, **kw)
                self.collectives_from_hooks = buckets.launched_from_hooks - before  # forked before finish() had to
                buckets.finish()  # launches what the hooks could not, then joins RCCL's stream (work.wait = stream wait)
                _clip_and_update(nets, optimizer, buckets, max_grad_norm, self.stats)
            self._validate_captured_collectives()
            return
        with torch.cuda.graph(self.g_fb):
            self.stats["loss"] = _forward_backward(nets, self.batch, optimizer if buckets is None else None, buckets, **kw)
            if buckets is None:
                _clip_and_update(nets, optimizer, None, max_grad_norm, self.stats)
        if buckets is not None:
            buckets._reset()
            self.g_up = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_up):
                _clip_and_update(nets, optimizer, buckets, max_grad_norm, self.stats)

    def _validate_captured_collectives(self):
        """The split "forked by a hook during the backward" / "launched by finish()" is baked into the graph at capture time.
        Buckets always launch in index order (GradientBuckets._launch), so the SEQUENCE of collectives is the same on every
        rank by construction; what could differ is where they sit in each rank's graph -- harmless for correctness (RCCL
        matches them by order on its own stream), but a rank whose buckets all fell to finish() serialises every other rank's
        overlap behind its backward.  Cross-check the count once per capture and refuse a mixed capture: a
        parameter that got no gradient on one rank during the capture step would otherwise go unnoticed for the whole run."""
        import torch.distributed as dist
        b = self.buckets
        if not dist.is_initialized() or dist.get_world_size(b.group) < 2:
            return
        n = float(self.collectives_from_hooks)
        t = torch.tensor([n, -n], dtype=torch.float32, device=b.flat[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=b.group)
        hi, lo = float(t[0]), -float(t[1])
        if hi != lo:
            raise RuntimeError(f"captured bucket collectives differ across ranks: between {int(lo)} and {int(hi)} of {len(b.buckets)} "
                               "buckets were launched from the backward's hooks (a parameter without a gradient on some rank "
                               "during the capture step?); capture with a batch that reaches every parameter on every rank")

    def _ensure_current(self):
        """Refresh the device (lr, weight_decay) pair and capture again if a hyper-parameter held by value, or an address the
        captured update points at, changed since the capture."""
        if hasattr(self.optimizer, "sync_hyper"):
            self.optimizer.sync_hyper()  # lr / weight_decay of an lr_scheduler step -> the device pair the kernel reads
        if self._hyper() != self._captured_hyper:
            # betas / eps changed (resume_from_checkpoint with other settings), or an optimizer that holds lr by value:
            # the captured update has the old values as kernel arguments.  Capture again (tens of ms, once per change)
            torch.cuda.synchronize()
            self._capture()

    def _eager(self, kw, max_grad_norm):
        st = {"loss": _forward_backward(self.nets, self.batch, self.optimizer, self.buckets, **kw)}
        if self.buckets is not None:
            self.buckets.finish()
        _clip_and_update(self.nets, self.optimizer, self.buckets, max_grad_norm, st)

    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """One optimisation step on ``batch`` (copied into the captured input buffers; None: the last one again).
        Returns device tensors (loss, grad_norm): reading them is the caller's synchronisation."""
        if batch is not None:
            for k, v in batch.items():
                self.batch[k].copy_(v)
        self._ensure_current()
        self.g_fb.replay()
        if self.buckets is not None and not self.capture_collectives:
            if self._host_sync_before_collectives:
                torch.cuda.current_stream().synchronize()
            self.buckets.finish()
            self.g_up.replay()  # clipping (torch._foreach_norm over the buckets) + optimizer step
        return self.stats

    def phase_times(self) -> Dict[str, float]:
        """One step with HIP events between the phases (synchronises): milliseconds of the forward + backward replay, of the
        bucket collectives (launch to completion, as the compute stream sees them) and of clipping + update.  With captured
        collectives the step is one graph and only its total is observable from outside (``overlapped_total``)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self._ensure_current()  # the same staleness check as step(): never replay an update holding freed addresses
        ev[0].record()
        self.g_fb.replay()
        ev[1].record()
        if self.buckets is not None and not self.capture_collectives:
            if self._host_sync_before_collectives:
                torch.cuda.current_stream().synchronize()
            self.buckets.finish()
            ev[2].record()
            self.g_up.replay()
        else:
            ev[2].record()
        ev[3].record()
        torch.cuda.synchronize()
        if self.capture_collectives or self.buckets is None:
            return dict(overlapped_total=ev[0].elapsed_time(ev[3]))
        return dict(replay=ev[0].elapsed_time(ev[1]), collectives=ev[1].elapsed_time(ev[2]), update=ev[2].elapsed_time(ev[3]))
