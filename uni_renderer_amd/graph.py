"""The dual-stream denoise step as one replayable HIP graph with two concurrent branches.

A step is ~870 kernel launches (enc + unet + dec at SD-1.x size), most of them small (a few hundred workgroups,
latency-bound on a 256-CU part).  Two MI355X-side measures:

  * the step is captured once into a hipGraph (through PyTorch's capture of the current stream -- the C-ABI
    launches land on that stream) and replayed, which removes the host from the loop;
  * the two diffusion streams are independent for most of the step -- the attribute ENCODER does not depend on
    the UNet's down path + mid block, and the attribute DECODER does not depend on the UNet's up path -- so they
    are enqueued on two HIP streams (fork/join with events) and become parallel branches of the graph: kernels
    of the two branches share the GPU and fill each other's idle CUs.

            main:  unet.conv_in/down/mid ----+--> (+enc residuals) unet.up/conv_out ---+--> join
            side:  enc (down/mid/zero-convs) -+--> dec (exchange 1x1 + up/conv_out) ----+

Call pattern reproduced: ``enc -> unet -> dec`` of models/pipeline.py:2660-2690 (inverse rendering) and
train/train.py:1324-1354; ``run_decoder=False`` gives the rendering direction (pipeline.py:1611-1629).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def dual_stream_step(unet, enc, dec, x_t, cond, ehs, t_img, t_attr, run_decoder: bool = True,
                     side: Optional[torch.cuda.Stream] = None, conditioning_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """One step: the same calls the reference's loops make (``conditioning_scale`` goes to the encoder like
    pipeline.py:2660-2667 passes ``cond_scale``).  With ``side`` (a second HIP stream) the encoder and
    the decoder run concurrently with the UNet's down and up halves; without it everything is serial."""
    if side is None:
        res, mid, raw_enc, raw_mid_enc = enc(x_t, t_attr, encoder_hidden_states=ehs, controlnet_cond=cond,
                                             conditioning_scale=conditioning_scale, return_dict=False)
        img_pred, raw_unet, raw_mid_unet, _ = unet(
            x_t, t_img, encoder_hidden_states=ehs, down_block_additional_residuals=res,
            mid_block_additional_residual=mid, return_dict=False)
        out = {"img_pred": img_pred}
        if run_decoder:
            out["attr_pred"] = dec(sample=raw_mid_enc, down_block_res_samples=raw_enc, timestep=t_attr,
                                   encoder_hidden_states=ehs, down_block_additional_residuals=raw_unet,
                                   mid_block_additional_residual=raw_mid_unet, return_dict=False)
        return out

    main = torch.cuda.current_stream()
    side.wait_stream(main)  # fork: inputs are ready
    with torch.cuda.stream(side):
        res, mid, raw_enc, raw_mid_enc = enc(x_t, t_attr, encoder_hidden_states=ehs, controlnet_cond=cond,
                                             conditioning_scale=conditioning_scale, return_dict=False)
    state = unet.forward_down_mid(x_t, t_img, ehs)  # main, concurrent with the encoder
    raw_unet = tuple(ops.as_nchw_view(s) for s in state["raw_down"])
    raw_mid_unet = ops.as_nchw_view(state["raw_mid"])
    main.wait_stream(side)  # the UNet's up half needs the encoder's residuals
    for t in list(res) + [mid]:
        t.record_stream(main)
    out = {}
    if run_decoder:
        side.wait_stream(main)  # the decoder needs the UNet's raw skips (everything enqueued on main so far)
        for t in raw_unet + (raw_mid_unet,):
            t.record_stream(side)
        with torch.cuda.stream(side):
            out["attr_pred"] = dec(sample=raw_mid_enc, down_block_res_samples=raw_enc, timestep=t_attr,
                                   encoder_hidden_states=ehs, down_block_additional_residuals=raw_unet,
                                   mid_block_additional_residual=raw_mid_unet, return_dict=False)
    out["img_pred"] = unet.forward_up(state, res, mid, return_dict=False)[0]  # main, concurrent with the decoder
    main.wait_stream(side)  # join
    if run_decoder:
        out["attr_pred"].record_stream(main)
    return out


class GraphedDualStreamStep:
    """Capture once, replay many times.  ``step(...)`` copies new inputs into the static buffers (device to
    device, on the replay stream) and launches the graph; the returned tensors are overwritten by the next call."""

    def __init__(self, unet, enc, dec, batch: int, latent_hw, cross_dim: int, dtype=torch.float16,
                 device="cuda", run_decoder: bool = True, cond_channels: int = 28, img_channels: int = 4,
                 ctx_len: int = 77, concurrent: bool = True, mode: Optional[str] = None,
                 conditioning_scale: float = 1.0):
        """mode: "grouped" (default: the two streams as one grouped launch per op, fused.py), "concurrent" (module
        path, two graph branches on two HIP streams), "serial" (module path, one stream).
        ``conditioning_scale``: the encoder's residual scale (pipeline.py:2660-2667), baked into the captured graph."""
        self.unet, self.enc, self.dec, self.run_decoder = unet, enc, dec, run_decoder
        self.conditioning_scale = float(conditioning_scale)
        self.mode = mode or ("grouped" if concurrent else "serial")
        if self.mode not in ("grouped", "concurrent", "serial"):
            raise ValueError(self.mode)
        self._grouped = None
        H, W = (latent_hw, latent_hw) if isinstance(latent_hw, int) else latent_hw
        dev = torch.device(device)
        self.x_t = torch.zeros(batch, img_channels, H, W, dtype=dtype, device=dev)
        self.cond = torch.zeros(batch, cond_channels, H, W, dtype=dtype, device=dev)
        self.ehs = torch.zeros(batch, ctx_len, cross_dim, dtype=dtype, device=dev)
        self.t_img = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.t_attr = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.side = torch.cuda.Stream(device=dev) if self.mode == "concurrent" else None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Optional[Dict[str, torch.Tensor]] = None

    def _run(self):
        if self.mode == "grouped":
            if self._grouped is None:
                from .fused import GroupedDualStreamStep

                self._grouped = GroupedDualStreamStep(self.unet, self.enc, self.dec)
            return self._grouped(self.x_t, self.cond, self.ehs, self.t_img, self.t_attr, self.run_decoder,
                                 conditioning_scale=self.conditioning_scale)
        return dual_stream_step(self.unet, self.enc, self.dec, self.x_t, self.cond, self.ehs, self.t_img,
                                self.t_attr, self.run_decoder, side=self.side, conditioning_scale=self.conditioning_scale)

    def load_inputs(self, x_t, cond, ehs, t_img, t_attr):
        self.x_t.copy_(x_t)
        self.cond.copy_(cond)
        self.ehs.copy_(ehs)
        self.t_img.copy_(torch.as_tensor(t_img, device=self.t_img.device).float().expand_as(self.t_img))
        self.t_attr.copy_(torch.as_tensor(t_attr, device=self.t_attr.device).float().expand_as(self.t_attr))

    @torch.no_grad()
    def capture(self, warmup: int = 2):
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):  # packs weights, sizes LDS attributes, fills the allocator
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        torch.cuda.synchronize()
        return self

    @torch.no_grad()
    def capture_with(self, post_fn):
        """A second graph over the same static buffers: the step followed by ``post_fn(out)`` -- e.g. the on-device
        sampler update that turns the prediction into the next step's input (pipeline.py), so that a sampling loop is
        nothing but replays.  Returns ``(graph, out)``."""
        if self.graph is None:
            self.capture()  # warm-up, weight packing
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._run()
            post_fn(out)
        torch.cuda.synchronize()
        return g, out

    @torch.no_grad()
    def step(self, x_t=None, cond=None, ehs=None, t_img=None, t_attr=None):
        if self.graph is None:
            self.capture()
        if x_t is not None:
            self.load_inputs(x_t, cond, ehs, t_img, t_attr)
        self.graph.replay()
        return self.out

    def replay(self):
        self.graph.replay()


class GraphedHoistedStep:
    """A sampling loop's step with the loop-invariant half hoisted (hoist.HoistedSamplingStep) as TWO replayable graphs over
    one set of static buffers: ``begin()`` replays the prologue (once per sampling call, after ``load_inputs``), ``replay()``
    the per-step part.  Same buffer names as ``GraphedDualStreamStep`` (``x_t``, ``cond``, ``ehs``, ``t_img``, ``t_attr``);
    inverse direction (``run_decoder``): ``x_t`` / ``t_img`` / ``ehs`` are the fixed inputs and ``cond`` / ``t_attr`` evolve,
    rendering direction: the reverse.  ``hoist=False`` replays the prologue in front of EVERY step -- the un-hoisted loop
    with the same kernels, what the hoisting is tested against bit for bit."""

    def __init__(self, unet, enc, dec, batch: int, latent_hw, cross_dim: int, dtype=torch.float16, device="cuda",
                 run_decoder: bool = True, cond_channels: int = 28, img_channels: int = 4, ctx_len: int = 77,
                 conditioning_scale: float = 1.0, hoist: bool = True, leaves=None):
        """``leaves``: a ``fused.GroupedDualStreamStep`` whose packed-weight cache is shared (tools: many captures of one model)."""
        from .hoist import HoistedSamplingStep

        self.run_decoder, self.hoist = run_decoder, hoist
        self.h = HoistedSamplingStep(unet, enc, dec, "inverse" if run_decoder else "render", conditioning_scale, leaves=leaves)
        H, W = (latent_hw, latent_hw) if isinstance(latent_hw, int) else latent_hw
        dev = torch.device(device)
        self.x_t = torch.zeros(batch, img_channels, H, W, dtype=dtype, device=dev)
        self.cond = torch.zeros(batch, cond_channels, H, W, dtype=dtype, device=dev)
        self.ehs = torch.zeros(batch, ctx_len, cross_dim, dtype=dtype, device=dev)
        self.t_img = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.t_attr = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.pro: Optional[torch.cuda.CUDAGraph] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Optional[Dict[str, torch.Tensor]] = None
        self._fixed_sig = None
        self.prologue_runs = 0  # diagnostics / tests: how often the loop-invariant half ran

    load_inputs = GraphedDualStreamStep.load_inputs

    def _prologue(self):
        if self.run_decoder:
            self.h.prologue(self.x_t, self.ehs, self.t_img)
        else:
            self.h.prologue(self.cond, self.ehs, self.t_attr)

    def _run(self, tables=None):
        if self.run_decoder:
            return self.h.step(self.cond, self.t_attr, tables=tables)
        return self.h.step(self.x_t, self.t_img, tables=tables)

    @torch.no_grad()
    def capture_time_tables(self, tvals: torch.Tensor):
        """A graph that fills the loop's per-step time-projection tables from the device vector ``tvals`` [n] (the loop's
        timesteps): ``(graph, (temb1_all, temb3_all))``.  Replay it once per sampling call after writing ``tvals``."""
        if self.graph is None:
            self.capture()
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            self.h.time_tables(tvals)  # tile lookups, allocator
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            tabs = self.h.time_tables(tvals)
        torch.cuda.synchronize()
        return g, tabs

    def load_evolving(self, x_t, cond, t_img, t_attr):
        """Only what changes between two steps of a loop: the evolving latent and its timestep."""
        src, dst, t, tb = (cond, self.cond, t_attr, self.t_attr) if self.run_decoder else (x_t, self.x_t, t_img, self.t_img)
        dst.copy_(src)
        tb.copy_(torch.as_tensor(t, device=tb.device).float().expand_as(tb))

    @torch.no_grad()
    def capture(self, warmup: int = 2):
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):  # packs weights, sizes LDS attributes, fills the allocator
            for _ in range(warmup):
                self._prologue()
                self._run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.pro = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.pro):
            self._prologue()  # its results (self.h.inv) live in this graph's pool for as long as this object does
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        torch.cuda.synchronize()
        return self

    @torch.no_grad()
    def capture_with(self, post_fn, tables=None):
        """The per-step graph followed by ``post_fn(out)`` (the on-device sampler update): ``(graph, out)``.  ``tables`` =
        (temb1_all, temb3_all, step counter): the step reads its time projections from per-call tables
        (``capture_time_tables``) at the row of the device-side step counter ``post_fn`` advances."""
        if self.graph is None:
            self.capture()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._run(tables)
            post_fn(out)
        torch.cuda.synchronize()
        return g, out

    def begin(self):
        """Once per sampling call, after the fixed inputs are in the static buffers."""
        if self.graph is None:
            self.capture()
        self.pro.replay()
        self.prologue_runs += 1

    @torch.no_grad()
    def step(self, x_t=None, cond=None, ehs=None, t_img=None, t_attr=None, first: bool = True):
        """``first``: the fixed inputs are (re)loaded and the prologue runs; later steps of the same loop only load the
        evolving latent and its timestep."""
        if self.graph is None:
            self.capture()
        fixed = (x_t, ehs, t_img) if self.run_decoder else (cond, ehs, t_attr)
        if not first and self.hoist and self._fixed_sig is not None and any(f is not None for f in fixed) \
                and self._sig(fixed) != self._fixed_sig:
            # a caller (e.g. a callback_on_step_end) handed over DIFFERENT fixed inputs in mid-loop: the prologue's results
            # are stale -- run it again instead of silently ignoring the change.  The check is by object
            # (storage address, version counter, shape), no device synchronisation.
            first = True
        if first or not self.hoist:
            if x_t is not None:
                self.load_inputs(x_t, cond, ehs, t_img, t_attr)
            self._fixed_sig = self._sig(fixed)
            self.pro.replay()
            self.prologue_runs += 1
        elif x_t is not None or cond is not None:
            self.load_evolving(x_t, cond, t_img, t_attr)
        self.graph.replay()
        return self.out

    @staticmethod
    def _sig(tensors):
        """Identity of the caller's fixed inputs without touching the device: (address, version, shape) per tensor; Python
        numbers by value."""
        sig = []
        for t in tensors:
            if isinstance(t, torch.Tensor):
                sig.append((t.data_ptr(), t._version, tuple(t.shape)))
            else:
                sig.append(t)
        return tuple(sig)

    def replay(self):
        if not self.hoist:
            self.pro.replay()
        self.graph.replay()
