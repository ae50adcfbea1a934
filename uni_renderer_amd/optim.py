"""AdamW for the training loop on one HIP kernel (``ur_adamw_multi``, csrc/backward.hip).

The reference builds ``torch.optim.AdamW`` (or bitsandbytes' 8-bit variant) over the parameters of the three networks in
ONE parameter group (train/train.py:1082-1100: lr, betas, weight_decay, eps from the command line) and calls
``optimizer.step()`` once per batch (1425).  This class is that optimizer with the update of up to 64 parameter tensors
per launch (descriptors as kernel arguments: ~12 launches for the 700 tensors of enc + unet + dec instead of ~95
multi-tensor launches), fp32 state, the arithmetic of torch's fused AdamW kernel.

Interchangeable with ``torch.optim.AdamW``: same constructor arguments, ``param_groups`` and ``state_dict()`` layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so ``optimizer.bin`` of checkpointing.py loads in either), the
``grad_scale`` / ``found_inf`` attributes of torch's AMP-aware fused optimizers (train_step.py folds gradient clipping
into the update through ``grad_scale``), and no host synchronisation in ``step()`` (the step counter is a device
scalar): the whole training step still captures into one HIP graph.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Tuple

import torch

from . import _lib
from ._lib import check
from .ops import _stream

MAX_TENSORS = 64


class _Tensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


class FusedAdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, fused: bool = True, capturable: bool = True):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        # ``fused`` / ``capturable`` are accepted (and recorded) so that code written for torch.optim.AdamW(fused=True,
        # capturable=True) constructs this class unchanged; this implementation is always both
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, fused=True, capturable=True,
                        amsgrad=False, maximize=False, foreach=None, differentiable=False)
        super().__init__(params, defaults)
        # bumped whenever the addresses a captured update may hold could have changed (load_state_dict that had to replace
        # state tensors, add_param_group): train_step.GraphedTrainStep compares it and captures again
        self.generation = 0

    def sync_hyper(self):
        """Copy every group's lr / weight_decay into its device scalar pair (created on first use).  The kernel reads them
        from there, so an ``lr_scheduler`` (train.py --lr_scheduler / --lr_warmup_steps) changes what a CAPTURED update
        does without a re-capture: ``step()`` calls this itself when it runs eagerly; around a graph replay the owner of
        the graph calls it before ``replay()`` (train_step.GraphedTrainStep.step)."""
        for group in self.param_groups:
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                raise NotImplementedError("FusedAdamW: tensor learning rates are not supported")
            want = (float(lr), float(group["weight_decay"]))
            hy = group.get("_ur_hyper")
            if hy is None:
                dev = next((p.device for p in group["params"]), None)
                if dev is None or dev.type != "cuda":
                    continue
                hy = group["_ur_hyper"] = [torch.empty(2, dtype=torch.float32, device=dev), None]
            if hy[1] != want:
                # staged through a pinned pair: an lr_scheduler changes lr on every step, and a blocking copy from pageable
                # memory would host-synchronise the stream each time.  The staging buffer is rewritten only after the
                # previous copy out of it has completed (event), so back-to-back changes never race.
                if len(hy) < 4:
                    hy += [torch.empty(2, dtype=torch.float32).pin_memory(), torch.cuda.Event()]
                    hy[3].record()
                hy[3].synchronize()
                hy[2][0], hy[2][1] = want
                hy[0].copy_(hy[2], non_blocking=True)
                hy[3].record()
                hy[1] = want

    def _init_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()  # inside a capture the device pair must already exist (and is refreshed outside the graph)
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdamW: amsgrad / maximize are not used by the reference (train.py:1093-1100)")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            gptrs = tuple(p.grad.data_ptr() for p in ps)
            cached = group.get("_ur_launches")
            if cached is not None and cached[0] == gptrs and cached[1] == tuple(p.data_ptr() for p in ps):
                # same tensors as last step (gradient views of flat buckets, or a captured graph's addresses): the
                # descriptor arrays are reused -- building them is ~6 ms of host time per step for 700 tensors
                step_t = cached[3]
                step_t += 1
                self._launch(lib, group, cached[2], step_t, grad_scale, found_inf)
                continue
            for p in ps:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdamW updates fp32 master parameters on the GPU (train.py:1082-1089)")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous parameters and gradients")
            states = [self._init_state(p) for p in ps]
            # one device step counter per group: the per-parameter ``step`` entries all alias it after the first call
            step_t = states[0]["step"]
            for st in states[1:]:
                if st["step"] is not step_t:
                    st["step"] = step_t
            step_t += 1
            arrays = []
            for i in range(0, len(ps), MAX_TENSORS):
                chunk = ps[i:i + MAX_TENSORS]
                arr = (_Tensor * len(chunk))()
                for k, p in enumerate(chunk):
                    st = states[i + k]
                    arr[k].p, arr[k].g = p.data_ptr(), p.grad.data_ptr()
                    arr[k].m, arr[k].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    arr[k].n = p.numel()
                arrays.append(arr)
            # the cache holds raw addresses: keep what they point into alive next to it (states live in self.state)
            group["_ur_launches"] = (gptrs, tuple(p.data_ptr() for p in ps), arrays, step_t)
            self._launch(lib, group, arrays, step_t, grad_scale, found_inf)
        return loss

    @staticmethod
    def _launch(lib, group, arrays, step_t, grad_scale, found_inf):
        beta1, beta2 = group["betas"]
        lr = group["lr"]
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("FusedAdamW: tensor learning rates are not supported")
        s_ = _stream()
        hy = group.get("_ur_hyper")
        if hy is None:
            raise RuntimeError("FusedAdamW.step() inside a graph capture before any eager step: call sync_hyper() first")
        for arr in arrays:
            check(lib.ur_adamw_multi(arr, len(arr), float(lr), float(beta1), float(beta2), float(group["eps"]),
                                     float(group["weight_decay"]), step_t.data_ptr(),
                                     grad_scale.data_ptr() if grad_scale is not None else None,
                                     found_inf.data_ptr() if found_inf is not None else None, hy[0].data_ptr(), s_), "ur_adamw_multi")
        if found_inf is not None:
            # the kernel leaves parameters and moments alone when found_inf != 0; like torch's fused AdamW, a skipped
            # step must not advance the bias correction either (a device op: no host synchronisation, capturable)
            step_t -= found_inf.to(step_t.dtype).reshape(())

    def __getstate__(self):
        state = super().__getstate__()
        state["param_groups"] = [{k: v for k, v in g.items() if k not in ("_ur_launches", "_ur_hyper")} for g in state["param_groups"]]
        return state  # the launch cache holds raw device addresses in ctypes arrays: never pickled, rebuilt on the next step

    def state_dict(self):
        """torch.optim.AdamW's layout.  The per-parameter ``step`` entries alias one device scalar inside this object;
        a state dict hands out independent copies (torch's optimizers increment every entry on their own)."""
        sd = super().state_dict()
        for grp in sd["param_groups"]:
            grp.pop("_ur_launches", None)  # host-side launch cache, not optimizer state
            grp.pop("_ur_hyper", None)     # device copy of (lr, weight_decay), rebuilt from the group's values
        # torch hands out the LIVE per-parameter dicts: build copies, never assign into them (replacing the live
        # ``step`` entry would cut its alias to the device counter the cached launches keep incrementing)
        sd["state"] = {k: ({**st, "step": st["step"].clone()} if "step" in st else dict(st)) for k, st in sd["state"].items()}
        return sd

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self.generation = getattr(self, "generation", 0) + 1

    def load_state_dict(self, state_dict):
        """Loads IN PLACE where it can: existing ``exp_avg`` / ``exp_avg_sq`` / step-counter tensors and the device (lr,
        weight_decay) pair keep their addresses and receive the loaded values, so a captured ``ur_adamw_multi`` (HIP-graph
        replays of the training step) keeps reading live memory after ``resume_from_checkpoint`` (train.py:1191-1218).
        Only state that did not exist yet, or changed shape, is adopted as new tensors -- ``generation`` is bumped then, and
        GraphedTrainStep captures again."""
        old_hyper = [g.get("_ur_hyper") for g in self.param_groups]
        old_launch = [g.get("_ur_launches") for g in self.param_groups]
        old_state = {p: dict(self.state[p]) for g in self.param_groups for p in g["params"] if self.state.get(p)}
        super().load_state_dict(state_dict)
        replaced = False
        for gi, group in enumerate(self.param_groups):
            group.pop("_ur_launches", None)
            group.pop("_ur_hyper", None)
            hy = old_hyper[gi] if gi < len(old_hyper) else None
            if hy is not None:
                hy[1] = None  # same device pair, value re-sent by the next sync_hyper()
                group["_ur_hyper"] = hy
            step_t = None
            for p in group["params"]:
                st = self.state.get(p)
                prev = old_state.get(p)
                if prev is not None and not (st and all(k in st for k in ("exp_avg", "exp_avg_sq", "step"))):
                    # live state that the loaded dict lacks (a checkpoint from before the first step, a partial state):
                    # torch dropped the old moments, so the cached descriptor arrays / a captured ur_adamw_multi would
                    # keep the addresses of freed tensors -- rebuild them
                    replaced = True
                if not st:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st and prev is not None and k in prev and prev[k].shape == st[k].shape and prev[k].dtype == torch.float32:
                        prev[k].copy_(st[k])
                        st[k] = prev[k]
                    elif k in st:
                        replaced = True
                if "step" in st:
                    new = torch.as_tensor(st["step"], dtype=torch.float32).to(p.device).reshape(())
                    if prev is not None and "step" in prev:
                        if step_t is None or prev["step"] is not step_t:
                            prev["step"].copy_(new)
                        st["step"] = step_t = prev["step"]  # the group's one device counter keeps its address
                    else:
                        st["step"] = new.clone()  # private fp32 device scalar (torch may hand back the caller's tensor uncopied)
                        replaced = True
            if not replaced and gi < len(old_launch) and old_launch[gi] is not None:
                group["_ur_launches"] = old_launch[gi]  # same addresses: the descriptor arrays are still right
        if replaced:
            self.generation = getattr(self, "generation", 0) + 1
