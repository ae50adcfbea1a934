"""Minimal stand-ins for the diffusers ``ConfigMixin`` / ``ModelMixin`` surface the reference's callers use
(SURVEY.md §8b "Object surface"): ``.config[...]`` get/set, ``register_to_config``, ``_internal_dict``,
``from_pretrained`` / ``save_pretrained`` in the diffusers on-disk layout (``config.json`` +
``diffusion_pytorch_model.safetensors``, optional ``subfolder``), ``.dtype`` / ``.device``.

The reference gets these from ``diffusers.ModelMixin/ConfigMixin`` (models/controlnet.py:19-20,49,1170,1781);
callers: train/train.py:961-996 (from_pretrained, config surgery), 1002-1045 (save/load hooks), 1082 (.dtype).
"""
from __future__ import annotations

import functools
import inspect
import json
import os
from collections import OrderedDict
from typing import Any, Dict

import torch
import torch.nn as nn

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"


class FrozenDict(OrderedDict):
    """Attribute + item access like diffusers' FrozenDict; item assignment stays allowed because
    train/train.py:985,996 patch ``config['in_channels']`` through ``_internal_dict``-style surgery."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def register_to_config(init):
    """Decorator for ``__init__``: records every (defaulted) keyword argument into ``self.config``."""

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.items())[1:]
        cfg = {name: p.default for name, p in params if p.default is not inspect.Parameter.empty}
        for (name, _), val in zip(params, args):
            cfg[name] = val
        cfg.update({k: v for k, v in kwargs.items() if k in dict(params)})
        init(self, *args, **kwargs)
        base = dict(getattr(self, "_internal_dict", {}))
        base.update(cfg)
        base["_class_name"] = self.__class__.__name__
        object.__setattr__(self, "_internal_dict", FrozenDict(base))

    return inner


def _jsonable(v):
    if isinstance(v, tuple):
        return [_jsonable(x) for x in v]
    if isinstance(v, list):
        return [_jsonable(x) for x in v]
    return v


class ConfigModelMixin:
    """Mixed into the three network classes (must come before nn.Module in the MRO)."""

    config_name = CONFIG_NAME

    @property
    def config(self) -> FrozenDict:
        return self._internal_dict

    def register_to_config(self, **kwargs):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kwargs)
        object.__setattr__(self, "_internal_dict", FrozenDict(d))

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            return p.dtype
        return torch.float32

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    # -- diffusers-layout checkpoint I/O ---------------------------------------------------------
    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **_):
        from safetensors.torch import save_file

        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: _jsonable(v) for k, v in self.config.items()}
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().contiguous().cpu() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def load_config(cls, path: str, subfolder: str = None, **_) -> Dict[str, Any]:
        p = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(p, CONFIG_NAME)) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config: Dict[str, Any], **overrides):
        sig = inspect.signature(cls.__init__).parameters
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in dict(config).items() if k in sig}
        kw.update(overrides)
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: str = None, torch_dtype=None,
                        revision=None, variant=None, **kwargs):
        from safetensors.torch import load_file

        p = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        model = cls.from_config(cls.load_config(p))
        name = WEIGHTS_NAME if variant is None else WEIGHTS_NAME.replace(".safetensors", f".{variant}.safetensors")
        sd = cls._convert_state_dict(load_file(os.path.join(p, name)))
        # channel surgery (train.py:976,988-989) changes conv_in/conv_out shapes: honour the file
        own = model.state_dict()
        for k, v in sd.items():
            if k in own and own[k].shape != v.shape:
                mod_name, _, pname = k.rpartition(".")
                setattr(model.get_submodule(mod_name), pname, nn.Parameter(torch.empty_like(v)))
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    @classmethod
    def _convert_state_dict(cls, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Hook for on-disk key layouts older than the module tree (AutoencoderKL overrides it)."""
        return sd

    # -- no-ops kept for caller compatibility (train.py:1066-1074) --------------------------------
    def enable_xformers_memory_efficient_attention(self, *_, **__):
        return None

    def enable_gradient_checkpointing(self):
        """Activation recompute as the reference does it (caller train.py:1073-1074; ``_set_gradient_checkpointing``,
        controlnet.py:745-747 / 1653-1655 / 2338-2340: every sub-block that HAS a ``gradient_checkpointing`` attribute gets it
        set; unet_2d_blocks.py:1172-1197: in training mode a checkpointed block runs each of its ResnetBlock2D under
        ``torch.utils.checkpoint.checkpoint(..., use_reentrant=False)``, its attention is not checkpointed).  The training
        forward (train_step._down_mid / _up_out) honours the flag: the resnet's activations (two GroupNorm outputs, the conv1
        output) are dropped after the forward and recomputed by the same HIP kernels in the backward -- bit-identical
        gradients (tests/test_train_gpu.py).  Not needed for memory here (cfg 4's per-GPU step peaks at ~38 GB of 288 GB)."""
        self._set_gc(True)

    def disable_gradient_checkpointing(self):
        self._set_gc(False)

    def _set_gc(self, value: bool):
        self.gradient_checkpointing = bool(value)
        for m in self.modules():
            if m is not self and hasattr(m, "gradient_checkpointing"):
                m.gradient_checkpointing = bool(value)

    @property
    def is_gradient_checkpointing(self) -> bool:
        return any(getattr(m, "gradient_checkpointing", False) for m in self.modules())

    def set_attention_slice(self, *_):
        return None
