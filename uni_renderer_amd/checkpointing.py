"""Training-state checkpoints in the reference's on-disk layout (train/train.py:1002-1045, 1191-1218,
1434-1457).

The reference saves through ``accelerator.save_state(output_dir/checkpoint-{global_step})`` with pre-hooks that write
each network with ``save_pretrained`` into the sub-folders ``controlnet/`` (AttributeEncoderModel), ``controldec/``
(AttributeDecoderModel) and ``unet/`` (UNet2DConditionModel) -- chosen by CLASS NAME (1010-1016) -- rotates old
checkpoints by ``checkpoints_total_limit`` BEFORE saving the new one (1436-1453), and resumes from ``"latest"`` = the
``checkpoint-N`` with the largest N (1196-1200), reloading each network with ``from_pretrained`` +
``register_to_config`` + ``load_state_dict`` (1022-1042) and then OVERWRITING lr / betas of every param group from the
command line (1209-1211).  This module is that behaviour without accelerate: plain functions over the three modules
and a torch optimizer (``optimizer.bin`` = ``torch.save(optimizer.state_dict())``, accelerate's file name).
"""
from __future__ import annotations

import os
import shutil
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

SUBFOLDERS = {"AttributeEncoderModel": "controlnet", "AttributeDecoderModel": "controldec", "UNet2DConditionModel": "unet"}
OPTIMIZER_NAME = "optimizer.bin"
PREFIX = "checkpoint"


def _unwrap(m):
    return getattr(m, "module", m)  # a DistributedDataParallel-style wrapper (train.py:1140-1142)


def list_checkpoints(output_dir: str) -> List[str]:
    """``checkpoint-N`` directory names sorted by N (train.py:1437-1439)."""
    if not os.path.isdir(output_dir):
        return []
    dirs = [d for d in os.listdir(output_dir) if d.startswith(PREFIX)]
    return sorted(dirs, key=lambda x: int(x.split("-")[1]))


def rotate_checkpoints(output_dir: str, checkpoints_total_limit: Optional[int]) -> List[str]:
    """Before saving a new checkpoint keep at most ``limit - 1`` old ones (train.py:1436-1453).  Returns what was
    removed."""
    if checkpoints_total_limit is None:
        return []
    ckpts = list_checkpoints(output_dir)
    removed = []
    if len(ckpts) >= checkpoints_total_limit:
        for d in ckpts[: len(ckpts) - checkpoints_total_limit + 1]:
            shutil.rmtree(os.path.join(output_dir, d))
            removed.append(d)
    return removed


def save_state(models: Sequence[torch.nn.Module], output_dir: str, global_step: int, optimizer=None,
               checkpoints_total_limit: Optional[int] = None, is_main_process: bool = True) -> Optional[str]:
    """``output_dir/checkpoint-{global_step}/{controlnet,controldec,unet}/`` (+ ``optimizer.bin``), after rotation.
    Only the main process writes (train.py:1433); every rank may call this."""
    if not is_main_process:
        return None
    os.makedirs(output_dir, exist_ok=True)
    rotate_checkpoints(output_dir, checkpoints_total_limit)
    path = os.path.join(output_dir, f"{PREFIX}-{int(global_step)}")
    for m in models:
        m = _unwrap(m)
        sub = SUBFOLDERS.get(m.__class__.__name__)
        if sub is None:
            raise ValueError(f"unknown network class {m.__class__.__name__} (the reference's hooks match by class name)")
        m.save_pretrained(os.path.join(path, sub))
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(path, OPTIMIZER_NAME))
    return path


def load_state(models: Sequence[torch.nn.Module], input_dir: str, optimizer=None) -> None:
    """The load hook of train.py:1022-1042: every network is re-read in diffusers layout from its sub-folder, its config
    is patched over the live module's (``register_to_config``) and the weights are loaded IN PLACE (the live modules
    keep their identity, device and dtype: optimizers and DDP wrappers built over them stay valid)."""
    for m in models:
        m = _unwrap(m)
        sub = SUBFOLDERS.get(m.__class__.__name__)
        if sub is None:
            raise ValueError(f"unknown network class {m.__class__.__name__}")
        loaded = type(m).from_pretrained(input_dir, subfolder=sub)
        m.register_to_config(**loaded.config)
        m.load_state_dict(loaded.state_dict())
        del loaded
    opt_path = os.path.join(input_dir, OPTIMIZER_NAME)
    if optimizer is not None and os.path.exists(opt_path):
        optimizer.load_state_dict(torch.load(opt_path, map_location="cpu"))


def resume_from_checkpoint(models: Sequence[torch.nn.Module], output_dir: str, resume: Optional[str], optimizer=None,
                           learning_rate: Optional[float] = None, betas: Optional[Tuple[float, float]] = None) -> int:
    """train.py:1191-1218.  ``resume``: ``None`` (fresh run), ``"latest"`` (largest ``checkpoint-N`` in ``output_dir``)
    or a checkpoint path / name (its basename is looked up in ``output_dir``, 1193).  Returns the global step to continue
    from (0 when nothing was loaded).  After loading, lr / betas of every param group are overwritten with the given
    values, as the reference does from its CLI flags (1209-1211)."""
    if not resume:
        return 0
    if resume != "latest":
        name = os.path.basename(os.path.normpath(resume))
    else:
        ckpts = list_checkpoints(output_dir)
        name = ckpts[-1] if ckpts else None
    if name is None or not os.path.isdir(os.path.join(output_dir, name)):
        return 0  # "Checkpoint ... does not exist. Starting a new training run." (1202-1206)
    load_state(models, os.path.join(output_dir, name), optimizer)
    if optimizer is not None:
        for g in optimizer.param_groups:
            if learning_rate is not None:
                g["lr"] = learning_rate
            if betas is not None:
                g["betas"] = tuple(betas)
    return int(name.split("-")[1])
