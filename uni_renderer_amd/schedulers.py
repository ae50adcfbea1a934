"""Minimal sampler used when the caller does not attach diffusers scheduler objects.

The reference attaches eight ``UniPCMultistepScheduler`` instances to the pipeline (eval/test_real.py:485-492)
and BASELINE.json's config 3 names a 50-step DDIM; both come from diffusers, which is not available here.  The
pipeline only needs the diffusers scheduler *protocol* -- ``set_timesteps``, ``timesteps``, ``order``,
``init_noise_sigma``, ``scale_model_input``, ``step(...)[0]``, ``add_noise`` -- so any diffusers scheduler can
still be attached; this class is a DDIM (eta = 0) with the model's x0 ("sample") prediction on the
SD scaled-linear beta schedule.  It is elementwise plumbing between denoise steps (SURVEY.md §8f rank 1 moves it
into a fused HIP kernel), not part of the measured hot path.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "sample", steps_offset: int = 1, set_alpha_to_one: bool = False):
        if prediction_type not in ("sample", "epsilon"):
            raise ValueError(prediction_type)
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).float()
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(0, num_inference_steps) * ratio).flip(0) + self.steps_offset
        self.timesteps = ts.clamp_max(self.num_train_timesteps - 1).to(device)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def _alpha(self, t, device):
        t = torch.as_tensor(t, device="cpu").long().clamp(0, self.num_train_timesteps - 1)
        return self.alphas_cumprod[t].to(device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False, **_):
        t = int(torch.as_tensor(timestep).flatten()[0].item())
        prev_t = t - self.num_train_timesteps // (self.num_inference_steps or self.num_train_timesteps)
        a_t = self._alpha(t, sample.device)
        a_prev = self._alpha(prev_t, sample.device) if prev_t >= 0 else self.final_alpha_cumprod.to(sample.device)
        out = model_output.to(torch.float32)
        x = sample.to(torch.float32)
        if self.prediction_type == "sample":
            x0 = out
            eps = (x - a_t.sqrt() * x0) / (1 - a_t).sqrt()
        else:
            eps = out
            x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        prev = a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps
        prev = prev.to(sample.dtype)
        return (prev,) if not return_dict else {"prev_sample": prev}

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        a = self.alphas_cumprod.to(original.device)[timesteps.long()].to(original.dtype)
        while a.dim() < original.dim():
            a = a[..., None]
        return a.sqrt() * original + (1 - a).sqrt() * noise


def retrieve_timesteps(scheduler, num_inference_steps: Optional[int] = None, device=None, timesteps=None, **kw
                       ) -> Tuple[torch.Tensor, int]:
    """models/pipeline.py:80-121."""
    if timesteps is not None:
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kw)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kw)
    return scheduler.timesteps, num_inference_steps
