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


class UniPCMultistepScheduler:
    """UniPC (Zhao et al. 2023) multistep predictor-corrector, restated from the published algorithm as diffusers 0.24's
    ``UniPCMultistepScheduler`` implements it (the scheduler eval/test_real.py:485-492 attaches eight times and runs for
    20 steps): B(h) variant ``bh2``, data prediction (``predict_x0=True``), ``solver_order`` 2 with first-order warm-up
    and ``lower_order_final``, corrector after every step but the first, ``linspace`` timestep spacing, sigmas
    interpolated from the training schedule.  diffusers itself is not available here, so this restatement
    is pinned by properties only (tests/test_schedulers_cpu.py): exact recovery of x0 under a perfect x0 predictor,
    first step == DDIM, order-2 convergence on a linear ODE, coefficient tables == step-by-step arithmetic.

    ``from_config`` takes the SD-1.x scheduler config (scaled_linear betas 0.00085 .. 0.012, ``prediction_type``
    "sample" for the x0 model).  Every update is a LINEAR combination of tensors with data-independent
    scalar weights; ``coefficient_table()`` exposes them so the pipeline can run the update as one HIP kernel
    (``ur_unipc_update``) inside the step graph."""

    init_noise_sigma = 1.0
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", solver_order: int = 2, prediction_type: str = "sample",
                 predict_x0: bool = True, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector=(), timestep_spacing: str = "linspace", steps_offset: int = 0, **_ignored):
        if solver_order not in (1, 2) or not predict_x0 or solver_type != "bh2":
            raise NotImplementedError("UniPC: solver_order 1 / 2, predict_x0, bh2 (the diffusers defaults) are implemented")
        if prediction_type not in ("sample", "epsilon"):
            raise ValueError(prediction_type)
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, solver_order=solver_order, prediction_type=prediction_type,
                           predict_x0=predict_x0, solver_type=solver_type, lower_order_final=lower_order_final,
                           disable_corrector=tuple(disable_corrector), timestep_spacing=timestep_spacing,
                           steps_offset=steps_offset)
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.solver_order = solver_order
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None
        self.sigmas = None
        self._reset()

    @classmethod
    def from_config(cls, config, **overrides):
        c = dict(config) if not isinstance(config, cls) else dict(config.config)
        c.update(overrides)
        return cls(**c)

    def _reset(self):
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1
        self._step_index = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np

        n_train = self.num_train_timesteps
        sp = self.config["timestep_spacing"]
        if sp == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif sp == "leading":
            ratio = n_train // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64)
            ts += self.config["steps_offset"]
        elif sp == "trailing":
            ts = (np.arange(n_train, 0, -n_train / num_inference_steps).round() - 1).astype(np.int64)
        else:
            raise ValueError(sp)
        ac = self.alphas_cumprod.numpy().astype(np.float64)
        sig = ((1 - ac) / ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        sigma_last = ((1 - ac[0]) / ac[0]) ** 0.5
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [sigma_last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(ts)
        self._reset()

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / ((sigma ** 2 + 1.0) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _lambda(self, i):
        a, s = self._alpha_sigma(self.sigmas[i])
        return torch.log(a) - torch.log(s)

    # ---- the data-independent scalars of one update ------------------------------------------------------------
    def _bh_terms(self, i_from: int, i_to: int, order: int, corrector: bool):
        """Scalars of the update from grid point ``i_from`` to ``i_to`` (indices into ``sigmas``):
        returns (sigma_t / sigma_s0, alpha_t * h_phi_1, alpha_t * B_h, rhos) with rhos = the weights of
        [D1(m_{-1}) .. , D1_t]: predictor ``order - 1`` of them, corrector ``order``; and the r_k of the older points."""
        a_t, s_t = self._alpha_sigma(self.sigmas[i_to])
        a_0, s_0 = self._alpha_sigma(self.sigmas[i_from])
        lam_t, lam_0 = torch.log(a_t) - torch.log(s_t), torch.log(a_0) - torch.log(s_0)
        h = lam_t - lam_0
        rks = []
        for k in range(1, order):
            rks.append((self._lambda(i_from - k) - lam_0) / h)
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r).to(h.dtype) for r in rks])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for k in range(1, order + 1):
            R.append(torch.pow(rks, k - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= k + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.stack(b)
        if corrector:
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        else:
            rhos = torch.tensor([0.5]) if order == 2 else (torch.linalg.solve(R[:-1, :-1], b[:-1]) if order > 2 else torch.zeros(0))
        return s_t / s_0, a_t * h_phi_1, a_t * B_h, rhos, rks

    def _orders(self):
        """this_order of the predictor at every step index (warm-up + lower_order_final)."""
        n, out, low = self.num_inference_steps, [], 0
        for i in range(n):
            o = min(self.solver_order, n - i) if self.config["lower_order_final"] else self.solver_order
            out.append(min(o, low + 1))
            if low < self.solver_order:
                low += 1
        return out

    def coefficient_table(self) -> torch.Tensor:
        """[n, 8] fp32.  With m_i the (converted) model output of step i, L the corrected sample ("last_sample") and
        x_next the next model input:
            L_i      = c0 * L_{i-1} + c1 * m_{i-1} + c2 * m_{i-2} + c3 * m_i     (step 0: L_0 = the initial sample)
            x_{i+1}  = p0 * L_i     + p1 * m_i     + p2 * m_{i-1}
        row i = (c0, c1, c2, c3, p0, p1, p2, 0)."""
        n, orders = self.num_inference_steps, self._orders()
        rows = []
        for i in range(n):
            c = [1.0, 0.0, 0.0, 0.0]
            if i > 0 and (i - 1) not in self.config["disable_corrector"]:
                o = orders[i - 1]
                ratio, ah1, aB, rhos, rks = self._bh_terms(i - 1, i, o, corrector=True)
                # x = ratio*L - ah1*m0 - aB*(sum_k rho_k (m_k - m0)/r_k + rho_last (m_t - m0))
                c0, c1, c2, c3 = float(ratio), -float(ah1), 0.0, -float(aB * rhos[-1])
                c1 += float(aB * rhos[-1])
                if o == 2:
                    w = float(aB * rhos[0] / rks[0])
                    c2 -= w
                    c1 += w
                c = [c0, c1, c2, c3]
            o = orders[i]
            ratio, ah1, aB, rhos, rks = self._bh_terms(i, i + 1, o, corrector=False)
            p0, p1, p2 = float(ratio), -float(ah1), 0.0
            if o == 2:
                w = float(aB * rhos[0] / rks[0])
                p2 -= w
                p1 += w
            rows.append(c + [p0, p1, p2, 0.0])
        return torch.tensor(rows, dtype=torch.float32)

    # ---- diffusers protocol: one step on tensors ------------------------------------------------------------------
    def _index_for(self, timestep):
        t = int(torch.as_tensor(timestep).flatten()[0].item())
        idx = (self.timesteps.cpu() == t).nonzero()
        return int(idx[1 if len(idx) > 1 else 0]) if len(idx) else len(self.timesteps) - 1

    def convert_model_output(self, model_output, sample):
        if self.prediction_type == "sample":
            return model_output
        a, s = self._alpha_sigma(self.sigmas[self._step_index])
        return (sample - s * model_output) / a

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False, **_):
        """One predictor-corrector step.  Arithmetic in fp32 whatever the input dtypes, the two samples (corrected
        ``last_sample``, returned ``prev_sample``) rounded through ``sample.dtype`` -- the precision contract of the
        fused kernel (``ur_unipc_update`` with ``round_master``).  diffusers evaluates the same expressions in the
        tensors' own dtype, i.e. with an fp16 rounding after every op when the model runs in fp16."""
        if self._step_index is None:
            self._step_index = self._index_for(timestep)
        i = self._step_index
        out_dtype = sample.dtype
        cd = torch.float64 if sample.dtype == torch.float64 else torch.float32
        m_t = self.convert_model_output(model_output.to(cd), sample.to(cd))
        sample = sample.to(cd)
        use_corrector = i > 0 and (i - 1) not in self.config["disable_corrector"] and self.last_sample is not None
        if use_corrector:
            o = self.this_order
            ratio, ah1, aB, rhos, rks = self._bh_terms(i - 1, i, o, corrector=True)
            m0 = self.model_outputs[-1]
            x = ratio * self.last_sample - ah1 * m0
            res = rhos[-1] * (m_t - m0)
            if o == 2:
                res = res + rhos[0] * ((self.model_outputs[-2] - m0) / rks[0])
            sample = (x - aB * res).to(out_dtype).to(cd)
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = m_t
        o = min(self.solver_order, len(self.timesteps) - i) if self.config["lower_order_final"] else self.solver_order
        self.this_order = min(o, self.lower_order_nums + 1)
        self.last_sample = sample
        ratio, ah1, aB, rhos, rks = self._bh_terms(i, i + 1, self.this_order, corrector=False)
        x = ratio * sample - ah1 * m_t
        if self.this_order == 2:
            x = x - aB * (rhos[0] * ((self.model_outputs[-2] - m_t) / rks[0]))
        prev = x.to(out_dtype)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else {"prev_sample": prev}

    def add_noise(self, original, noise, timesteps):
        a = self.alphas_cumprod.to(original.device)[timesteps.long()].to(original.dtype)
        while a.dim() < original.dim():
            a = a[..., None]
        return a.sqrt() * original + (1 - a).sqrt() * noise
