"""TEST INFRASTRUCTURE -- independent CPU restatement of the two samplers that sit between denoise steps.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

The reference attaches diffusers schedulers to its pipeline (eval/test_real.py:485-492: eight
``UniPCMultistepScheduler.from_config(...)``; BASELINE.json's config 3 names a 50-step DDIM) and calls their
``step(pred, t, latents)[0]`` between denoise steps (models/pipeline.py:2691-2730, 1645-1649).  diffusers 0.24 is an
un-vendored dependency that is absent here, so this file restates the PUBLISHED algorithms the way
diffusers 0.24 implements them -- and deliberately NOT the way ``uni_renderer_amd/schedulers.py`` does: numpy float64,
the ``D1s`` / ``einsum`` form of UniPC's B(h) updates (Zhao et al. 2023, Alg. 5-8; diffusers
``multistep_uni_p_bh_update`` / ``multistep_uni_c_bh_update``), no coefficient table, no helper shared with the product.
The product's host scheduler, its coefficient table and the fused HIP kernels (``ur_ddim_update``,
``ur_unipc_update``) are all checked against THIS file (tests/test_schedulers_cpu.py, tests/test_pipeline_gpu.py).

PARITY UNPINNED against diffusers itself (it cannot be imported); pinned by the closed-form probability-flow solution
for Gaussian data (order-of-convergence test) and by DDIM == first-order UniPC step identities.
"""
from __future__ import annotations

import numpy as np


def _alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear"):
    if schedule == "scaled_linear":  # diffusers builds the betas in float32
        betas = np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, n, dtype=np.float32) ** 2
    elif schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float32)
    else:
        raise ValueError(schedule)
    return np.cumprod(1.0 - betas.astype(np.float32), dtype=np.float32).astype(np.float64)


class DDIMOracle:
    """DDIM, eta = 0 (Song et al. 2021, eq. 12) as diffusers' DDIMScheduler.step: leading timestep spacing with
    ``steps_offset`` 1, ``set_alpha_to_one`` False, model output = x0 ("sample") or epsilon."""

    def __init__(self, num_train_timesteps=1000, prediction_type="sample", steps_offset=1, set_alpha_to_one=False):
        self.n = num_train_timesteps
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        self.ac = _alphas_cumprod(num_train_timesteps)
        self.final_ac = 1.0 if set_alpha_to_one else self.ac[0]

    def set_timesteps(self, num_inference_steps):
        self.steps = num_inference_steps
        ratio = self.n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = np.minimum(ts, self.n - 1)

    def step(self, model_output, t, sample):
        x = np.asarray(sample, dtype=np.float64)
        out = np.asarray(model_output, dtype=np.float64)
        prev_t = int(t) - self.n // self.steps
        a_t = self.ac[int(t)]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final_ac
        b_t = 1.0 - a_t
        if self.prediction_type == "sample":
            x0 = out
            eps = (x - a_t ** 0.5 * x0) / b_t ** 0.5
        else:
            eps = out
            x0 = (x - b_t ** 0.5 * eps) / a_t ** 0.5
        direction = (1.0 - a_prev) ** 0.5 * eps
        return a_prev ** 0.5 * x0 + direction


class UniPCOracle:
    """UniPC multistep (bh2, data prediction, ``solver_order`` p <= 3, ``lower_order_final``, corrector after every
    step but the first), ``linspace`` timestep spacing, sigmas interpolated from the training schedule with the
    ``sigma_last`` = sigma(t = 0) terminal value -- diffusers 0.24 ``UniPCMultistepScheduler``."""

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="sample", lower_order_final=True,
                 disable_corrector=()):
        self.n = num_train_timesteps
        self.solver_order = solver_order
        self.prediction_type = prediction_type
        self.lower_order_final = lower_order_final
        self.disable_corrector = tuple(disable_corrector)
        self.ac = _alphas_cumprod(num_train_timesteps)

    def set_timesteps(self, num_inference_steps):
        ts = np.linspace(0, self.n - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = ((1 - self.ac) / self.ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sigmas, [sig[0]]]).astype(np.float32).astype(np.float64)  # stored as float32 there
        self.timesteps = ts
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1
        self.step_index = 0

    @staticmethod
    def _sigma_to_alpha_sigma_t(sigma):
        alpha_t = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha_t, sigma * alpha_t

    def _lambda(self, idx):
        a, s = self._sigma_to_alpha_sigma_t(self.sigmas[idx])
        return np.log(a) - np.log(s)

    def convert_model_output(self, model_output, sample):
        if self.prediction_type == "sample":
            return model_output
        a, s = self._sigma_to_alpha_sigma_t(self.sigmas[self.step_index])
        return (sample - s * model_output) / a

    def _R_b(self, rks, hh, order):
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        B_h = np.expm1(hh)  # bh2
        R, b, factorial_i = [], [], 1
        for i in range(1, order + 1):
            R.append(np.power(rks, i - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / factorial_i
        return np.stack(R), np.array(b), h_phi_1, B_h

    def multistep_uni_p_bh_update(self, sample, order):
        m0 = self.model_outputs[-1]
        x = sample
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(self.sigmas[self.step_index + 1])
        _, sigma_s0 = self._sigma_to_alpha_sigma_t(self.sigmas[self.step_index])
        lambda_s0 = self._lambda(self.step_index)
        h = self._lambda(self.step_index + 1) - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lambda(self.step_index - i) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = np.array(rks)
        hh = -h  # predict_x0
        R, b, h_phi_1, B_h = self._R_b(rks, hh, order)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        if D1s:
            rhos_p = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            pred_res = np.einsum("k,k...->...", rhos_p, np.stack(D1s, 0))
        else:
            pred_res = 0.0
        return x_t_ - alpha_t * B_h * pred_res

    def multistep_uni_c_bh_update(self, this_model_output, last_sample, order):
        m0 = self.model_outputs[-1]
        x = last_sample
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(self.sigmas[self.step_index])
        _, sigma_s0 = self._sigma_to_alpha_sigma_t(self.sigmas[self.step_index - 1])
        lambda_s0 = self._lambda(self.step_index - 1)
        h = self._lambda(self.step_index) - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lambda(self.step_index - (i + 1)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = np.array(rks)
        hh = -h
        R, b, h_phi_1, B_h = self._R_b(rks, hh, order)
        rhos_c = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = np.einsum("k,k...->...", rhos_c[:-1], np.stack(D1s, 0)) if D1s else 0.0
        D1_t = this_model_output - m0
        return x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)

    def step(self, model_output, t, sample):
        sample = np.asarray(sample, dtype=np.float64)
        model_output = np.asarray(model_output, dtype=np.float64)
        use_corrector = (self.step_index > 0 and (self.step_index - 1) not in self.disable_corrector
                         and self.last_sample is not None)
        m = self.convert_model_output(model_output, sample)
        if use_corrector:
            sample = self.multistep_uni_c_bh_update(m, self.last_sample, self.this_order)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m
        this_order = (min(self.solver_order, len(self.timesteps) - self.step_index) if self.lower_order_final
                      else self.solver_order)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self.multistep_uni_p_bh_update(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev
