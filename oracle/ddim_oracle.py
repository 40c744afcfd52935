"""CPU restatement of CogVideoXDDIMScheduler as the reference loop uses it
(pipeline_cogvideox_image2video_lowpass.py:958 set_timesteps via retrieve_timesteps cog:95-151,
cog:424 init_noise_sigma, cog:1065 scale_model_input, cog:1001 order, cog:1112 step).

TEST INFRASTRUCTURE -- never imported by ``alg_amd``.

PARITY UNPINNED: the arithmetic lives in diffusers @ be2fb77 (requirements.txt:13), absent here.
Restated from the published algorithm: scaled-linear betas, SNR shift, zero-terminal-SNR rescale,
'trailing' timestep spacing, v-prediction, eta = 0.
"""
from __future__ import annotations

import numpy as np
import torch


class DDIMOracle:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=1.0,
                 rescale_betas_zero_snr=True, set_alpha_to_one=True, timestep_spacing="trailing"):
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            s = ac.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * (s0 / (s0 - sT))
            ac = s ** 2
        self.alphas_cumprod = ac
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else ac[0]
        self.timestep_spacing = timestep_spacing
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            ratio = n / num_inference_steps
            ts = np.round(np.arange(n, 0, -ratio)).astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        else:  # linspace
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self, timestep):
        """float64 scalars (sqrt_alpha_t, sqrt_beta_t, a_t, b_t) of the eta=0 v-prediction update."""
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        ca = ((1 - a_prev) / (1 - a_t)) ** 0.5
        cb = a_prev ** 0.5 - a_t ** 0.5 * ca
        return a_t ** 0.5, beta_t ** 0.5, ca, cb

    def step(self, model_output, timestep, sample):
        """x0 = sqrt(a_t) x - sqrt(1-a_t) v ;  x_prev = ca x + cb x0.  The scalars are 0-dim float64
        tensors, so torch's type promotion reproduces the reference's rounding points (a bf16
        ``sample`` times a 0-dim fp64 tensor stays bf16; adding the fp32 ``model_output`` gives fp32)."""
        sa, sb, ca, cb = self.coefficients(timestep)
        x0 = sa * sample - sb * model_output
        return ca * sample + cb * x0


class DPMOracle(DDIMOracle):
    """CogVideoXDPMScheduler (cog:1114-1122: ``step(model_output, old_pred_original_sample, timestep, timestep_back,
    sample, generator=)`` -> (prev_sample, pred_original_sample)): second-order SDE-DPM-Solver++ in the log-SNR variable,
    fresh Gaussian noise in the sample's dtype every step, first-order on the first and last step.  Same tables as the
    DDIM scheduler.  PARITY UNPINNED (diffusers @ be2fb77, absent)."""

    def _variables(self, a_t, a_prev, a_back):
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
        h = lamb_next - lamb
        r = None
        if a_back is not None:
            lamb_prev = ((a_back / (1 - a_back)) ** 0.5).log()
            r = (lamb - lamb_prev) / h
        return h, r

    def multipliers(self, timestep, timestep_back):
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        a_back = self.alphas_cumprod[int(timestep_back)] if timestep_back is not None else None
        h, r = self._variables(a_t, a_prev, a_back)
        m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_prev ** 0.5
        m_noise = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        m3 = m4 = None
        if r is not None:
            m3, m4 = 1 + 1 / (2 * r), 1 / (2 * r)
        return a_t ** 0.5, (1 - a_t) ** 0.5, m1, m2, m3, m4, m_noise, prev

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, generator=None):
        sa, sb, m1, m2, m3, m4, mn, prev = self.multipliers(timestep, timestep_back)
        x0 = sa * sample - sb * model_output
        noise = torch.randn(sample.shape, generator=generator, dtype=sample.dtype)
        prev_sample = m1 * sample - m2 * x0 + mn * noise
        if old_pred_original_sample is None or prev < 0:
            return prev_sample, x0
        denoised_d = m3 * x0 - m4 * old_pred_original_sample
        noise = torch.randn(sample.shape, generator=generator, dtype=sample.dtype)
        return m1 * sample - m2 * denoised_d + mn * noise, x0
