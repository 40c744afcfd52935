"""CogVideoXDDIMScheduler as the reference loop uses it (pipeline_cogvideox_image2video_lowpass.py):
``set_timesteps`` via retrieve_timesteps (cog:95-151, 958), ``init_noise_sigma`` (cog:424),
``scale_model_input`` (cog:1065), ``order`` (cog:1001) and ``step`` (cog:1112).

The schedule tables are host float64 (as in diffusers); the per-element update runs in the fused HIP kernel
``alg_cfg_ddim_step`` (CFG combine + step + cast, alg_amd/csrc/cfg_step.hip).  The arithmetic follows the
published CogVideoXDDIMScheduler (diffusers @ be2fb77, not in the reference tree -- parity unpinned).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import _lib


class CogVideoXDDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=True, steps_offset=0, prediction_type="v_prediction",
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="trailing",
                 rescale_betas_zero_snr=True, snr_shift_scale=1.0, **unused):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError("only the scaled_linear beta schedule of the CogVideoX checkpoints is built")
        if prediction_type != "v_prediction":
            raise NotImplementedError("only v_prediction (CogVideoX) is built")
        if clip_sample:
            raise NotImplementedError("clip_sample is off in every CogVideoX scheduler config")
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
            steps_offset=steps_offset, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
            rescale_betas_zero_snr=rescale_betas_zero_snr, snr_shift_scale=snr_shift_scale)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        acp = acp / (snr_shift_scale + (1 - snr_shift_scale) * acp)  # SNR shift
        if rescale_betas_zero_snr:  # zero terminal SNR
            root = acp.sqrt()
            first, last = root[0].clone(), root[-1].clone()
            root = (root - last) * (first / (first - last))
            acp = root ** 2
        self.alphas_cumprod = acp
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else acp[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **overrides):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(overrides)
        return cls(**d)

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError("num_inference_steps %d exceeds num_train_timesteps %d" % (num_inference_steps, n))
        self.num_inference_steps = num_inference_steps
        spacing = self.config.timestep_spacing
        if spacing == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        elif spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy()
            ts = ts.astype(np.int64) + self.config.steps_offset
        elif spacing == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError("unsupported timestep_spacing %r" % spacing)
        self.timesteps = torch.from_numpy(ts)  # host int64: the loop never syncs on them

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timestep):
        """(sqrt(alpha_t), sqrt(1 - alpha_t), a_t, b_t) as Python floats (float64) for the eta = 0 update
        x0 = sqrt(alpha_t) x - sqrt(1-alpha_t) v;  x_prev = a_t x + b_t x0."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        coef_a = ((1 - a_prev) / (1 - a_t)) ** 0.5
        coef_b = a_prev ** 0.5 - a_t ** 0.5 * coef_a
        return float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(coef_a), float(coef_b)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        """Generic scheduler API (cog:1112): returns the previous sample as a NEW tensor of ``sample``'s dtype
        promoted with fp32 the way the reference's expression does (the loop casts it back, cog:1123)."""
        if eta != 0.0:
            raise NotImplementedError("eta > 0 (stochastic DDIM) is not used by the ALG configs")
        sa, sb, ca, cb = self.step_coefficients(timestep)
        out = sample.clone()
        _lib.cfg_ddim_step_(model_output.contiguous(), out, 1, 1.0, sa, sb, ca, cb)
        if not return_dict:
            return (out,)
        return SimpleNamespace(prev_sample=out)

    def fused_cfg_step_(self, noise_pred, latents, n_pass, guidance_scale, timestep):
        """cog:1091-1123 in one kernel, in place on ``latents``."""
        sa, sb, ca, cb = self.step_coefficients(timestep)
        return _lib.cfg_ddim_step_(noise_pred, latents, n_pass, guidance_scale, sa, sb, ca, cb)
