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


class _ConfigLoading:
    """`from_pretrained` / `from_config` of the diffusers schedulers: construct from `scheduler/scheduler_config.json` (or a
    dict), keeping only the keys the constructor knows."""

    @classmethod
    def from_pretrained(cls, path, subfolder="scheduler", **_):
        from .weights import scheduler_from_pretrained
        return scheduler_from_pretrained(cls, path, subfolder)

    @classmethod
    def from_config(cls, config, **overrides):
        import inspect
        known = set(inspect.signature(cls.__init__).parameters) - {"self"}
        d = dict(config) if isinstance(config, dict) else dict(vars(config))
        kw = {k: v for k, v in d.items() if k in known}
        kw.update({k: v for k, v in overrides.items() if k in known})
        return cls(**kw)


class CogVideoXDDIMScheduler(_ConfigLoading):
    """Constructor defaults are the values `THUDM/CogVideoX-5b-I2V/scheduler/scheduler_config.json` ships (snr_shift_scale
    1.0, trailing spacing, v_prediction, zero terminal SNR) so that `CogVideoXDDIMScheduler()` IS the C2 scheduler; the
    published class defaults (`_published_defaults`) fill the keys a checkpoint's config file leaves out."""
    order = 1
    init_noise_sigma = 1.0
    _published_defaults = dict(snr_shift_scale=3.0, timestep_spacing="leading", prediction_type="epsilon",
                               rescale_betas_zero_snr=False, set_alpha_to_one=True, steps_offset=0)

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
        d = dict(getattr(cls, "_published_defaults", {}))     # keys a partial config leaves out: the published class defaults
        d.update(dict(vars(config)) if not isinstance(config, dict) else dict(config))
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
        # `eta` is part of the published signature and is never read by the published update (no variance term): ignored here too
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


def _step_index_for(timesteps, timestep):
    """index_for_timestep of the diffusers schedulers: second match on duplicates, last index on no match."""
    t = float(timestep)
    hits = [i for i, v in enumerate(timesteps.tolist()) if float(v) == t]
    if not hits:
        return len(timesteps) - 1
    return hits[1] if len(hits) > 1 else hits[0]


class FlowMatchEulerDiscreteScheduler(_ConfigLoading):
    """Flow-match Euler scheduler as the HunyuanVideo loop drives it (hy:1111-1112 custom ``sigmas`` through
    retrieve_timesteps, hy:1265-1269 ``step``; run.py:82-86 ``from_config(flow_shift=, invert_sigmas=)``).
    Static shift only (the HunyuanVideo-I2V config; dynamic shifting / karras / beta sigmas are not used by it).
    Host tables are fp32 like the published scheduler; the update x + (sigma_next - sigma) * v runs in
    ``alg_lincomb`` with torch-eager rounding and returns the model output's dtype."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, invert_sigmas=False,
                 shift_terminal=None, flow_shift=None, **unused):
        if use_dynamic_shifting or shift_terminal:
            raise NotImplementedError("dynamic shifting / shift_terminal are not part of the ALG configs")
        # run.py:82-86 passes `flow_shift`, which is not a parameter of this scheduler class: it is recorded and has
        # no effect on the sigmas (the checkpoint's own `shift` does) -- same as the reference run
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift,
                                      use_dynamic_shifting=False, invert_sigmas=invert_sigmas,
                                      shift_terminal=None, flow_shift=flow_shift)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self._step_index = None

    from_config = classmethod(CogVideoXDDIMScheduler.from_config.__func__)

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        n = self.config.num_train_timesteps
        if sigmas is None:
            ts = np.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps)
            sigmas = ts / n
        else:
            sigmas = np.array(sigmas).astype(np.float32)
            num_inference_steps = len(sigmas)
        shift = self.config.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        sig = torch.from_numpy(np.asarray(sigmas)).to(torch.float32)
        if self.config.invert_sigmas:
            sig = 1.0 - sig
            tail = torch.ones(1)
        else:
            tail = torch.zeros(1)
        self.num_inference_steps = num_inference_steps
        self.timesteps = sig * n                      # host fp32: the loop never syncs on them
        self.sigmas = torch.cat([sig, tail])
        self._step_index = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        if self._step_index is None:
            self._step_index = _step_index_for(self.timesteps, timestep)
        i = self._step_index
        dt = self.sigmas[i + 1] - self.sigmas[i]                   # fp32 subtraction as published
        # the published scheduler keeps its sigmas ON THE DEVICE: a 0-dim device tensor times a bf16 tensor is first
        # cast to bf16 by TensorIterator (unlike a python / CPU scalar, which enters at fp32) -- same value here
        dt = dt.to(model_output.dtype).float().item()
        x = sample if sample.dtype == torch.float32 else sample.float()
        out = _lib.lincomb([(1.0, x.contiguous()), (dt, model_output.contiguous())], model_output.dtype)
        self._step_index += 1
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)


class UniPCMultistepScheduler(_ConfigLoading):
    """UniPC (bh1/bh2, predict_x0, flow sigmas) as the Wan loop drives it (wan:815-816 ``set_timesteps``, wan:927
    ``step``; run.py:63 ``from_config(flow_shift=3.0 | 5.0)``).  solver_order <= 2 (the Wan checkpoints ship 2).
    Scalars follow the published scheduler's fp32 0-dim tensor arithmetic on the host; the per-element updates run
    in ``alg_lincomb`` (x0 = sample - sigma * v) and ``alg_unipc_update`` (corrector, predictor)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", predict_x0=True,
                 solver_type="bh2", lower_order_final=True, disable_corrector=(), use_flow_sigmas=True,
                 flow_shift=1.0, final_sigmas_type="zero", thresholding=False, **unused):
        if not (use_flow_sigmas and prediction_type == "flow_prediction" and predict_x0):
            raise NotImplementedError("only the flow-sigma / flow_prediction / predict_x0 UniPC of the Wan configs")
        if solver_order not in (1, 2) or solver_type not in ("bh1", "bh2") or thresholding:
            raise NotImplementedError("UniPC: solver_order <= 2, bh1/bh2, no thresholding")
        if final_sigmas_type not in ("zero", "sigma_min"):
            raise ValueError("final_sigmas_type must be 'zero' or 'sigma_min'")
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type,
            predict_x0=predict_x0, solver_type=solver_type, lower_order_final=lower_order_final,
            disable_corrector=list(disable_corrector), use_flow_sigmas=use_flow_sigmas, flow_shift=flow_shift,
            final_sigmas_type=final_sigmas_type, thresholding=False)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps,
                                                      dtype=np.float32)[::-1].copy())
        self.sigmas = None
        self._reset()

    from_config = classmethod(CogVideoXDDIMScheduler.from_config.__func__)

    def _reset(self):
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps, device=None):
        n, sh = self.config.num_train_timesteps, self.config.flow_shift
        alphas = np.linspace(1, 1 / n, num_inference_steps + 1)
        sig = 1.0 - alphas
        sig = np.flip(sh * sig / (1 + (sh - 1) * sig))[:-1].copy()
        ts = (sig * n).copy()
        last = sig[-1] if self.config.final_sigmas_type == "sigma_min" else 0.0
        self.sigmas = torch.from_numpy(np.concatenate([sig, [last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(torch.int64)   # host: the loop never syncs on them
        self.num_inference_steps = len(ts)
        self._reset()

    def scale_model_input(self, sample, timestep=None):
        return sample

    # ---- host scalars (fp32 0-dim tensors, the published order of operations) -------------------------------------
    def _lambda(self, sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _bh_scalars(self, sigma_t, sigma_s0, prev_sigma, order, corrector):
        """(r, c, k, rk, rhos) of  x_t = r x - c m0 - k (...)  for one predictor / corrector update."""
        alpha_t = 1 - sigma_t
        lam_t, lam_s0 = self._lambda(sigma_t), self._lambda(sigma_s0)
        h = lam_t - lam_s0
        rks, rk = [], None
        if order == 2:
            rk = (self._lambda(prev_sigma) - lam_s0) / h
            rks.append(rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.tensor(b)
        if corrector:
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        else:
            rhos = torch.tensor([0.5]) if order == 2 else None
        return sigma_t / sigma_s0, alpha_t * h_phi_1, alpha_t * B_h, rk, rhos

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        if self._step_index is None:
            self._step_index = _step_index_for(self.timesteps, timestep)
        i, sig = self._step_index, self.sigmas
        sample = sample.contiguous()
        if sample.dtype != torch.float32:
            raise _lib.AlgHipError("UniPC step keeps the sample in float32 (wan:500 prepares fp32 latents)")
        x0 = _lib.lincomb([(1.0, sample), (-sig[i].item(), model_output.contiguous())], torch.float32)
        use_corrector = i > 0 and (i - 1) not in self.config.disable_corrector and self.last_sample is not None
        if use_corrector:
            order = self.this_order
            r, c, k, rk, rhos = self._bh_scalars(sig[i], sig[i - 1], sig[i - 2] if order == 2 else None, order, True)
            m0 = self.model_outputs[-1]
            if order == 2:
                sample = _lib.unipc_update(self.last_sample, m0, self.model_outputs[-2], x0, r, c, k, rk,
                                           rhos[0], rhos[1])
            else:
                sample = _lib.unipc_update(self.last_sample, m0, None, x0, r, c, k, 1.0, 0.0, rhos[0])
        self.model_outputs = self.model_outputs[1:] + [x0]
        order = self.config.solver_order
        if self.config.lower_order_final:
            order = min(order, len(self.timesteps) - i)
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order
        r, c, k, rk, rhos = self._bh_scalars(sig[i + 1], sig[i], sig[i - 1] if order == 2 else None, order, False)
        if order == 2:
            out = _lib.unipc_update(sample, x0, self.model_outputs[-2], None, r, c, k, rk, rhos[0], 0.0)
        else:
            out = _lib.unipc_update(sample, x0, None, None, r, c, k)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)


class CogVideoXDPMScheduler(CogVideoXDDIMScheduler):
    """CogVideoXDPMScheduler as the reference loop drives it (cog:1111-1122: the second ``step`` signature, returning
    ``(prev_sample, pred_original_sample)``): SDE-DPM-Solver++ (2M) in the log-SNR variable over the same alpha tables
    as the DDIM scheduler, fresh noise in the sample's dtype every step.  Host scalars in float64; the per-element
    updates are ``alg_lincomb`` launches with torch-eager rounding (a 0-dim fp64 scalar times a bf16 tensor is a bf16
    tensor, the running sum is fp32); the noise comes from ``torch.randn`` on the generator's device (plumbing)."""

    def multipliers(self, timestep, timestep_back):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        lam = lambda a: ((a / (1 - a)) ** 0.5).log()
        h = lam(a_prev) - lam(a_t)
        m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_prev ** 0.5
        mn = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        m3 = m4 = None
        if timestep_back is not None:
            r = (lam(a_t) - lam(self.alphas_cumprod[int(timestep_back)])) / h
            m3, m4 = 1 + 1 / (2 * r), 1 / (2 * r)
        f = lambda v: None if v is None else float(v)
        return f(a_t ** 0.5), f((1 - a_t) ** 0.5), f(m1), f(m2), f(m3), f(m4), f(mn), prev

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta=0.0,
             use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=False):
        sa, sb, m1, m2, m3, m4, mn, prev = self.multipliers(timestep, timestep_back)
        sample = sample.contiguous()

        def noise():
            gdev = generator.device if generator is not None else sample.device
            return torch.randn(sample.shape, generator=generator, device=gdev, dtype=sample.dtype).to(sample.device)

        x0 = _lib.lincomb([(sa, sample), (-sb, model_output.contiguous())], torch.float32)
        if old_pred_original_sample is None or prev < 0:
            out = _lib.lincomb([(m1, sample), (-m2, x0), (mn, noise())], torch.float32)
        else:
            first = noise()  # the published step draws (and discards) the first-order noise before the second-order one
            del first
            d = _lib.lincomb([(m3, x0), (-m4, old_pred_original_sample.contiguous())], torch.float32)
            out = _lib.lincomb([(m1, sample), (-m2, d), (mn, noise())], torch.float32)
        if not return_dict:
            return out, x0
        return SimpleNamespace(prev_sample=out, pred_original_sample=x0)
