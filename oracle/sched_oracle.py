"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch eager on CPU tensors) of the two third-party schedulers the
Wan and HunyuanVideo ALG loops call.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Neither scheduler is in /root/reference: they come from the `diffusers` dependency (requirements.txt pins
diffusers from git @ be2fb77dc164083bf8f033874066c96bc0a75a11, not installed in this image), so the restatement
follows the published algorithms and parity is UNPINNED (no golden vectors of the real classes can be made here):

* FlowMatchEulerDiscreteScheduler  (call sites hy:1111-1112, hy:1265-1269, run.py:82-86)
    x_prev = x + (sigma_next - sigma) * v, static shift, optional inverted sigmas.
* UniPCMultistepScheduler          (call sites wan:815-816, wan:927, run.py:63)
    UniPC-bh1/bh2 predictor-corrector (Zhao et al. 2023, Alg. 5-8) in x0-prediction form over flow sigmas
    (alpha = 1 - sigma), lower_order_final, order warm-up.

The tensor-form updates below keep the published op order (D1 = (m_i - m_0)/r_k stacks, einsum with rho, ...),
which is what the HIP `alg_unipc_update` / `alg_lincomb` kernels are checked against.
"""
from __future__ import annotations

import numpy as np
import torch


def _index_for(timesteps, t):
    cand = (timesteps == t).nonzero()
    if len(cand) == 0:
        return len(timesteps) - 1
    return (cand[1] if len(cand) > 1 else cand[0]).item()


class FlowMatchEulerOracle:
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=1.0, invert_sigmas=False):
        self.n, self.shift, self.invert = num_train_timesteps, shift, invert_sigmas
        t = np.linspace(1, self.n, self.n, dtype=np.float32)[::-1].copy()
        s = torch.from_numpy(t) / self.n
        s = shift * s / (1 + (shift - 1) * s)
        self.sigma_min, self.sigma_max = s[-1].item(), s[0].item()
        self.idx = None

    def set_timesteps(self, num_inference_steps=None, sigmas=None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max * self.n, self.sigma_min * self.n, num_inference_steps) / self.n
        else:
            sigmas = np.array(sigmas).astype(np.float32)
        sigmas = self.shift * sigmas / (1 + (self.shift - 1) * sigmas)
        s = torch.from_numpy(sigmas).to(torch.float32)
        if self.invert:
            s = 1.0 - s
            self.sigmas = torch.cat([s, torch.ones(1)])
        else:
            self.sigmas = torch.cat([s, torch.zeros(1)])
        self.timesteps = s * self.n
        self.idx = None

    def step(self, model_output, timestep, sample):
        if self.idx is None:
            self.idx = _index_for(self.timesteps, timestep)
        sample = sample.to(torch.float32)
        prev = sample + (self.sigmas[self.idx + 1] - self.sigmas[self.idx]) * model_output
        self.idx += 1
        return prev.to(model_output.dtype)


class UniPCOracle:
    order = 1

    def __init__(self, num_train_timesteps=1000, solver_order=2, solver_type="bh2", flow_shift=1.0,
                 lower_order_final=True, final_sigmas_type="zero"):
        self.n, self.solver_order, self.solver_type = num_train_timesteps, solver_order, solver_type
        self.flow_shift, self.lower_order_final, self.final = flow_shift, lower_order_final, final_sigmas_type

    def set_timesteps(self, num_inference_steps):
        alphas = np.linspace(1, 1 / self.n, num_inference_steps + 1)
        sigmas = 1.0 - alphas
        sigmas = np.flip(self.flow_shift * sigmas / (1 + (self.flow_shift - 1) * sigmas))[:-1].copy()
        timesteps = (sigmas * self.n).copy()
        last = sigmas[-1] if self.final == "sigma_min" else 0
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(torch.int64)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums, self.last_sample, self.this_order, self.idx = 0, None, None, None

    @staticmethod
    def _alpha_sigma(sigma):
        return 1 - sigma, sigma

    def _coeffs(self, sigma_t, sigma_s0, prev_idx, order):
        alpha_t, sigma_t = self._alpha_sigma(sigma_t)
        alpha_s0, sigma_s0 = self._alpha_sigma(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        m0 = self.model_outputs[-1]
        rks, D1s = [], []
        for j in range(1, order):
            mi = self.model_outputs[-(j + 1)]
            alpha_si, sigma_si = self._alpha_sigma(self.sigmas[prev_idx - (j - 1)])
            lambda_si = torch.log(alpha_si) - torch.log(sigma_si)
            rk = (lambda_si - lambda_s0) / h
            rks.append(rk)
            # published: (mi - m0) / rk with rk a CPU 0-dim tensor; on a GPU ATen evaluates tensor / cpu_scalar as a
            # multiply by the fp32 reciprocal (div_true_cuda) -- the GPU behaviour is spelled out
            D1s.append((mi - m0) * (1.0 / rk))
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, factorial_i = [], [], 1
        for j in range(1, order + 1):
            R.append(torch.pow(rks, j - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= j + 1
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        return alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, torch.stack(R), torch.tensor(b), D1s

    def _predict(self, sample, order):
        i = self.idx
        alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, R, b, D1s = self._coeffs(self.sigmas[i + 1], self.sigmas[i], i - 1,
                                                                           order)
        m0, x = self.model_outputs[-1], sample
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        if D1s:
            D1s = torch.stack(D1s, dim=1)
            rhos_p = torch.tensor([0.5], dtype=x.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            pred_res = torch.einsum("k,bkc...->bc...", rhos_p, D1s)
        else:
            pred_res = 0
        return (x_t_ - alpha_t * B_h * pred_res).to(x.dtype)

    def _correct(self, this_x0, last_sample, order):
        i = self.idx
        alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, R, b, D1s = self._coeffs(self.sigmas[i], self.sigmas[i - 1], i - 2,
                                                                           order)
        m0, x = self.model_outputs[-1], last_sample
        rhos_c = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = torch.einsum("k,bkc...->bc...", rhos_c[:-1], torch.stack(D1s, dim=1)) if D1s else 0
        D1_t = this_x0 - m0
        return (x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)).to(x.dtype)

    def step(self, model_output, timestep, sample):
        if self.idx is None:
            self.idx = _index_for(self.timesteps, timestep)
        # flow_prediction, predict_x0: x0 = sample - sigma * v.  UniPC keeps its sigmas on the CPU ("to avoid too much
        # CPU/GPU communication"), and a CPU 0-dim scalar enters a GPU bf16 multiply at fp32 precision; all-CPU eager
        # would round sigma to bf16 first, so the GPU behaviour is spelled out
        x0 = sample - (self.sigmas[self.idx] * model_output.float()).to(model_output.dtype)
        if self.idx > 0 and self.last_sample is not None:
            sample = self._correct(x0, self.last_sample, self.this_order)
        for j in range(self.solver_order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.solver_order, len(self.timesteps) - self.idx) if self.lower_order_final \
            else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.idx += 1
        return prev
