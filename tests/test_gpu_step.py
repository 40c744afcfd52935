"""GPU parity of the fused CFG-combine + DDIM-step kernel against the reference's expression sequence
(cog:1091-1123) evaluated by torch on CPU (oracle) -- same rounding points, so bit-exact is expected."""
import numpy as np
import pytest
import torch

from alg_amd import _lib
from alg_amd.schedulers import CogVideoXDDIMScheduler
from oracle import ddim_oracle

pytestmark = pytest.mark.gpu


def reference_step(pred, latents, n_pass, gs, sched, t):
    """cog:1091-1123 with torch ops (CPU)."""
    noise = pred.float()
    if n_pass == 3:
        u0, u, tx = noise.chunk(3)
        noise = u0 + gs * (tx - u)
    elif n_pass == 2:
        u, tx = noise.chunk(2)
        noise = u + gs * (tx - u)
    return sched.step(noise, t, latents).to(latents.dtype)


@pytest.mark.parametrize("n_pass", [1, 2, 3])
@pytest.mark.parametrize("pred_dtype,lat_dtype", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16),
                                                  (torch.float32, torch.float32)])
@pytest.mark.parametrize("shape", [(1, 13, 16, 60, 90), (1, 3, 16, 32, 32), (1, 1, 3, 5, 7)])
def test_cfg_ddim_step(device, n_pass, pred_dtype, lat_dtype, shape):
    g = torch.Generator().manual_seed(23 + n_pass)
    pred = torch.randn((n_pass,) + shape[1:], generator=g).to(pred_dtype)
    lat = torch.randn(shape, generator=g).to(lat_dtype)
    sched, orc = CogVideoXDDIMScheduler(), ddim_oracle.DDIMOracle()
    sched.set_timesteps(50)
    orc.set_timesteps(50)
    for t in (sched.timesteps[0], sched.timesteps[17], sched.timesteps[-1]):
        ref = reference_step(pred, lat, n_pass, 6.0, orc, t)
        out = lat.clone().to(device)
        sched.fused_cfg_step_(pred.to(device), out, n_pass, 6.0, t)
        got = out.cpu()
        assert got.dtype == lat_dtype
        if lat_dtype == torch.float32:
            assert (got - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item())
        else:
            mism = (got != ref).float().mean().item()
            ulp = (got.float() - ref.float()).abs() / ref.float().abs().clamp_min(1e-30)
            assert mism <= 1e-3 and ulp.max() <= 2.0 ** -6, (mism, ulp.max().item())


def test_generic_step_api(device):
    sched, orc = CogVideoXDDIMScheduler(), ddim_oracle.DDIMOracle()
    sched.set_timesteps(10)
    orc.set_timesteps(10)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(1, 2, 4, 8, 8, generator=g)
    x = torch.randn(1, 2, 4, 8, 8, generator=g).to(torch.bfloat16)
    t = sched.timesteps[3]
    out = sched.step(v.to(device), t, x.to(device), return_dict=False)[0]
    ref = orc.step(v, t, x).to(torch.bfloat16)
    assert out.data_ptr() != x.data_ptr()
    assert (out.cpu() != ref).float().mean() <= 1e-3
    # last step: x_prev is the predicted x0 (alpha_prev = 1)
    t_last = sched.timesteps[-1]
    sa, sb, ca, cb = sched.step_coefficients(t_last)
    assert ca == 0.0 and abs(cb - 1.0) < 1e-12
