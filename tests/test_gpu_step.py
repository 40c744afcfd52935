"""GPU parity of the fused CFG-combine + DDIM-step kernel against the reference's expression sequence
(cog:1091-1123), evaluated by torch through the oracle scheduler in two ways:

  * on DEVICE tensors -- exactly what the reference executes on a GPU: the scheduler's 0-dim float64 scalars are CPU
    tensors, and a GPU elementwise op reads such a scalar as fp32 (opmath) without rounding it to the tensor dtype.
    The kernel reproduces these rounding points -> bit-exact expected;
  * on CPU tensors -- torch's CPU TensorIterator first casts the 0-dim scalar to the common dtype (bf16!), so the CPU
    result differs from the GPU one by up to ~1 bf16 ulp.  Checked within the stated bf16 tolerance (2 ulps)."""
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
        ref_cpu = reference_step(pred, lat, n_pass, 6.0, orc, t)
        ref_dev = reference_step(pred.to(device), lat.to(device), n_pass, 6.0, orc, t).cpu()
        out = lat.clone().to(device)
        sched.fused_cfg_step_(pred.to(device), out, n_pass, 6.0, t)
        got = out.cpu()
        assert got.dtype == lat_dtype
        if lat_dtype == torch.float32:
            assert (got - ref_cpu).abs().max() <= 2e-6 * max(1.0, ref_cpu.abs().max().item())
            assert (got - ref_dev).abs().max() <= 2e-6 * max(1.0, ref_dev.abs().max().item())
        else:
            assert torch.equal(got, ref_dev), (got != ref_dev).float().mean().item()  # the reference's GPU semantics
            # CPU semantics: each bf16-rounded scalar moves its term by <= 2^-8 relative -> bound on the terms' sizes
            err = (got.float() - ref_cpu.float()).abs()
            sa, sb, ca, cb = (float(c) for c in orc.coefficients(t))
            v = pred.float()
            if n_pass == 3:
                v = v[0:1] + 6.0 * (v[2:3] - v[1:2])
            elif n_pass == 2:
                v = v[0:1] + 6.0 * (v[1:2] - v[0:1])
            x = lat.float()
            terms = (ca * x).abs() + abs(cb) * ((sa * x).abs() + (sb * v).abs()) + ref_cpu.float().abs()
            assert (err <= 2.0 ** -7 * terms + 1e-6).all()


def test_generic_step_api(device):
    sched, orc = CogVideoXDDIMScheduler(), ddim_oracle.DDIMOracle()
    sched.set_timesteps(10)
    orc.set_timesteps(10)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(1, 2, 4, 8, 8, generator=g)
    x = torch.randn(1, 2, 4, 8, 8, generator=g).to(torch.bfloat16)
    t = sched.timesteps[3]
    out = sched.step(v.to(device), t, x.to(device), return_dict=False)[0]
    ref = orc.step(v.to(device), t, x.to(device)).to(torch.bfloat16).cpu()
    assert torch.equal(out.cpu(), ref)
    # last step: x_prev is the predicted x0 (alpha_prev = 1)
    t_last = sched.timesteps[-1]
    sa, sb, ca, cb = sched.step_coefficients(t_last)
    assert ca == 0.0 and abs(cb - 1.0) < 1e-12
