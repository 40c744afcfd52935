"""GPU parity for the Wan / HunyuanVideo loop rows: every elementwise launch against
torch eager on the device (bit-exact), the two schedulers and the two whole loops against the CPU oracles with a
deterministic stand-in transformer (elementwise bf16 ops only, so it is bit-identical on CPU and GPU)."""
import numpy as np
import pytest
import torch

from alg_amd import _lib
from alg_amd.pipeline_hunyuan_video_image2video_lowpass import HunyuanVideoImageToVideoPipeline, assemble_first_frame
from alg_amd.pipeline_wan_image2video_lowpass import WanImageToVideoPipeline, assemble_channel_concat
from alg_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
from oracle import loop_oracle
from oracle.sched_oracle import FlowMatchEulerOracle, UniPCOracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rand(shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
@pytest.mark.parametrize("n_pass", [2, 3])
@pytest.mark.parametrize("numel", [0, 1, 1000, 16 * 21 * 60 * 104])
def test_cfg_combine_matches_torch_eager(dtype, n_pass, numel):
    pred = _rand((n_pass, numel), 11, dtype).to(DEV)
    got = _lib.cfg_combine(pred, n_pass, 5.0)
    chunks = pred.chunk(n_pass)
    want = chunks[0] + 5.0 * (chunks[-1] - chunks[-2])
    assert got.dtype == dtype and torch.equal(got, want)
    # bf16 on the CPU rounds after every op in the same places
    assert torch.equal(got.cpu(), pred.cpu().chunk(n_pass)[0] + 5.0 * (pred.cpu().chunk(n_pass)[-1]
                                                                        - pred.cpu().chunk(n_pass)[-2]))


def test_cfg_combine_rejects_bad_arguments():
    with pytest.raises(_lib.AlgHipError):
        _lib.cfg_combine(torch.zeros(4, 8, device=DEV), 3, 5.0)
    with pytest.raises(_lib.AlgHipError):
        _lib.cfg_combine(torch.zeros(4, 8), 2, 5.0)


@pytest.mark.parametrize("vd", [BF, torch.float32])
@pytest.mark.parametrize("od", [BF, torch.float32])
def test_lincomb_matches_torch_eager(vd, od):
    x, v = _rand((3, 1001), 12).to(DEV), _rand((3, 1001), 13, vd).to(DEV)
    got = _lib.lincomb([(1.0, x), (-0.0234375, v)], od)
    want = (x + (-0.0234375) * v).to(od)
    assert torch.equal(got, want)
    w = _rand((3, 1001), 14).to(DEV)
    got = _lib.lincomb([(0.75, x), (0.3, v), (-1.5, w), (2.0, x)], torch.float32)
    want = 0.75 * x + (0.3 * v) + (-1.5) * w + 2.0 * x
    assert torch.equal(got, want.float())


def test_concat_cast_channel_concat_and_first_frame():
    lat, c0, c1 = _rand((2, 16, 3, 6, 10), 15).to(DEV), _rand((2, 20, 3, 6, 10), 16).to(DEV), \
        _rand((2, 20, 3, 6, 10), 17).to(DEV)
    got = assemble_channel_concat(lat, [c0, c1, c1], BF)
    want = torch.cat([torch.cat([lat] * 3), torch.cat([c0, c1, c1], dim=0)], dim=1).to(BF)
    assert got.shape == (6, 36, 3, 6, 10) and torch.equal(got, want)
    img, lp = _rand((2, 16, 1, 6, 10), 18).to(DEV), _rand((2, 16, 1, 6, 10), 19).to(DEV)
    got = assemble_first_frame(lat, [img, lp, lp], BF)
    want = torch.cat([torch.cat([img, lp, lp], dim=0), torch.cat([lat] * 3)[:, :, 1:]], dim=2).to(BF)
    assert got.shape == (6, 16, 3, 6, 10) and torch.equal(got, want)
    got = assemble_first_frame(lat.to(BF), [img], torch.float32)  # mixed source dtypes, fp32 out (hy:1270)
    assert torch.equal(got, torch.cat([img, lat.to(BF)[:, :, 1:]], dim=2))


@pytest.mark.parametrize("case", ["p1", "p2", "c1", "c2"])
def test_unipc_update_matches_the_published_op_order(case):
    x, m0, m1, mt = (_rand((1, 16, 3, 6, 10), s).to(DEV) for s in (20, 21, 22, 23))
    r, c, k, rk, rho0, rho1 = 0.97, -0.031, -0.029, -1.7, 0.52, 0.49
    rkt = torch.tensor(rk, dtype=torch.float32)  # the published scheduler divides by a CPU 0-dim fp32 tensor
    if case == "p1":
        got, want = _lib.unipc_update(x, m0, None, None, r, c, k), (r * x - c * m0) - k * 0
    elif case == "p2":
        got = _lib.unipc_update(x, m0, m1, None, r, c, k, rk, 0.5, 0.0)
        want = (r * x - c * m0) - k * (0.5 * ((m1 - m0) / rkt))
    elif case == "c1":
        got = _lib.unipc_update(x, m0, None, mt, r, c, k, 1.0, 0.0, 0.5)
        want = (r * x - c * m0) - k * (0 + 0.5 * (mt - m0))
    else:
        got = _lib.unipc_update(x, m0, m1, mt, r, c, k, rk, rho0, rho1)
        want = (r * x - c * m0) - k * (rho0 * ((m1 - m0) / rkt) + rho1 * (mt - m0))
    # python floats enter device ops at fp32, the kernel receives the same floats rounded to fp32; tensor / cpu_scalar
    # is ATen's multiply by the fp32 reciprocal, which the kernel reproduces -> bit-exact
    assert torch.equal(got, want)


def test_flow_match_euler_scheduler_matches_oracle():
    p, o = FlowMatchEulerDiscreteScheduler(shift=7.0), FlowMatchEulerOracle(shift=7.0)
    sig = np.linspace(1.0, 0.0, 11)[:-1]
    p.set_timesteps(sigmas=sig)
    o.set_timesteps(sigmas=sig)
    x = _rand((1, 16, 4, 6, 10), 24)
    xg = x.to(DEV)
    for i, t in enumerate(o.timesteps):
        v = _rand((1, 16, 4, 6, 10), 30 + i, BF)
        x = o.step(v, t, x)
        xg = p.step(v.to(DEV), p.timesteps[i], xg, return_dict=False)[0]
        assert xg.dtype == BF and torch.equal(xg.cpu(), x), i


@pytest.mark.parametrize("order,flow_shift", [(2, 3.0), (2, 5.0), (1, 3.0)])
def test_unipc_scheduler_matches_oracle(order, flow_shift):
    p = UniPCMultistepScheduler(solver_order=order, flow_shift=flow_shift)
    o = UniPCOracle(solver_order=order, flow_shift=flow_shift)
    p.set_timesteps(12)
    o.set_timesteps(12)
    x = _rand((1, 16, 3, 6, 10), 25)
    xg = x.to(DEV)
    worst = 0.0
    for i, t in enumerate(o.timesteps):
        v = _rand((1, 16, 3, 6, 10), 50 + i, BF)
        x = o.step(v, t, x)
        xg = p.step(v.to(DEV), p.timesteps[i], xg, return_dict=False)[0]
        assert xg.dtype == torch.float32
        worst = max(worst, (xg.cpu() - x).abs().max().item())
    # same op order in fp32; only libm (log / expm1) of the host scalars could differ -- they run on the host in both
    assert worst <= 1e-6, worst


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class WanStandIn:
    """Elementwise bf16 'DiT' (bit-identical on CPU and GPU): mixes noisy latents, condition latents, mask, prompt and
    timestep so every input of the loop reaches the output."""
    dtype = BF
    config = _Cfg(patch_size=(1, 2, 2))

    def __call__(self, hidden_states=None, timestep=None, encoder_hidden_states=None, encoder_hidden_states_image=None,
                 attention_kwargs=None, return_dict=False):
        return (self.f(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image),)

    @staticmethod
    def f(x, timestep, ehs, ehs_image):
        n = x.shape[0]
        tf = (timestep.float() / 1024.0).to(BF).view(n, 1, 1, 1, 1).to(x.device)
        e = (ehs[:, 0, :1] + ehs_image[:, 0, :1]).view(n, 1, 1, 1, 1)
        y = x[:, :16] * 0.5
        y = y - x[:, 20:36] * 0.25
        y = y + x[:, 16:17] * 0.125
        y = y + e * 0.0625
        return y + tf * 0.03125


class HunyuanStandIn:
    dtype = BF

    def __init__(self, image_condition_type="token_replace"):
        self.config = _Cfg(image_condition_type=image_condition_type, in_channels=16, guidance_embeds=True,
                           patch_size=2)

    def __call__(self, hidden_states=None, timestep=None, encoder_hidden_states=None, encoder_attention_mask=None,
                 pooled_projections=None, guidance=None, attention_kwargs=None, return_dict=False):
        return (self.f(hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                       guidance),)

    @staticmethod
    def f(x, timestep, ehs, mask, pooled, guidance):
        n = x.shape[0]
        assert timestep.dtype == BF and guidance.dtype == BF
        tf = (timestep.view(n, 1, 1, 1, 1) * 0.0009765625).to(x.device)
        e = (ehs[:, 0, :1] * mask[:, :1] + pooled[:, :1]).view(n, 1, 1, 1, 1)
        y = x * 0.5
        y = y - x[:, :, :1] * 0.25          # every frame sees the (possibly low-passed) first frame
        y = y + e * 0.0625
        return y + tf * (guidance.view(-1, 1, 1, 1, 1)[:1] * 0.0000152587890625)


ALG_WAN = dict(use_low_pass_guidance=True, lp_filter_type="down_up", lp_filter_in_latent=True, lp_resize_factor=0.4,
               lp_strength_schedule_type="interval", schedule_interval_start_time=0.0, schedule_interval_end_time=0.2)


@pytest.mark.parametrize("alg", [True, False])
def test_wan_loop_matches_oracle(alg):
    lat, cond = _rand((1, 16, 3, 30, 52), 60), _rand((1, 20, 3, 30, 52), 61)
    pe, ne, ie = _rand((1, 4, 8), 62, BF), _rand((1, 4, 8), 63, BF), _rand((1, 5, 8), 64, BF)
    kw = dict(ALG_WAN) if alg else {}
    trace_o, trace_p = [], []
    want = loop_oracle.wan_denoise_loop(WanStandIn.f, UniPCOracle(flow_shift=3.0), lat, cond, pe, ne, ie, 10,
                                        guidance_scale=5.0, use_low_pass_guidance=alg, trace=trace_o,
                                        **{k: v for k, v in kw.items() if k not in ("use_low_pass_guidance",
                                                                                     "lp_filter_in_latent")})
    pipe = WanImageToVideoPipeline(transformer=WanStandIn(), scheduler=UniPCMultistepScheduler(flow_shift=3.0)).to(DEV)
    out = pipe(prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), image_embeds=ie.to(DEV),
               image_condition=cond.to(DEV), latents=lat.to(DEV), height=240, width=416, num_frames=9,
               num_inference_steps=10, guidance_scale=5.0, output_type="latent", step_trace=trace_p, **kw)
    got = out.frames.cpu()
    assert [n for _, n, _ in trace_p] == [n for _, n, _ in trace_o] == ([3, 3] + [2] * 8 if alg else [2] * 10)
    assert got.dtype == torch.float32 and got.shape == lat.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-5, err  # fp32 latents of O(1); a 1-ulp filter difference passes through 10 linear steps


@pytest.mark.parametrize("sched", ["linear", "exponential"])
def test_wan_loop_gaussian_blur_schedules(sched):
    """BASELINE config 3 style: gaussian_blur in latent with a decaying strength (every step filters differently)."""
    lat, cond = _rand((1, 16, 3, 30, 52), 65), _rand((1, 20, 3, 30, 52), 66)
    pe, ne, ie = _rand((1, 4, 8), 67, BF), _rand((1, 4, 8), 68, BF), _rand((1, 5, 8), 69, BF)
    kw = dict(lp_filter_type="gaussian_blur", lp_blur_sigma=3.0, lp_blur_kernel_size=9,
              lp_strength_schedule_type=sched, schedule_blur_kernel_size=False, schedule_linear_start_weight=1.0,
              schedule_linear_end_weight=0.0, schedule_linear_end_time=0.5, schedule_exp_decay_rate=6.0)
    trace_o, trace_p = [], []
    want = loop_oracle.wan_denoise_loop(WanStandIn.f, UniPCOracle(flow_shift=3.0), lat, cond, pe, ne, ie, 8,
                                        guidance_scale=5.0, use_low_pass_guidance=True, trace=trace_o, **kw)
    pipe = WanImageToVideoPipeline(transformer=WanStandIn(), scheduler=UniPCMultistepScheduler(flow_shift=3.0)).to(DEV)
    out = pipe(prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), image_embeds=ie.to(DEV),
               image_condition=cond.to(DEV), latents=lat.to(DEV), height=240, width=416, num_frames=9,
               num_inference_steps=8, guidance_scale=5.0, output_type="latent", step_trace=trace_p,
               use_low_pass_guidance=True, lp_filter_in_latent=True, **kw)
    assert [(s, n) for s, n, _ in trace_p] == [(s, n) for s, n, _ in trace_o]      # strengths bit-exact, same branches
    assert 3 in [n for _, n, _ in trace_p]
    # the filters agree to ~5e-7 (fp32), but the stand-in DiT computes in bf16: a last-bit difference in the filtered
    # condition can flip one bf16 rounding of the prediction (2^-8 x guidance 5 x dt) -> isolated 1e-3 outliers
    d = (out.frames.cpu() - want).abs()
    assert d.max().item() <= 3e-3 and d.mean().item() <= 2e-5, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("true_cfg,alg,noisy", [(6.0, True, False), (6.0, True, True), (6.0, False, False),
                                                (1.0, False, False), (1.0, True, False)])
def test_hunyuan_loop_matches_oracle(true_cfg, alg, noisy):
    lat, img = _rand((1, 16, 4, 30, 52), 70), _rand((1, 16, 1, 30, 52), 71)
    pos = (_rand((1, 4, 8), 72, BF), _rand((1, 8), 73, BF), torch.ones(1, 4, dtype=BF))
    neg = (_rand((1, 4, 8), 74, BF), _rand((1, 8), 75, BF), torch.ones(1, 4, dtype=BF))
    lp = dict(lp_filter_type="down_up", lp_resize_factor=0.625, lp_strength_schedule_type="interval",
              schedule_interval_start_time=0.0, schedule_interval_end_time=0.25)
    trace_o, trace_p = [], []
    want = loop_oracle.hunyuan_denoise_loop(HunyuanStandIn.f, FlowMatchEulerOracle(shift=7.0), lat, img, pos, neg, 8,
                                            true_cfg_scale=true_cfg, guidance_scale=6.0, use_low_pass_guidance=alg,
                                            lp_on_noisy_latent=noisy, trace=trace_o, **lp)
    pipe = HunyuanVideoImageToVideoPipeline(transformer=HunyuanStandIn(),
                                            scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0)).to(DEV)
    d = lambda t: t.to(DEV)
    out = pipe(prompt_embeds=d(pos[0]), pooled_prompt_embeds=d(pos[1]), prompt_attention_mask=d(pos[2]),
               negative_prompt_embeds=d(neg[0]), negative_pooled_prompt_embeds=d(neg[1]),
               negative_prompt_attention_mask=d(neg[2]), negative_prompt=None, image_latents=d(img), latents=d(lat),
               height=240, width=416, num_frames=13, num_inference_steps=8, true_cfg_scale=true_cfg,
               guidance_scale=6.0, output_type="latent", use_low_pass_guidance=alg, lp_filter_in_latent=True,
               lp_on_noisy_latent=noisy, step_trace=trace_p, **lp)
    got = out.frames.cpu()
    assert [n for _, n, _ in trace_p] == [n for _, n, _ in trace_o]
    assert got.dtype == torch.float32 and got.shape == lat.shape
    assert torch.equal(got[:, :, :1], img)
    # frames 1.. carry bf16-rounded values (the Euler step returns the prediction dtype): a 1-ulp fp32 difference in
    # the filtered frame can flip a bf16 rounding, so allow one bf16 ulp on O(1) values
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -6, err
    assert (got != want).float().mean().item() < 0.01


def test_lincomb_bf16_vector_path_and_aliasing():
    a, b = _rand((4, 4096), 90, BF).to(DEV), _rand((4, 4096), 91, BF).to(DEV)
    want = a + b
    got = _lib.lincomb([(1.0, a), (1.0, b)], BF, out=a)   # in place on the first term (Wan cross-attention sum)
    assert got is a and torch.equal(a, want)
    c = _rand((3, 1000), 92, BF).to(DEV)
    assert torch.equal(_lib.lincomb([(0.5, c), (-2.0, c)], BF), (0.5 * c + (-2.0) * c))
