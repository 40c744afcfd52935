"""CPU checks for the Wan / HunyuanVideo loop rows: scheduler restatements, host
scalars of the product schedulers against the oracle's tensor-form coefficients, branch tables of the loop oracles,
and the boundary behaviour of the two pipeline classes without a GPU."""
import numpy as np
import pytest
import torch

from alg_amd import _lib
from alg_amd.pipeline_hunyuan_video_image2video_lowpass import HunyuanVideoImageToVideoPipeline
from alg_amd.pipeline_wan_image2video_lowpass import WanImageToVideoPipeline
from alg_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
from oracle import loop_oracle
from oracle.sched_oracle import FlowMatchEulerOracle, UniPCOracle


def test_unipc_sigma_table_and_timesteps():
    o = UniPCOracle(flow_shift=3.0)
    o.set_timesteps(50)
    p = UniPCMultistepScheduler(flow_shift=3.0)
    p.set_timesteps(50)
    assert torch.equal(o.sigmas, p.sigmas) and torch.equal(o.timesteps, p.timesteps)
    s = o.sigmas
    assert s.shape == (51,) and s[-1] == 0 and bool((s[1:] < s[:-1]).all())
    # shift: sigma' = 3 s / (1 + 2 s) at s = 1 - 1/1000 (first alpha of linspace(1, 1/1000, 51) flipped)
    s0 = 1 - 1 / 1000
    assert abs(s[0].item() - 3 * s0 / (1 + 2 * s0)) < 1e-6
    assert o.timesteps.dtype == torch.int64 and o.timesteps[0].item() == 999
    assert len(set(o.timesteps.tolist())) == 50


@pytest.mark.parametrize("solver_type", ["bh1", "bh2"])
@pytest.mark.parametrize("order", [1, 2])
def test_unipc_oracle_is_exact_on_a_straight_flow(order, solver_type):
    """x_t = (1 - s) x0 + s eps, v = eps - x0: the x0-prediction is exact at every step, so UniPC of any order must
    land on x0 (and pass through the exact marginals)."""
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(1, 4, 2, 6, 5, generator=g), torch.randn(1, 4, 2, 6, 5, generator=g)
    o = UniPCOracle(solver_order=order, solver_type=solver_type, flow_shift=5.0)
    o.set_timesteps(12)
    x = (1 - o.sigmas[0]) * x0 + o.sigmas[0] * eps
    for i, t in enumerate(o.timesteps):
        x = o.step(eps - x0, t, x)
        if solver_type == "bh1" and i == 11:
            break  # B(h) = h is -inf at sigma = 0: the published bh1 update is 0 * inf there (bh2 is what Wan ships)
        want = (1 - o.sigmas[i + 1]) * x0 + o.sigmas[i + 1] * eps
        assert torch.allclose(x, want, atol=2e-5), i


def test_flow_match_euler_oracle_tables_and_exactness():
    o = FlowMatchEulerOracle(shift=7.0)
    o.set_timesteps(sigmas=np.linspace(1.0, 0.0, 11)[:-1])
    p = FlowMatchEulerDiscreteScheduler(shift=7.0, flow_shift=7.0)
    p.set_timesteps(sigmas=np.linspace(1.0, 0.0, 11)[:-1])
    assert torch.equal(o.sigmas, p.sigmas) and torch.equal(o.timesteps, p.timesteps)
    assert o.sigmas[0] == 1.0 and o.sigmas[-1] == 0.0 and o.timesteps[0] == 1000.0
    assert abs(o.sigmas[5].item() - 7 * 0.5 / (1 + 6 * 0.5)) < 1e-6
    inv = FlowMatchEulerDiscreteScheduler(shift=7.0, invert_sigmas=True)
    inv.set_timesteps(sigmas=np.linspace(1.0, 0.0, 11)[:-1])
    assert inv.sigmas[0] == 0.0 and inv.sigmas[-1] == 1.0
    g = torch.Generator().manual_seed(4)
    x0, eps = torch.randn(2, 3, 4, generator=g), torch.randn(2, 3, 4, generator=g)
    x = eps.clone()
    for t in o.timesteps:
        x = o.step(eps - x0, t, x)
    assert torch.allclose(x, x0, atol=1e-5)
    # default (no custom sigmas) path
    o.set_timesteps(num_inference_steps=8)
    p.set_timesteps(num_inference_steps=8)
    assert torch.equal(o.sigmas, p.sigmas)


def test_unipc_host_scalars_match_the_oracle_tensor_form():
    """The product computes (r, c, k, rk, rhos) on the host and hands them to alg_unipc_update; the oracle keeps the
    published tensor form.  Both must describe the same update: check on scalar 'tensors'."""
    p = UniPCMultistepScheduler(flow_shift=3.0)
    p.set_timesteps(9)
    o = UniPCOracle(flow_shift=3.0)
    o.set_timesteps(9)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 1, 7, generator=g)
    hist = [torch.randn(1, 1, 7, generator=g) for _ in range(3)]
    for i in range(9):
        order = 1 if i in (0, 8) else 2
        o.idx, o.model_outputs = i, [hist[0], hist[1]]
        want = o._predict(x, order)
        r, c, k, rk, rhos = p._bh_scalars(p.sigmas[i + 1], p.sigmas[i], p.sigmas[i - 1] if order == 2 else None,
                                          order, False)
        got = r * x - c * hist[1]
        if order == 2:
            got = got - k * (rhos[0] * ((hist[0] - hist[1]) / rk))
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), ("predict", i)
        if i == 0:
            continue
        order = 1 if i == 1 else 2  # the corrector reuses the order of the previous step's predictor
        want = o._correct(hist[2], x, order)
        r, c, k, rk, rhos = p._bh_scalars(p.sigmas[i], p.sigmas[i - 1], p.sigmas[i - 2] if order == 2 else None,
                                          order, True)
        res = rhos[-1] * (hist[2] - hist[1])
        if order == 2:
            res = rhos[0] * ((hist[0] - hist[1]) / rk) + res
        got = r * x - c * hist[1] - k * res
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), ("correct", i)


def _wan_standin(x, timestep, ehs, ehs_image):
    return x[:, :16] * 0.5


def _hy_standin(x, timestep, ehs, mask, pooled, guidance):
    return x * 0.5


def test_wan_loop_oracle_branch_table():
    g = torch.Generator().manual_seed(6)
    lat, cond = torch.randn(1, 16, 3, 8, 10, generator=g), torch.randn(1, 20, 3, 8, 10, generator=g)
    pe, ne = torch.randn(1, 4, 8).bfloat16(), torch.randn(1, 4, 8).bfloat16()
    ie = torch.randn(1, 5, 8).bfloat16()
    trace = []
    out = loop_oracle.wan_denoise_loop(_wan_standin, UniPCOracle(flow_shift=3.0), lat, cond, pe, ne, ie, 10,
                                       trace=trace, lp_filter_type="down_up", lp_resize_factor=0.4,
                                       lp_strength_schedule_type="interval", schedule_interval_end_time=0.2)
    assert out.shape == lat.shape and out.dtype == torch.float32
    # wan_alg.yaml: interval [0, 0.2] of 10 steps -> steps 0 and 1 are 3-pass (pinned schedule, tests/golden)
    assert [n for _, n, _ in trace] == [3, 3] + [2] * 8
    with pytest.raises(UnboundLocalError):
        loop_oracle.wan_denoise_loop(_wan_standin, UniPCOracle(), lat, cond, pe, ne, ie, 4, guidance_scale=1.0)


@pytest.mark.parametrize("true_cfg,alg,noisy,want", [
    (6.0, True, False, [3, 2, 2, 2, 2]), (6.0, True, True, [2] * 5), (6.0, False, False, [2] * 5),
    (1.0, False, False, [1] * 5), (1.0, True, False, [1] * 5)])
def test_hunyuan_loop_oracle_branch_table(true_cfg, alg, noisy, want):
    g = torch.Generator().manual_seed(7)
    lat, img = torch.randn(1, 16, 3, 8, 10, generator=g), torch.randn(1, 16, 1, 8, 10, generator=g)
    mk = lambda: (torch.randn(1, 4, 8).bfloat16(), torch.randn(1, 8).bfloat16(), torch.ones(1, 4).bfloat16())
    trace = []
    out = loop_oracle.hunyuan_denoise_loop(_hy_standin, FlowMatchEulerOracle(shift=7.0), lat, img, mk(), mk(), 5,
                                           true_cfg_scale=true_cfg, guidance_scale=6.0, use_low_pass_guidance=alg,
                                           lp_on_noisy_latent=noisy, trace=trace, lp_filter_type="down_up",
                                           lp_resize_factor=0.625, lp_strength_schedule_type="interval",
                                           schedule_interval_end_time=0.04)
    assert [n for _, n, _ in trace] == want
    assert out.shape == lat.shape and out.dtype == torch.float32
    assert torch.equal(out[:, :, :1], img)  # token replace: the clean first frame is re-prepended every step


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _T:
    dtype = torch.bfloat16

    def __init__(self, **kw):
        self.config = _Cfg(**kw)


def test_wan_pipeline_boundary_without_gpu():
    pipe = WanImageToVideoPipeline(transformer=_T(patch_size=(1, 2, 2)), scheduler=UniPCMultistepScheduler())
    pe = torch.zeros(1, 4, 8)
    with pytest.raises(ValueError, match="divisible by 16"):
        pipe(prompt_embeds=pe, image_embeds=pe, height=100, width=832)
    with pytest.raises(ValueError, match="Provide either `image`"):
        pipe(prompt_embeds=pe)
    with pytest.raises(ValueError, match="Cannot forward both `image`"):
        pipe(image=torch.zeros(1), image_embeds=pe, prompt_embeds=pe)
    with pytest.raises(ValueError, match="Provide either `prompt`"):
        pipe(image_embeds=pe)
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        pipe(prompt_embeds=pe, negative_prompt_embeds=pe, image_embeds=pe)


def test_hunyuan_pipeline_boundary_without_gpu():
    t = _T(image_condition_type="token_replace", in_channels=16, guidance_embeds=True, patch_size=2)
    pipe = HunyuanVideoImageToVideoPipeline(transformer=t, scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0))
    pe = torch.zeros(1, 4, 8)
    with pytest.raises(ValueError, match="divisible by 16"):
        pipe(prompt_embeds=pe, height=100)
    with pytest.raises(ValueError, match="Provide either `prompt`"):
        pipe()
    with pytest.raises(ValueError, match="has to contain a key `template`"):
        pipe(prompt_embeds=pe, prompt_template={})
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        pipe(prompt_embeds=pe)


def test_wan_condition_builder():
    """wan:439-456: [mask4 | latent16]; only the first latent frame carries the 4 mask flags (plus the last one for
    first-last-frame conditioning)."""
    from alg_amd.pipeline_wan_image2video_lowpass import build_wan_condition
    lat = torch.randn(2, 16, 21, 6, 8)
    c = build_wan_condition(lat, 81)
    assert c.shape == (2, 20, 21, 6, 8) and torch.equal(c[:, 4:], lat)
    assert bool((c[:, :4, 0] == 1).all()) and bool((c[:, :4, 1:] == 0).all())
    c2 = build_wan_condition(lat, 81, has_last_image=True)
    assert bool((c2[:, :4, 0] == 1).all()) and bool((c2[:, :4, 1:20] == 0).all())
    assert bool((c2[:, 3, 20] == 1).all()) and bool((c2[:, :3, 20] == 0).all())   # pixel frame 80 = last slot of group 20


def test_wan_prompt_encoding_trims_and_zero_pads():
    """wan:185-226 with stand-in tokenizer / encoder: rows past each prompt's length are zero, cleaning collapses
    whitespace and HTML entities, negative prompts default to ''."""
    from alg_amd.pipeline_wan_image2video_lowpass import prompt_clean
    assert prompt_clean("  a &amp;amp; b \n\t c ") == "a & b c"

    class Tok:
        def __call__(self, texts, max_length=None, **_):
            ids = torch.zeros(len(texts), max_length, dtype=torch.long)
            mask = torch.zeros(len(texts), max_length, dtype=torch.long)
            for i, t in enumerate(texts):
                n = min(len(t) + 1, max_length)
                ids[i, :n] = torch.arange(1, n + 1)
                mask[i, :n] = 1
            return type("Enc", (), {"input_ids": ids, "attention_mask": mask})()

    class Enc:
        dtype = torch.float32

        def __call__(self, ids, mask):
            assert mask is not None
            return type("Out", (), {"last_hidden_state": torch.ones(ids.shape + (4,)) * ids[..., None].float() + 100})()

    pipe = WanImageToVideoPipeline(tokenizer=Tok(), text_encoder=Enc(), transformer=_T(patch_size=(1, 2, 2)),
                                   scheduler=UniPCMultistepScheduler())
    pe, ne = pipe.encode_prompt(["abc", "  abcdefg  "], None, True, 1, None, None, 12, torch.device("cpu"))
    assert pe.shape == ne.shape == (2, 12, 4)
    assert bool((pe[0, :4] > 100).all()) and bool((pe[0, 4:] == 0).all()) and bool((pe[1, :8] > 100).all())
    assert bool((ne[:, :1] > 100).all()) and bool((ne[:, 1:] == 0).all())          # '' -> just the end-of-sequence token
    with pytest.raises(ValueError, match="batch size"):
        pipe.encode_prompt(["a", "b"], ["x"], True, 1, None, None, 12, torch.device("cpu"))


def test_pixel_branch_oracle_structure():
    """oracle/loop_oracle.py's restatement of the pixel-space ALG branch (cog:628-680, wan:493-540), host-side facts only: one
    posterior draw per call in the VAE's dtype stream, zero frames / mask channels where the reference puts them, the identity
    filter still re-encodes, and the Wan mask equals the product's condition builder."""
    from alg_amd.pipeline_wan_image2video_lowpass import build_wan_condition
    calls = []

    def moments(x):                                   # stand-in encoder: 8x spatial mean pool -> 2 x 16 moment planes
        calls.append(tuple(x.shape))
        b, _, t, h, w = x.shape
        t_lat = 1 + (t - 1) // 4
        m = torch.nn.functional.avg_pool2d(x[:, :, 0], 8).mean(1, keepdim=True)          # [B, 1, h/8, w/8]
        return m[:, :, None].expand(b, 32, t_lat, h // 8, w // 8).contiguous() * 0.5

    img = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(0)) * 2 - 1
    g = torch.Generator().manual_seed(1)
    a = loop_oracle.prepare_lp_pixel_cog(img, moments, g, 9, "down_up", 0.0, 0, 0.25, torch.float32)
    assert a.shape == (1, 3, 16, 4, 6) and bool((a[:, 1:] == 0).all()) and calls[-1] == (1, 3, 1, 32, 48)
    b = loop_oracle.prepare_lp_pixel_cog(img, moments, g, 9, "down_up", 0.0, 0, 0.25, torch.float32)
    assert not torch.equal(a, b)                                                          # fresh noise on every call (cog:645)
    g2 = torch.Generator().manual_seed(1)
    a2 = loop_oracle.prepare_lp_pixel_cog(img, moments, g2, 9, "down_up", 0.0, 0, 0.25, torch.float32)
    assert torch.equal(a, a2)
    # the draw follows the VAE's dtype: bf16 noise is NOT fp32 noise rounded (torch CPU generator), hence `noise_dtype`
    n_bf = loop_oracle.posterior_sample(torch.zeros(1, 32, 1, 4, 6), torch.Generator().manual_seed(3), torch.bfloat16)
    n_32 = loop_oracle.posterior_sample(torch.zeros(1, 32, 1, 4, 6), torch.Generator().manual_seed(3))
    assert torch.equal(n_bf, torch.randn(1, 16, 1, 4, 6, generator=torch.Generator().manual_seed(3), dtype=torch.bfloat16).float())
    assert not torch.equal(n_bf, n_32.bfloat16().float())
    # CogVideoX 1.5: 3 latent frames, patch_size_t 2 -> the leading frame repeated in front (cog:673-680)
    c = loop_oracle.prepare_lp_pixel_cog(img, moments, torch.Generator().manual_seed(1), 9, "none", 0.0, 0, 1.0,
                                         torch.float32, patch_size_t=2)
    assert c.shape == (1, 4, 16, 4, 6) and torch.equal(c[:, 0], c[:, 1]) and bool((c[:, 2:] == 0).all())
    # Wan: [mask4 | (z - mean) / std], the condition video is [image_lp, zeros x (F - 1)], sampled (not the mode)
    mean, std = [0.1] * 16, [2.0] * 16
    w = loop_oracle.prepare_lp_pixel_wan(img, moments, torch.Generator().manual_seed(2), 9, "gaussian_blur", 2.0, 5, 1.0,
                                         torch.float32, mean, std)
    assert w.shape == (1, 20, 3, 4, 6) and calls[-1] == (1, 3, 9, 32, 48)
    assert bool((w[:, :4, 0] == 1).all()) and bool((w[:, :4, 1:] == 0).all())
    assert torch.equal(w, build_wan_condition(w[:, 4:], 9))
    mode = (torch.chunk(moments(torch.cat([loop_oracle.apply_low_pass_filter_torch(img, "gaussian_blur", 2.0, 5, 1.0)[:, :, None],
                                           torch.zeros(1, 3, 8, 32, 48)], 2)), 2, 1)[0] - 0.1) * 0.5
    assert not torch.equal(w[:, 4:], mode)
