"""GPU parity of the HunyuanVideo DiT forward against the fp32 CPU restatement in
oracle/hy_oracle.py (parity unpinned: diffusers is absent), of its specific kernels against torch on the device, and of
the HunyuanVideo ALG sampler with the HIP DiT plugged in."""
import pytest
import torch

from alg_amd import _lib
from alg_amd.pipeline_hunyuan_video_image2video_lowpass import HunyuanVideoImageToVideoPipeline
from alg_amd.schedulers import FlowMatchEulerDiscreteScheduler
from alg_amd.transformer_hunyuan_video import (HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig,
                                               parameter_shapes)
from oracle import hy_oracle, loop_oracle
from oracle.sched_oracle import FlowMatchEulerOracle
from _parity import check_floor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def small(**over):
    kw = dict(num_attention_heads=4, num_layers=1, num_single_layers=1, num_refiner_layers=1, text_embed_dim=64,
              pooled_projection_dim=64)
    kw.update(over)
    return HunyuanVideoTransformerConfig(**kw), hy_oracle.HyConfig(**kw)


def inputs(N, F, H, W, L, valid, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 16, F, H, W, generator=g).to(BF)
    txt = torch.randn(N, L, 64, generator=g).to(BF)
    mask = torch.zeros(N, L)
    for b, v in enumerate(valid):
        mask[b, :v] = 1
    pooled = torch.randn(N, 64, generator=g).to(BF)
    return x, txt, mask, pooled


def test_headnorm_rope_masked_mean_silu():
    B, S, L, heads = 2, 48, 10, 4
    D = heads * 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, S + L, 2 * D, generator=g).to(BF).to(DEV)
    w = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).to(DEV)
    ang = torch.rand(S, 64, generator=g) * 6
    cos, sin = ang.cos().repeat_interleave(2, dim=1).to(DEV), ang.sin().repeat_interleave(2, dim=1).to(DEV)
    y = x.clone()
    _lib.headnorm_rope_(y, w, cos.contiguous(), sin.contiguous(), 2 * D, (S + L) * 2 * D, B, S + L, heads, S, 1e-6, x_off=D)
    xs = x[:, :, D:].reshape(B, S + L, heads, 128)
    var = xs.float().pow(2).mean(-1, keepdim=True)
    n = (xs.float() * torch.rsqrt(var + 1e-6)).to(BF) * w
    xr, xi = n[:, :S].reshape(B, S, heads, 64, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    roped = (n[:, :S].float() * cos[None, :, None] + rot.float() * sin[None, :, None]).to(BF)
    ref = torch.cat([roped, n[:, S:]], dim=1).reshape(B, S + L, D)
    assert torch.equal(y[:, :, :D], x[:, :, :D])
    assert (y[:, :, D:].float() - ref.float()).abs().max().item() <= 2.0 ** -5
    assert (y[:, :, D:] != ref).float().mean().item() < 0.02
    t = torch.randn(B, L, 64, generator=g).to(BF).to(DEV)
    valid = torch.tensor([7, 10], dtype=torch.int32, device=DEV)
    out = torch.empty(B, 64, dtype=BF, device=DEV)
    _lib.masked_mean(t, valid, out, B, L, 64)
    ref = torch.stack([t[0, :7].float().mean(0), t[1, :10].float().mean(0)]).to(BF)
    assert (out.float() - ref.float()).abs().max().item() <= 2.0 ** -8
    z = torch.randn(3, 100, generator=g).to(BF).to(DEV)
    assert (_lib.silu(z, torch.empty_like(z)).float() - torch.nn.functional.silu(z).float()).abs().max().item() <= 2.0 ** -7


@pytest.mark.parametrize("mode", ["token_replace", "plain_guidance"])
def test_hunyuan_forward_small(mode):
    over = dict(image_condition_type="token_replace", guidance_embeds=False) if mode == "token_replace" else \
        dict(image_condition_type="latent_concat", guidance_embeds=True)
    cfg, ocfg = small(**over)
    sd = hy_oracle.init_weights(ocfg, seed=3)
    assert set(sd) == set(parameter_shapes(cfg))
    model = HunyuanVideoTransformer3DModel(cfg, sd, device=DEV)
    x, txt, mask, pooled = inputs(2, 3, 16, 16, 20, (13, 20), 4)
    t = torch.tensor([996.0, 996.0])
    guid = torch.tensor([6000.0, 6000.0]) if cfg.guidance_embeds else None
    sd32 = {k: v.float() for k, v in sd.items()}
    ref = hy_oracle.hy_forward(ocfg, sd32, x.float(), t, txt.float(), mask, pooled.float(), guid)
    out = model(hidden_states=x.to(DEV), timestep=t.to(DEV), encoder_hidden_states=txt.to(DEV),
                encoder_attention_mask=mask.to(DEV).to(BF), pooled_projections=pooled.to(DEV),
                guidance=None if guid is None else guid.to(DEV), return_dict=False)[0]
    assert out.shape == ref.shape and out.dtype == BF
    eager = hy_oracle.hy_forward(ocfg, sd, x, t, txt, mask, pooled, guid, dtype=BF)
    check_floor("hunyuan_forward_small_" + mode, out, ref, eager)
    # padded prompt tokens are outside the contract: garbage there must not reach the latents
    txt2 = txt.clone()
    txt2[0, 13:] = 50.0
    out2 = model(x.to(DEV), t.to(DEV), txt2.to(DEV), mask.to(DEV).to(BF), pooled.to(DEV),
                 None if guid is None else guid.to(DEV), return_dict=False)[0]
    assert torch.equal(out2[1], out[1])
    check_floor("hunyuan_forward_padded_garbage_" + mode, out2[0], ref[0], eager[0])
    # the packed weights of GEMM schedule 11 (default) and the row-major ones on schedule 10 give the same bits
    assert model.packed_weights
    model.packed_weights = False
    out3 = model(x.to(DEV), t.to(DEV), txt.to(DEV), mask.to(DEV).to(BF), pooled.to(DEV), None if guid is None else guid.to(DEV),
                 return_dict=False)[0]
    model.packed_weights = True
    assert torch.equal(out3, out)


def test_hunyuan_forward_latent_tokens_not_a_multiple_of_16():
    """Bucketed resolutions give per-frame token counts like 836: the prompt's V^T columns then start inside a 16-group
    of the permuted joint layout (alg_gemm_bf16 perm_col0)."""
    cfg, ocfg = small(num_layers=2, num_single_layers=2)
    sd = hy_oracle.init_weights(ocfg, seed=11)
    model = HunyuanVideoTransformer3DModel(cfg, sd, device=DEV)
    x, txt, mask, pooled = inputs(2, 3, 12, 20, 22, (22, 5), 12)       # 3 x 6 x 10 = 180 latent tokens, 180 % 16 = 4
    t = torch.tensor([500.0, 500.0])
    ref = hy_oracle.hy_forward(ocfg, {k: v.float() for k, v in sd.items()}, x.float(), t, txt.float(), mask, pooled.float())
    out = model(x.to(DEV), t.to(DEV), txt.to(DEV), mask.to(DEV).to(BF), pooled.to(DEV), return_dict=False)[0]
    check_floor("hunyuan_forward_180_tokens", out, ref, hy_oracle.hy_forward(ocfg, sd, x, t, txt, mask, pooled, dtype=BF))


def test_hunyuan_alg_sampler_with_hip_dit():
    """hy:1127-1270 end to end with true CFG + ALG: HIP filters, first-frame token replace assembly, HIP DiT, combine,
    flow-match Euler -- vs the loop oracle driving the fp32 oracle DiT."""
    cfg, ocfg = small()
    sd = hy_oracle.init_weights(ocfg, seed=7)
    sd32 = {k: v.float() for k, v in sd.items()}
    model = HunyuanVideoTransformer3DModel(cfg, sd, device=DEV)
    g = torch.Generator().manual_seed(8)
    lat, img = torch.randn(1, 16, 3, 16, 16, generator=g), torch.randn(1, 16, 1, 16, 16, generator=g)
    mk = lambda v: (torch.randn(1, 20, 64, generator=g).to(BF), torch.randn(1, 64, generator=g).to(BF),
                    torch.cat([torch.ones(1, v), torch.zeros(1, 20 - v)], dim=1).to(BF))
    pos, neg = mk(17), mk(9)
    alg = dict(lp_filter_type="down_up", lp_resize_factor=0.625, lp_strength_schedule_type="interval",
               schedule_interval_start_time=0.0, schedule_interval_end_time=0.3)

    def oracle_dit(x, timestep, ehs, mask, pooled, guidance):
        return hy_oracle.hy_forward(ocfg, sd32, x.float(), timestep.float(), ehs.float(), mask.float(), pooled.float(),
                                    None).to(BF)

    trace_o, trace_p = [], []
    want = loop_oracle.hunyuan_denoise_loop(oracle_dit, FlowMatchEulerOracle(shift=7.0), lat, img, pos, neg, 4,
                                            true_cfg_scale=6.0, guidance_scale=1.0, use_low_pass_guidance=True,
                                            guidance_embeds=False, trace=trace_o, **alg)
    pipe = HunyuanVideoImageToVideoPipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0)).to(DEV)
    d = lambda t_: t_.to(DEV)
    out = pipe(prompt_embeds=d(pos[0]), pooled_prompt_embeds=d(pos[1]), prompt_attention_mask=d(pos[2]),
               negative_prompt_embeds=d(neg[0]), negative_pooled_prompt_embeds=d(neg[1]),
               negative_prompt_attention_mask=d(neg[2]), negative_prompt=None, image_latents=d(img), latents=d(lat),
               height=128, width=128, num_frames=9, num_inference_steps=4, true_cfg_scale=6.0, guidance_scale=1.0,
               output_type="latent", use_low_pass_guidance=True, lp_filter_in_latent=True, step_trace=trace_p, **alg)
    passes = [n for _, n, _ in trace_p]
    assert passes == [n for _, n, _ in trace_o] and passes[0] == 3 and passes[-1] == 2
    assert torch.equal(out.frames[:, :, :1].cpu(), img)
    eager = loop_oracle.hunyuan_denoise_loop(
        lambda x, ts, e, m, p_, g_: hy_oracle.hy_forward(ocfg, sd, x.to(BF), ts.float(), e, m.float(), p_, None, dtype=BF),
        FlowMatchEulerOracle(shift=7.0), lat, img, pos, neg, 4, true_cfg_scale=6.0, guidance_scale=1.0,
        use_low_pass_guidance=True, guidance_embeds=False, **alg)
    check_floor("hunyuan_sampler_4steps", out.frames[:, :, 1:], want[:, :, 1:], eager[:, :, 1:])
