"""GPU parity of the Wan DiT forward against the fp32 CPU restatement in oracle/wan_oracle.py
(parity unpinned: diffusers is absent), and of the whole Wan ALG sampler with the HIP DiT plugged in."""
import pytest
import torch

from alg_amd.pipeline_wan_image2video_lowpass import WanImageToVideoPipeline
from alg_amd.schedulers import UniPCMultistepScheduler
from alg_amd.transformer_wan import WanTransformer3DModel, WanTransformerConfig, parameter_shapes
from oracle import loop_oracle, wan_oracle
from oracle.sched_oracle import UniPCOracle
from _parity import check_floor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def small(layers=2, heads=4, image=True):
    kw = dict(num_attention_heads=heads, ffn_dim=1024, num_layers=layers, text_dim=64, image_dim=64 if image else None,
              added_kv_proj_dim=heads * 128 if image else None)
    return WanTransformerConfig(**kw), wan_oracle.WanConfig(**kw)


def inputs(N, F, H, W, seed, n_txt=512, n_img=257, text_dim=64, image_dim=64):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 36, F, H, W, generator=g).to(BF)
    txt = torch.randn(N, n_txt, text_dim, generator=g).to(BF)
    img = torch.randn(N, n_img, image_dim, generator=g).to(BF) if n_img else None
    return x, txt, img


@pytest.mark.parametrize("N,image", [(2, True), (3, True), (2, False)])
def test_wan_forward_small(N, image):
    cfg, ocfg = small(image=image)
    sd = wan_oracle.init_weights(ocfg, seed=3)
    assert set(sd) == set(parameter_shapes(cfg))
    model = WanTransformer3DModel(cfg, sd, device=DEV)
    x, txt, img = inputs(N, 3, 16, 24, 4, n_img=257 if image else 0)
    t = torch.tensor([999.0] * N)
    ref = wan_oracle.wan_forward(ocfg, sd, x.float(), t, txt.float(), None if img is None else img.float())
    out = model(hidden_states=x.to(DEV), timestep=t.to(DEV), encoder_hidden_states=txt.to(DEV),
                encoder_hidden_states_image=None if img is None else img.to(DEV), return_dict=False)[0]
    assert out.shape == ref.shape and out.dtype == BF
    # anchored bound: the bf16-eager restatement of the published module (the reference's execution mode) vs fp32
    ref_bf = wan_oracle.wan_forward(ocfg, sd, x, t, txt, img, dtype=BF)
    check_floor("wan_forward_small_N%d_%s" % (N, "img" if image else "noimg"), out, ref, ref_bf)
    # deterministic and batch-consistent: sample 0 alone gives the same bits as sample 0 inside the batch
    out1 = model(hidden_states=x[:1].to(DEV), timestep=t[:1].to(DEV), encoder_hidden_states=txt[:1].to(DEV),
                 encoder_hidden_states_image=None if img is None else img[:1].to(DEV), return_dict=False)[0]
    assert torch.equal(out1[0], out[0])


@pytest.mark.parametrize("N", [2, 3])
def test_wan_forward_first_last_frame_image_position_embedding(N):
    """FLF2V checkpoints (`last_image=`, /root/reference/pipeline_wan_image2video_lowpass.py:603,805-812): the first and the last
    frame's CLIP tokens arrive as TWO batch rows per sample ([2 N, 257, I]); WanImageEmbedding views them as [N, 514, I], adds its
    learned position embedding, and the blocks attend 514 image tokens.  Against the fp32 oracle on the bf16-eager floor; the
    position embedding must matter (the same inputs without it give another result)."""
    kw = dict(num_attention_heads=4, ffn_dim=1024, num_layers=2, text_dim=64, image_dim=64, added_kv_proj_dim=512, pos_embed_seq_len=514)
    cfg, ocfg = WanTransformerConfig(**kw), wan_oracle.WanConfig(**kw)
    sd = wan_oracle.init_weights(ocfg, seed=7)
    assert set(sd) == set(parameter_shapes(cfg)) and sd["condition_embedder.image_embedder.pos_embed"].shape == (1, 514, 64)
    model = WanTransformer3DModel(cfg, sd, device=DEV)
    x, txt, _ = inputs(N, 3, 16, 24, 8)
    g = torch.Generator().manual_seed(80)
    img = torch.randn(2 * N, 257, 64, generator=g).to(BF)            # [first_0, last_0, first_1, last_1, ...]
    t = torch.tensor([500.0] * N)
    ref = wan_oracle.wan_forward(ocfg, sd, x.float(), t, txt.float(), img.float())
    run = lambda m, im: m(hidden_states=x.to(DEV), timestep=t.to(DEV), encoder_hidden_states=txt.to(DEV),
                          encoder_hidden_states_image=im.to(DEV), return_dict=False)[0]
    out = run(model, img)
    assert out.shape == ref.shape
    check_floor("wan_forward_flf2v_N%d" % N, out, ref, wan_oracle.wan_forward(ocfg, sd, x, t, txt, img, dtype=BF))
    sd0 = dict(sd)
    sd0["condition_embedder.image_embedder.pos_embed"] = torch.zeros_like(sd["condition_embedder.image_embedder.pos_embed"])
    assert rel(run(WanTransformer3DModel(cfg, sd0, device=DEV), img), out) > 1e-3
    with pytest.raises(ValueError, match="FLF2V"):                     # one batch row per sample: not a first / last pair
        run(model, img[:N])


def test_wan_forward_ragged_tokens_and_scalar_timestep():
    cfg, ocfg = small(layers=1, heads=8)
    sd = wan_oracle.init_weights(ocfg, seed=5)
    model = WanTransformer3DModel(cfg, sd, device=DEV)
    x, txt, img = inputs(2, 5, 14, 18, 6)          # 5 x 7 x 9 = 315 tokens: ragged q blocks and kv tiles
    t = torch.tensor(37.0)
    ref = wan_oracle.wan_forward(ocfg, sd, x.float(), t.expand(2), txt.float(), img.float())
    out = model(x.to(DEV), t, txt.to(DEV), img.to(DEV), return_dict=False)[0]
    check_floor("wan_forward_ragged", out, ref, wan_oracle.wan_forward(ocfg, sd, x, t.expand(2), txt, img, dtype=BF))


@pytest.mark.parametrize("fp8", [False, True])
def test_wan_forward_dual_cross_attention_is_bit_identical_to_two_launches_and_the_add(fp8):
    """`dual_cross` (default): the text + image cross-attention of every I2V block as ONE alg_flash_attn_d128_dual launch instead of
    two attention launches and an add pass -- the same bits, through the whole forward (ragged token count, bf16 and fp8 linears)."""
    cfg, ocfg = small(layers=2, heads=8)
    sd = wan_oracle.init_weights(ocfg, seed=5)
    model = WanTransformer3DModel(cfg, sd, device=DEV, fp8=fp8)
    # bf16: 315 tokens (ragged query blocks); fp8: 288 (the per-token scale rows of a batch item must start 16-byte aligned)
    x, txt, img = inputs(2, 3, 16, 24, 4) if fp8 else inputs(2, 5, 14, 18, 6)
    t = torch.tensor(37.0)
    assert model.dual_cross
    run = lambda: model(x.to(DEV), t, txt.to(DEV), img.to(DEV), return_dict=False)[0].clone()
    one = run()
    model.dual_cross = False
    two = run()
    assert torch.equal(one, two) and bool(torch.isfinite(one.float()).all())


def test_wan_forward_packed_weights_are_bit_identical_to_the_row_major_ones():
    """`packed_weights` (default, bf16): out / cross-q / cross-out / ff1 / ff2 on GEMM schedule 11 with the weight packed in fragment
    order -- the same bits as the row-major weights on schedule 10, through the whole forward (ragged token count)."""
    cfg, ocfg = small(layers=2, heads=8)
    sd = wan_oracle.init_weights(ocfg, seed=6)
    model = WanTransformer3DModel(cfg, sd, device=DEV)
    x, txt, img = inputs(2, 5, 14, 18, 6)
    t = torch.tensor(501.0)
    assert model.packed_weights and all(len(L.packed) == 5 for L in model.blocks)
    run = lambda: model(x.to(DEV), t, txt.to(DEV), img.to(DEV), return_dict=False)[0].clone()
    one = run()
    model.packed_weights = False
    two = run()
    assert torch.equal(one, two) and bool(torch.isfinite(one.float()).all())


def test_wan_alg_sampler_with_hip_dit():
    """wan:843-927 end to end: HIP filters + batch assembly + HIP DiT + CFG combine + UniPC, vs the loop oracle driving
    the fp32 oracle DiT on the CPU."""
    cfg, ocfg = small(layers=1)
    sd = wan_oracle.init_weights(ocfg, seed=7)
    model = WanTransformer3DModel(cfg, sd, device=DEV)
    g = torch.Generator().manual_seed(8)
    lat, cond = torch.randn(1, 16, 3, 16, 24, generator=g), torch.randn(1, 20, 3, 16, 24, generator=g)
    pe, ne = torch.randn(1, 512, 64, generator=g).to(BF), torch.randn(1, 512, 64, generator=g).to(BF)
    ie = torch.randn(1, 257, 64, generator=g).to(BF)
    alg = dict(lp_filter_type="down_up", lp_resize_factor=0.4, lp_strength_schedule_type="interval",
               schedule_interval_start_time=0.0, schedule_interval_end_time=0.3)

    def oracle_dit(x, timestep, ehs, ehs_img):
        return wan_oracle.wan_forward(ocfg, sd, x.float(), timestep.float(), ehs.float(), ehs_img.float()).to(BF)

    trace_o, trace_p = [], []
    want = loop_oracle.wan_denoise_loop(oracle_dit, UniPCOracle(flow_shift=3.0), lat, cond, pe, ne, ie, 4,
                                        guidance_scale=5.0, use_low_pass_guidance=True, trace=trace_o, **alg)
    pipe = WanImageToVideoPipeline(transformer=model, scheduler=UniPCMultistepScheduler(flow_shift=3.0)).to(DEV)
    out = pipe(prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), image_embeds=ie.to(DEV),
               image_condition=cond.to(DEV), latents=lat.to(DEV), height=128, width=192, num_frames=9,
               num_inference_steps=4, guidance_scale=5.0, output_type="latent", use_low_pass_guidance=True,
               lp_filter_in_latent=True, step_trace=trace_p, **alg)
    passes = [n for _, n, _ in trace_p]
    assert passes == [n for _, n, _ in trace_o] and passes[0] == 3 and passes[-1] == 2   # both loop branches run
    eager = loop_oracle.wan_denoise_loop(
        lambda x, ts, e, ei: wan_oracle.wan_forward(ocfg, sd, x.to(BF), ts.float(), e, ei, dtype=BF),
        UniPCOracle(flow_shift=3.0), lat, cond, pe, ne, ie, 4, guidance_scale=5.0, use_low_pass_guidance=True, **alg)
    check_floor("wan_sampler_4steps", out.frames, want, eager)


def test_wan_forward_fp8_weights():
    """BASELINE config 5: e4m3 weights (per output channel) x e4m3 activations (per token) on the fp8 MFMA for the seven
    large linears of every block.  VERDICT r4 item 1: the bound is ANCHORED like every bf16 comparison -- the fp32 oracle is
    the reference, the floor is the same oracle in the configuration's own execution mode (bf16 activations, the seven
    linears quantise-dequantised to e4m3 in eager op order: `wan_oracle.wan_forward(dtype=bf16, fp8=True)`,
    oracle/fp8_oracle.py), and HIP-fp8 has to land within 1.5 x that floor globally, 4 x its p99.9 per token, 2 x its worst
    element (tests/_parity.py).  (The scheme itself -- amax / 448 per token and per channel -- is the build's, the reference
    has no fp8 code: run.py:38,59-61 only forward a dtype.)"""
    cfg, ocfg = small()
    sd = wan_oracle.init_weights(ocfg, seed=3)
    x, txt, img = inputs(2, 3, 16, 24, 4)
    t = torch.tensor([999.0, 999.0])
    ref = wan_oracle.wan_forward(ocfg, sd, x.float(), t, txt.float(), img.float())
    run = lambda m: m(x.to(DEV), t.to(DEV), txt.to(DEV), img.to(DEV), return_dict=False)[0].cpu()
    out8 = run(WanTransformer3DModel(cfg, sd, device=DEV, fp8=True))
    out16 = run(WanTransformer3DModel(cfg, sd, device=DEV))
    eager8 = wan_oracle.wan_forward(ocfg, sd, x, t, txt, img, dtype=BF, fp8=True)
    e8, floor8 = check_floor("wan_forward_fp8_e4m3_2blocks", out8, ref, eager8)
    assert rel(out16, ref) < e8                        # quantisation costs accuracy, it does not hide it
    assert floor8 > rel(wan_oracle.wan_forward(ocfg, sd, x, t, txt, img, dtype=BF), ref)   # and the e4m3 floor is above the bf16 one
    # the norm -> e4m3 fusion (default) produces the same bytes as the separate quantiser pass
    m = WanTransformer3DModel(cfg, sd, device=DEV, fp8=True)
    m.fuse_quant = False
    assert torch.equal(run(m), out8)
