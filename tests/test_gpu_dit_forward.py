"""GPU self-consistency of the whole DiT forward and of the ALG sampler against the CPU oracle (fp32 restatement).
The DiT arithmetic itself is parity-UNPINNED (diffusers is not in the reference tree): these tests check the HIP
path against our own restatement of the published architecture, at sizes the CPU finishes in seconds."""
import numpy as np
import pytest
import torch

import alg_amd
from alg_amd import CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel
from alg_amd.transformer_cogvideox import CogVideoXTransformerConfig
from oracle import ddim_oracle, dit_oracle, loop_oracle
from _parity import check_floor

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

SMALL = dict(num_attention_heads=8, attention_head_dim=64, in_channels=16, out_channels=8, num_layers=2,
             time_embed_dim=64, text_embed_dim=128, max_text_seq_length=10, sample_width=12, sample_height=8,
             sample_frames=9, patch_size=2)


def make_pair(device, overrides=None, seed=3):
    kw = dict(SMALL, **(overrides or {}))
    ocfg = dit_oracle.DiTConfig(**kw)
    w32 = dit_oracle.init_weights(ocfg, seed=seed, std=0.05, randomize_affine=True)
    wbf = {k: v.to(BF) for k, v in w32.items()}
    w_ref = {k: v.float() for k, v in wbf.items()}  # the oracle sees the same bf16-rounded weights, in fp32
    model = CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**kw), wbf, device=device)
    model.w_bf16 = wbf   # the bf16-eager oracle (the reference's execution mode) runs on these
    return ocfg, w_ref, model


def eager_bf16(ocfg, model, hs, ehs, ts, rope, **kw):
    """The reference's own execution mode on the same inputs: bf16 weights and activations, eager op order (CPU)."""
    return dit_oracle.dit_forward(ocfg, model.w_bf16, hs.to(BF), ehs.to(BF), ts, rope, **kw)


def rel(got, ref):
    return ((got.double() - ref.double()).norm() / ref.double().norm()).item()


def test_dit_forward_small(device):
    ocfg, w, model = make_pair(device)
    g = torch.Generator().manual_seed(1)
    N, Fr, C, H, W = 3, 3, 8, 8, 12
    hs = torch.randn(N, Fr, 2 * C, H, W, generator=g).to(BF)
    ehs = torch.randn(N, 10, 128, generator=g).to(BF)
    ts = torch.tensor([999, 999, 999])
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr)
    col = {}
    ref = dit_oracle.dit_forward(ocfg, w, hs.float(), ehs.float(), ts, rope, collect=col)
    out = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    assert out.shape == ref.shape and out.dtype == BF
    check_floor("cog_forward_small_2layers", out, ref, eager_bf16(ocfg, model, hs, ehs, ts, rope), channel_dim=2)
    # the folded batch assembly gives the same result as the materialised concat
    lat, conds = hs[:1, :, :C], [hs[n:n + 1, :, C:] for n in range(N)]
    hs2 = torch.cat([torch.cat([lat] * N), torch.cat(conds)], dim=2)
    out2 = model.forward_assembled(lat.to(device), [c.to(device) for c in conds], ehs.to(device), ts, rope)
    ref2 = model(hs2.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    assert torch.equal(out2, ref2)
    # determinism
    out3 = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    assert torch.equal(out, out3)
    # the packed weights of GEMM schedule 11 (default) and the row-major ones on schedule 10 give the same bits
    assert model.packed_weights
    model.packed_weights = False
    try:
        out4 = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    finally:
        model.packed_weights = True
    assert torch.equal(out, out4)


def test_dit_forward_wider_and_ragged_tokens(device):
    """More heads than one XCD slot group, token count not a multiple of any tile (S = 7 + 3*5*7 = 112 ... )."""
    over = dict(num_attention_heads=16, num_layers=1, max_text_seq_length=7, sample_width=14, sample_height=10)
    ocfg, w, model = make_pair(device, over, seed=8)
    g = torch.Generator().manual_seed(2)
    N, Fr, C, H, W = 2, 3, 8, 10, 14
    hs = torch.randn(N, Fr, 2 * C, H, W, generator=g).to(BF)
    ehs = torch.randn(N, 7, 128, generator=g).to(BF)
    ts = torch.tensor([459, 459])
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr)
    ref = dit_oracle.dit_forward(ocfg, w, hs.float(), ehs.float(), ts, rope)
    out = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    check_floor("cog_forward_16heads_ragged", out, ref, eager_bf16(ocfg, model, hs, ehs, ts, rope), channel_dim=2)


def test_alg_sampler_vs_loop_oracle(device):
    """BASELINE config 1 in miniature: 9 frames, 2 denoise steps (one 3-pass + one 2-pass), ALG down_up in latent
    space, interval schedule -- HIP sampler vs the fp32 CPU loop oracle on identical seeds."""
    ocfg, w, model = make_pair(device, seed=5)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(42)
    Fr, C, H, W = 3, 8, 8, 12
    latents = torch.randn(1, Fr, C, H, W, generator=g).to(BF)
    first = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(BF)
    pe = torch.randn(1, 10, 128, generator=g).to(BF)
    ne = torch.randn(1, 10, 128, generator=g).to(BF)
    kw = dict(num_inference_steps=2, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.04)
    trace = []
    out = pipe(image=None, image_latents=first, latents=latents, prompt_embeds=pe, negative_prompt_embeds=ne,
               height=H * 8, width=W * 8, num_frames=9, output_type="latent", lp_filter_in_latent=True,
               step_trace=trace, **kw).frames
    assert [(tp, n) for _, tp, n in trace] == [(False, 3), (True, 2)]
    cond = torch.zeros(1, Fr, C, H, W)
    cond[:, :1] = first.float()
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr)
    tf = lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, w, x, e, ts, r)
    otrace = []
    ref = loop_oracle.alg_denoise_loop(tf, ddim_oracle.DDIMOracle(), latents.float(), cond, pe.float(), ne.float(),
                                       image_rotary_emb=rope, trace=otrace, **kw)
    assert [(tp, n) for _, tp, n in otrace] == [(tp, n) for _, tp, n in trace]   # branch flags bit-exact
    assert [s for s, _, _ in otrace] == [s for s, _, _ in trace]                  # schedule values bit-exact
    # the reference's own bf16 eager run of the same two steps sets the tolerance (no bare 4e-2)
    eager = loop_oracle.alg_denoise_loop(lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, model.w_bf16, x, e, ts, r),
                                         ddim_oracle.DDIMOracle(), latents, cond.to(BF), pe, ne, image_rotary_emb=rope, **kw)
    check_floor("cog_sampler_2steps", out, ref, eager, channel_dim=2)


def test_sampler_schedules_and_filter_cache(device):
    """linear-decay gaussian schedule (C3's ALG settings) on the small model: the filter is launched once per
    distinct strength; zero-strength steps reuse the sharp condition object."""
    _, _, model = make_pair(device, seed=6)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(7)
    first = (torch.randn(1, 1, 8, 8, 12, generator=g) * 0.7).to(BF)
    pe, ne = torch.randn(1, 10, 128, generator=g).to(BF), torch.randn(1, 10, 128, generator=g).to(BF)
    trace = []
    out = pipe(image_latents=first, prompt_embeds=pe, negative_prompt_embeds=ne, height=64, width=96, num_frames=9,
               num_inference_steps=8, output_type="latent", use_low_pass_guidance=True, lp_filter_in_latent=True,
               lp_filter_type="gaussian_blur", lp_blur_sigma=3.0, lp_blur_kernel_size=3,
               lp_strength_schedule_type="linear", generator=torch.Generator().manual_seed(0), step_trace=trace).frames
    assert torch.isfinite(out.float()).all()
    assert [n for _, _, n in trace] == [3, 3, 3, 3, 2, 2, 2, 2]
    assert len(pipe._lp_cache) == 5  # 4 distinct non-zero sigmas + the sigma == 0 identity
    with pytest.raises(NameError):  # reference quirk a-Q1
        pipe(image_latents=first, prompt_embeds=pe, negative_prompt_embeds=ne, height=64, width=96, num_frames=9,
             num_inference_steps=2, output_type="latent", use_low_pass_guidance=True, guidance_scale=1.0)
    with pytest.raises(alg_amd.AlgHipError):  # no VAE attached -> cannot decode
        pipe(image_latents=first, prompt_embeds=pe, negative_prompt_embeds=ne, height=64, width=96, num_frames=9,
             num_inference_steps=1)


class _ThreadPair:
    """Stands in for alg_amd.parallel.CFGPairSplit on ONE GPU: the two 'ranks' are threads with their own model
    instance (own workspace); merge()'s all-gather goes through a shared dict and a barrier."""

    def __init__(self):
        import threading
        self.barrier = threading.Barrier(2)
        self.box = {}

    def rank(self, pair_rank):
        from alg_amd.parallel import CFGPairSplit
        outer = self

        class _Rank(CFGPairSplit):
            def all_gather(self, parts, t):
                outer.box[self.pair_rank] = t.clone()
                outer.barrier.wait()
                for r in (0, 1):
                    parts[r].copy_(outer.box[r])
                outer.barrier.wait()
        return _Rank(group=None, pair_rank=pair_rank)


def test_cfg_pair_split_reproduces_the_single_gpu_sampler(device):
    """cond / uncond passes of one video on two 'ranks' (threads): both must end on the bits of the unsplit sampler --
    the DiT kernels are batch-consistent and the merged prediction is exact."""
    import threading
    g = torch.Generator().manual_seed(43)
    Fr, C, H, W = 3, 8, 8, 12
    latents = torch.randn(1, Fr, C, H, W, generator=g).to(BF)
    first = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(BF)
    pe, ne = torch.randn(1, 10, 128, generator=g).to(BF), torch.randn(1, 10, 128, generator=g).to(BF)
    kw = dict(image=None, image_latents=first, latents=latents, prompt_embeds=pe, negative_prompt_embeds=ne, height=H * 8,
              width=W * 8, num_frames=9, output_type="latent", lp_filter_in_latent=True, num_inference_steps=4,
              guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up", lp_resize_factor=0.25,
              lp_strength_schedule_type="interval", schedule_interval_start_time=0.0, schedule_interval_end_time=0.3)

    def pipe():
        _, _, model = make_pair(device, seed=5)
        return CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)

    trace = []
    ref = pipe()(step_trace=trace, **kw).frames
    assert sorted(set(n for _, _, n in trace)) == [2, 3]       # both the 3-pass and the 2-pass split are exercised
    pair, outs, errs = _ThreadPair(), {}, []

    def run(r):
        try:
            outs[r] = pipe()(cfg_split=pair.rank(r), **kw).frames
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            pair.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    assert torch.equal(outs[0], ref) and torch.equal(outs[1], ref)


def test_alg_sampler_with_dpm_scheduler(device):
    """cog:1111-1122: the CogVideoXDPMScheduler branch (two-output step, fresh noise each step) vs the loop oracle
    driving oracle.ddim_oracle.DPMOracle on the same CPU generator stream."""
    from alg_amd import CogVideoXDPMScheduler
    ocfg, w, model = make_pair(device, seed=5)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDPMScheduler()).to(device)
    g = torch.Generator().manual_seed(44)
    Fr, C, H, W = 3, 8, 8, 12
    # fp32-exact inputs in bf16 so both sides see the same values; latents are bf16 in the reference loop
    latents = torch.randn(1, Fr, C, H, W, generator=g).to(BF)
    first = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(BF)
    pe, ne = torch.randn(1, 10, 128, generator=g).to(BF), torch.randn(1, 10, 128, generator=g).to(BF)
    kw = dict(num_inference_steps=4, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.3)
    out = pipe(image=None, image_latents=first, latents=latents, prompt_embeds=pe, negative_prompt_embeds=ne,
               height=H * 8, width=W * 8, num_frames=9, output_type="latent", lp_filter_in_latent=True,
               generator=torch.Generator().manual_seed(7), **kw).frames
    cond = torch.zeros(1, Fr, C, H, W)
    cond[:, :1] = first.float()
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr)
    tf = lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, w, x, e, ts, r)
    # the oracle keeps bf16 latents like the reference loop (the noise is drawn in the latents' dtype)
    ref = loop_oracle.alg_denoise_loop(tf, ddim_oracle.DPMOracle(), latents, cond.to(BF), pe, ne,
                                       image_rotary_emb=rope, generator=torch.Generator().manual_seed(7), **kw)
    assert out.dtype == BF and torch.isfinite(out.float()).all()
    eager = loop_oracle.alg_denoise_loop(lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, model.w_bf16, x, e, ts, r),
                                         ddim_oracle.DPMOracle(), latents, cond.to(BF), pe, ne, image_rotary_emb=rope,
                                         generator=torch.Generator().manual_seed(7), **kw)
    check_floor("cog_sampler_dpm_4steps", out, ref.float(), eager, channel_dim=2)


def test_cogvideox_1_5_forward_matches_the_oracle(device):
    """CogVideoX 1.5 variant of row a-6 (`patch_size_t`, cog:380-382/553-582/998): Linear patch embed over (c, t, py, px),
    slice rotary grid, ofs embedding added to the timestep embedding, p_t-fold unpatchify."""
    ocfg, w, model = make_pair(device, overrides=dict(patch_size_t=2, ofs_embed_dim=64,
                                                      use_learned_positional_embeddings=False), seed=9)
    g = torch.Generator().manual_seed(3)
    N, Fr, C, H, W = 2, 4, 8, 8, 12
    x = torch.randn(N, Fr, 2 * C, H, W, generator=g).to(BF)
    e = torch.randn(N, 10, 128, generator=g).to(BF)
    t = torch.tensor([731.0, 731.0])
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr)
    assert rope[0].shape == (2 * 4 * 6, 64)
    ofs = torch.full((1,), 2.0)
    want = dit_oracle.dit_forward(ocfg, w, x.float(), e.float(), t, rope, ofs=ofs)
    got = model(x.to(device), e.to(device), t.to(device), ofs=ofs.to(device),
                image_rotary_emb=(rope[0].to(device), rope[1].to(device)), return_dict=False)[0]
    assert got.shape == want.shape == (N, Fr, C, H, W)
    check_floor("cog15_forward", got, want, eager_bf16(ocfg, model, x, e, t, rope, ofs=ofs), channel_dim=2)
    with pytest.raises(ValueError, match="ofs"):
        model(x.to(device), e.to(device), t.to(device), image_rotary_emb=(rope[0].to(device), rope[1].to(device)))
    with pytest.raises(ValueError, match="multiple of patch_size_t"):
        model(x[:, :3].contiguous().to(device), e.to(device), t.to(device), ofs=ofs.to(device))


def test_cogvideox_1_5_sampler_pads_frames_and_matches_the_loop_oracle(device):
    """cog:961-968: 9 frames -> 3 latent frames -> padded to 4 (13 frames); the padding frame is dropped on decode paths
    (cog:1144) and kept for output_type='latent'; loop vs oracle with the ofs embedding in the stand-in closure."""
    ocfg, w, model = make_pair(device, overrides=dict(patch_size_t=2, ofs_embed_dim=64,
                                                      use_learned_positional_embeddings=False), seed=10)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(5)
    C, H, W = 8, 8, 12
    first = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(BF)
    pe, ne = torch.randn(1, 10, 128, generator=g).to(BF), torch.randn(1, 10, 128, generator=g).to(BF)
    kw = dict(num_inference_steps=3, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.4)
    latents = torch.randn(1, 4, C, H, W, generator=g).to(BF)
    out = pipe(image=None, image_latents=first, latents=latents, prompt_embeds=pe, negative_prompt_embeds=ne,
               height=H * 8, width=W * 8, num_frames=9, output_type="latent", lp_filter_in_latent=True,
               generator=torch.Generator().manual_seed(7), **kw).frames
    assert out.shape == (1, 4, C, H, W)
    cond = torch.zeros(1, 4, C, H, W)
    cond[:, :1] = first.float()                      # [image, 0, 0] + one more zero frame: 3 % 2 == 1 -> 4 frames
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, 4)
    ofs = torch.full((1,), 2.0)
    tf = lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, w, x, e, ts, r, ofs=ofs)
    ref = loop_oracle.alg_denoise_loop(tf, ddim_oracle.DDIMOracle(), latents.float(), cond, pe.float(), ne.float(),
                                       image_rotary_emb=rope, **kw)
    eager = loop_oracle.alg_denoise_loop(lambda x, e, ts, r: dit_oracle.dit_forward(ocfg, model.w_bf16, x, e, ts, r, ofs=ofs),
                                         ddim_oracle.DDIMOracle(), latents, cond.to(BF), pe, ne, image_rotary_emb=rope, **kw)
    check_floor("cog15_sampler_3steps", out, ref.float(), eager, channel_dim=2)


def test_sampler_batch_of_two_prompts_equals_two_runs(device):
    """cog:929-937, 1060-1070: two prompts in one call -> CFG batches of 4 / 6 samples; each video must come out exactly as
    in its own call (same latents, embeddings and condition): rows of every kernel are independent of the batch."""
    ocfg, w, model = make_pair(device, seed=12)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(21)
    Fr, C, H, W = 3, 8, 8, 12
    latents = torch.randn(2, Fr, C, H, W, generator=g).to(BF)
    first = (torch.randn(2, 1, C, H, W, generator=g) * 0.7).to(BF)
    pe, ne = torch.randn(2, 10, 128, generator=g).to(BF), torch.randn(2, 10, 128, generator=g).to(BF)
    kw = dict(num_inference_steps=3, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.4, height=H * 8, width=W * 8, num_frames=9, output_type="latent",
              lp_filter_in_latent=True, image=None)
    trace = []
    both = pipe(image_latents=first, latents=latents, prompt_embeds=pe, negative_prompt_embeds=ne, step_trace=trace,
                **kw).frames
    assert both.shape == (2, Fr, C, H, W) and [n for _, _, n in trace] == [6, 4, 4]
    for b in range(2):
        one = pipe(image_latents=first[b:b + 1], latents=latents[b:b + 1], prompt_embeds=pe[b:b + 1],
                   negative_prompt_embeds=ne[b:b + 1], **kw).frames
        assert torch.equal(one[0], both[b]), b


@pytest.mark.parametrize("case", ["i2v_other_frame_count", "2b_style_sincos_only"])
def test_sincos_positional_embeddings_match_the_oracle(device, case):
    """CogVideoXPatchEmbed adds the LEARNED joint table only at the configured frame count; at any other count -- and for
    checkpoints with neither learned nor rotary embeddings (CogVideoX-2B style) -- the 3-D sincos embedding of the actual grid
    (VERDICT r3 missing 5: this used to raise).  HIP forward vs the fp32 oracle at the bf16-eager floor."""
    if case == "i2v_other_frame_count":
        over, Fr, use_rope = dict(), 5, True                  # configured for 9 frames (3 latent frames): run 17 frames (5)
    else:
        over, Fr, use_rope = dict(use_learned_positional_embeddings=False, use_rotary_positional_embeddings=False), 3, False
    ocfg, w, model = make_pair(device, over, seed=12)
    g = torch.Generator().manual_seed(2)
    N, C, H, W = 2, 8, 8, 12
    hs = torch.randn(N, Fr, 2 * C, H, W, generator=g).to(BF)
    ehs = torch.randn(N, 10, 128, generator=g).to(BF)
    ts = torch.tensor([500, 500])
    rope = dit_oracle.rope_tables(ocfg, H * 8, W * 8, Fr) if use_rope else None
    ref = dit_oracle.dit_forward(ocfg, w, hs.float(), ehs.float(), ts, rope)
    out = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    check_floor("cog_forward_sincos_" + case, out, ref, eager_bf16(ocfg, model, hs, ehs, ts, rope), channel_dim=2)
    # the embedding is live: dropping it changes the prediction
    model._sincos = {k: torch.zeros_like(v) for k, v in model._sincos.items()}
    assert len(model._sincos) == 1
    out0 = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    assert rel(out0, out) > 1e-3


def test_eta_is_accepted_and_changes_nothing(device):
    """cog:446-461 hands `eta` to scheduler.step; the published CogVideoX schedulers accept and ignore it: same latents."""
    ocfg, w, model = make_pair(device)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(4)
    kw = dict(image_latents=(torch.randn(1, 1, 8, 8, 12, generator=g) * 0.7).to(BF), latents=torch.randn(1, 3, 8, 8, 12, generator=g).to(BF),
              prompt_embeds=torch.randn(1, 10, 128, generator=g).to(BF), negative_prompt_embeds=torch.randn(1, 10, 128, generator=g).to(BF),
              height=64, width=96, num_frames=9, num_inference_steps=3, guidance_scale=6.0, output_type="latent")
    a = pipe(**kw).frames
    b = pipe(eta=0.7, **kw).frames
    assert torch.equal(a, b)
