"""Parity at BASELINE.json's full C2 sizes (CogVideoX-5B-I2V shapes: 17,776 tokens x 3072, 48 heads) through
size-independent properties and spot checks; torch's own GPU ops are used as an independent checker where the CPU oracle
would take minutes."""
import numpy as np
import pytest
import torch

from _parity import assert_repeatable
from alg_amd import (CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel,
                     CogVideoXTransformerConfig, _lib, lp_utils)
from alg_amd.pipeline_cogvideox_image2video_lowpass import get_resize_crop_region_for_grid, rotary_tables
from oracle import lp_oracle

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
S, D, H, T = 17776, 3072, 48, 226


def swap23(n):
    return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1)


def test_attention_full_sequence_rows(device):
    """S = 17,776 (278 KV tiles, ragged last tile of 48): rows from first/last query blocks and both half-waves vs fp32."""
    g = torch.Generator(device=device).manual_seed(1)
    nb, heads = 1, 8
    Dh = heads * 64
    S_pad = (S + 63) // 64 * 64
    qk = (torch.randn(nb, S, 2 * Dh, generator=g, device=device)).to(BF)
    v = torch.randn(nb, S, Dh, generator=g, device=device).to(BF)
    perm = torch.tensor([swap23(n) for n in range(S)], device=device)
    vt = torch.zeros(nb, Dh, S_pad, dtype=BF, device=device)
    vt[:, :, perm] = v.transpose(1, 2)
    o = torch.zeros(nb, S, Dh, dtype=BF, device=device)
    _lib.flash_attn_d64(qk, qk, vt, o, nb, heads, S, S * 2 * Dh, 2 * Dh, Dh * S_pad, S_pad, S * Dh, Dh, 0.125, k_off=Dh)
    rows = torch.tensor([0, 31, 32, 63, 255, 256, 8191, 17407, 17408, 17727, 17775], device=device)
    for hh in (0, 5, 7):
        q = qk[0, rows, hh * 64:(hh + 1) * 64].float()
        k = qk[0, :, Dh + hh * 64:Dh + (hh + 1) * 64].float()
        p = torch.softmax(q @ k.t() * 0.125, dim=-1)
        ref = p @ v[0, :, hh * 64:(hh + 1) * 64].float()
        got = o[0, rows, hh * 64:(hh + 1) * 64].float()
        # outputs are O(0.01) (a mean of 17,776 N(0,1) values), so the bound is RELATIVE to each row: bf16 rounding of P
        # and of the output contribute ~2^-9 each; 1e-2 of the row norm leaves a 2.5x margin and no more
        err = (got - ref).norm(dim=-1) / ref.norm(dim=-1)
        assert err.max().item() <= 1e-2, err
    # softmax rows are convex combinations: V == const  =>  O == const exactly up to bf16 rounding of P
    vt.fill_(0)
    vt[:, :, :S] = 1.5
    _lib.flash_attn_d64(qk, qk, vt, o, nb, heads, S, S * 2 * Dh, 2 * Dh, Dh * S_pad, S_pad, S * Dh, Dh, 0.125, k_off=Dh)
    assert (o.float() - 1.5).abs().max().item() <= 1.5 * 2 ** -7


def test_gemm_full_shapes_vs_torch(device):
    """The five GEMM shapes of a C2 layer (N = 1 sample): sampled output rows against torch's fp32 matmul."""
    g = torch.Generator(device=device).manual_seed(2)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=device) * sc).to(BF)
    y = rn(S, D)
    rows = torch.tensor([0, 1, 255, 256, 4095, 17663, 17664, 17775], device=device)
    for N, K, act in ((2 * D, D, _lib.ACT_NONE), (4 * D, D, _lib.ACT_GELU_TANH), (D, D, _lib.ACT_NONE)):
        w, b = rn(N, K, sc=0.02), rn(N, sc=0.1)
        out = torch.empty(S, N, dtype=BF, device=device)
        _lib.gemm(y, w, out, S, N, K, K, K, N, bias=b, act=act)
        lin = (y[rows].float() @ w.float().t() + b.float()).to(BF).float()
        ref = torch.nn.functional.gelu(lin, approximate="tanh") if act else lin
        assert (out[rows].float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    # FF2 form: K = 12288, gated residual in place
    h, w2, b2 = rn(S, 4 * D), rn(D, 4 * D, sc=0.02), rn(D, sc=0.1)
    x = rn(S, D)
    gate = rn(1, 2 * D, sc=0.5)
    x0 = x.clone()
    _lib.gemm(h, w2, x, S, D, 4 * D, 4 * D, 4 * D, D, bias=b2, R=x, ldr=D, gate=gate, strideGate=2 * D, seg_split=T)
    lin = (h[rows].float() @ w2.float().t() + b2.float()).to(BF).float()
    gsel = torch.where(rows[:, None] < T, gate[0, :D].float(), gate[0, D:].float())
    ref = (x0[rows].float() + (gsel * lin).to(BF).float()).to(BF).float()
    assert (x[rows].float() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    # transposed V projection at full S: pad columns stay zero, live columns match
    wv, bv = rn(D, D, sc=0.02), rn(D, sc=0.1)
    S_pad = (S + 63) // 64 * 64
    vt = torch.zeros(D, S_pad, dtype=BF, device=device)
    _lib.gemm(wv, y, vt, D, S, D, D, D, S_pad, bias=bv, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
    assert torch.count_nonzero(vt[:, S:]) == 0
    cols = torch.tensor([0, 4, 8, 12, 17775, 17771], device=device)
    ref = (y[cols].float() @ wv.float().t() + bv.float()).t()
    pos = torch.tensor([swap23(int(c)) for c in cols], device=device)
    assert (vt[:, pos].float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


def test_dit_full_width_two_layers_consistency(device, monkeypatch):
    """Full C2 token count and width, 2 of the 42 layers: deterministic, finite, and a sample's prediction does not
    depend on its position in the CFG batch or on the batch size (N = 2 vs N = 3): bit for bit with one attention
    schedule (ALG_ATTN_SPLIT_TAIL=0), and to bf16 rounding with the default one, whose split-KV tail sums the last
    few (head, q-block) units of a launch in a launch-shape dependent order."""
    cfg = CogVideoXTransformerConfig(num_layers=2)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=7, device=device)
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    c0 = torch.zeros(1, 13, 16, 60, 90, dtype=BF, device=device)
    c0[:, 0] = (torch.randn(1, 16, 60, 90, generator=g) * 0.7).to(device, BF)
    c1 = lp_utils.apply_low_pass_filter(c0, "down_up", 0.0, 0, 0.25)
    pe, ne = (torch.randn(1, 226, 4096, generator=g).to(device, BF) for _ in range(2))
    rope = rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), 13)
    ts3, ts2 = torch.full((3,), 999.0), torch.full((2,), 999.0)
    out3 = model.forward_assembled(lat, [c0, c1, c1], torch.cat([ne, ne, pe]), ts3, rope)
    out2 = model.forward_assembled(lat, [c1, c1], torch.cat([ne, pe]), ts2, rope)
    assert out3.shape == (3, 13, 16, 60, 90) and torch.isfinite(out3.float()).all()
    for a, b in ((out3[1], out2[0]), (out3[2], out2[1])):
        r = ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert r < 1e-2, r
    monkeypatch.setenv("ALG_ATTN_SPLIT_TAIL", "0")
    out3 = model.forward_assembled(lat, [c0, c1, c1], torch.cat([ne, ne, pe]), ts3, rope)
    out2 = model.forward_assembled(lat, [c1, c1], torch.cat([ne, pe]), ts2, rope)
    assert torch.equal(out3[1], out2[0]) and torch.equal(out3[2], out2[1])
    assert not torch.equal(out3[0], out3[1])  # the sharp and the low-passed condition give different predictions
    again = assert_repeatable(lambda: model.forward_assembled(lat, [c0, c1, c1], torch.cat([ne, ne, pe]), ts3, rope), 8,
                              "c2 3-pass forward (single-launch attention)")
    assert torch.equal(again, out3)
    monkeypatch.delenv("ALG_ATTN_SPLIT_TAIL")
    assert_repeatable(lambda: model.forward_assembled(lat, [c1, c1], torch.cat([ne, pe]), ts2, rope), 8,
                      "c2 2-pass forward (default split-KV tail)")


def test_sampler_full_latent_size_one_layer(device):
    """C2 sampler settings (49 frames @ 480x720, interval [0, 0.04], down_up 0.25) on a 1-layer DiT, 50 steps:
    102 sample-forwards, 2 filter launches (the sharp condition object is reused at strength 0), finite result."""
    cfg = CogVideoXTransformerConfig(num_layers=1)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=9, device=device)
    pipe = CogVideoXImageToVideoPipeline(transformer=model, scheduler=CogVideoXDDIMScheduler()).to(device)
    g = torch.Generator().manual_seed(42)
    first = (torch.randn(1, 1, 16, 60, 90, generator=g) * 0.7).to(BF)
    pe, ne = (torch.randn(1, 226, 4096, generator=g).to(BF) for _ in range(2))
    trace = []
    out = pipe(image_latents=first, prompt_embeds=pe, negative_prompt_embeds=ne, num_frames=49, num_inference_steps=50,
               guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up", lp_filter_in_latent=True,
               lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
               schedule_interval_end_time=0.04, output_type="latent", generator=torch.Generator().manual_seed(42),
               step_trace=trace).frames
    assert out.shape == (1, 13, 16, 60, 90) and torch.isfinite(out.float()).all()
    assert sum(n for _, _, n in trace) == 102 and [n for _, _, n in trace[:3]] == [3, 3, 2]
    assert [s for s, _, _ in trace[:3]] == [1.0, 1.0, 0.0]
    assert len(pipe._lp_cache) == 2  # factor 0.25 (filtered once) and factor 1.0 (identity: the input object itself)


def test_wan_self_attention_full_size_properties():
    """Wan-480p self-attention shape (40 heads x 32,760 tokens x 128): size-independent properties -- softmax rows sum to
    one (V = 1 -> O = 1) and with K = 0 every query returns the mean of V."""
    from alg_amd import _lib
    S, H = 32760, 4                      # 4 of the 40 heads: same kernel path, 10x less memory
    D = H * 128
    s_pad = (S + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(1, S, D, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, S, D, generator=g, device="cuda").to(torch.bfloat16)
    vt = torch.zeros(1, D, s_pad, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :S] = 1.0
    o = torch.empty(1, S, D, dtype=torch.bfloat16, device="cuda")
    args = (1, H, S, S, S * D, D, S * D, D, D * s_pad, s_pad, S * D, D, 128 ** -0.5)
    _lib.flash_attn_d128(q, k, vt, o, *args)
    assert (o.float() - 1.0).abs().max().item() <= 2.0 ** -7
    v = torch.randn(1, D, S, generator=g, device="cuda").to(torch.bfloat16)
    vt[:, :, :S] = v
    _lib.flash_attn_d128(q, torch.zeros_like(k), vt, o, *args)
    mean = v.float().mean(dim=2)          # the kv permutation inside 16-blocks does not change a mean
    assert (o.float() - mean[:, None, :]).abs().max().item() <= 3e-3


def test_vae_decode_full_size_properties(device):
    """C2's decode (13 x 60 x 90 latents -> 49 x 480 x 720) through size-independent properties of the published batched
    decoder: later latent batches cannot change earlier frames (causal convolutions, per-batch GroupNorm: frames < 9 depend
    on latent frames 0-2 only), run-to-run bit identity, and the fused uint8 writer == postprocess of the bf16 frames."""
    from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    from oracle import vae_oracle
    vae = AutoencoderKLCogVideoX.from_synthetic(device=device)
    g = torch.Generator(device=device).manual_seed(2)
    lat = torch.randn(1, 13, 16, 60, 90, generator=g, device=device).to(BF)
    a = vae.decode_latents(lat)
    assert a.shape == (1, 3, 49, 480, 720) and bool(torch.isfinite(a.float()).all())
    assert torch.equal(vae.decode_latents(lat), a)
    lat2 = lat.clone()
    lat2[:, 5:] += 0.5                                   # latent batches: [0-2], [3-4], [5-6], ... -> frames 17.. change
    b = vae.decode_latents(lat2)
    assert torch.equal(b[:, :, :17], a[:, :, :17]) and not torch.equal(b[:, :, 17:25], a[:, :, 17:25])
    u8 = vae.decode_latents(lat, to_uint8=True)
    sl = slice(0, 49, 12)
    assert torch.equal(u8[0, sl].cpu(), vae_oracle.postprocess_uint8(a[0][:, sl].cpu()))


def test_text_and_image_encoders_full_size_properties(device):
    """T5 v1.1 XXL at the CogVideoX prompt length (226, no mask), UMT5-width at Wan's 512 with a mask, ViT-H/14 at 257
    tokens: finite, unit RMS after the final T5LayerNorm (synthetic gains ~ 1), padded tokens never reach valid rows."""
    from alg_amd import CLIPVisionModel, T5EncoderConfig, T5EncoderModel, UMT5EncoderModel
    t5 = T5EncoderModel.from_synthetic(T5EncoderConfig(num_layers=2), seed=1, device=device)   # XXL widths, 2 of 24 blocks
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 32128, (1, 226), generator=g).to(device)
    out = t5(ids)[0]
    assert out.shape == (1, 226, 4096) and bool(torch.isfinite(out.float()).all())
    rms = out.float().pow(2).mean(-1).sqrt()
    assert 0.9 < rms.mean().item() < 1.1
    um = UMT5EncoderModel.from_synthetic(T5EncoderConfig(num_layers=2, vocab_size=4096), seed=2, device=device)
    ids = torch.randint(0, 4096, (2, 512), generator=g)
    mask = torch.ones(2, 512, dtype=torch.long)
    mask[0, 40:] = 0
    mask[1, 300:] = 0
    x = um(ids.to(device), mask.to(device)).last_hidden_state
    ids2 = ids.clone()
    ids2[mask == 0] = 5
    y = um(ids2.to(device), mask.to(device)).last_hidden_state
    keep = mask.bool().to(device)
    assert torch.equal(x[keep], y[keep]) and bool(torch.isfinite(x.float()).all())
    from alg_amd.image_encoder_clip import CLIPVisionEncoderConfig
    clip = CLIPVisionModel.from_synthetic(CLIPVisionEncoderConfig(num_hidden_layers=3), seed=3, device=device)
    px = torch.randn(1, 3, 224, 224, generator=g).to(device)
    hs = clip(pixel_values=px, output_hidden_states=True).hidden_states
    assert len(hs) == 4 and hs[-2].shape == (1, 257, 1280) and bool(torch.isfinite(hs[-2].float()).all())


def test_attention_split_kv_tail_matches_the_unsplit_kernel(device, monkeypatch):
    """2-sample C2 launch: 840 (head, q-block) units per XCD = 13.125 rounds of 64 slots, so the last 8 units per XCD are
    cut into 8 KV chunks + a merge.  Same softmax, different fp32 summation order: rows of the tail units agree with the
    unsplit kernel to bf16 rounding, all other rows are bit-identical."""
    g = torch.Generator(device=device).manual_seed(5)
    N, heads = 2, 48
    Dh = heads * 64
    S_pad = (S + 127) // 128 * 128
    qk = torch.randn(N, S, 2 * Dh, generator=g, device=device).to(BF)
    vt = torch.randn(N, Dh, S_pad, generator=g, device=device).to(BF)
    vt[:, :, S:] = 0
    run = lambda: _lib.flash_attn_d64(qk, qk, vt, torch.empty(N, S, Dh, dtype=BF, device=device), N, heads, S,
                                      S * 2 * Dh, 2 * Dh, Dh * S_pad, S_pad, S * Dh, Dh, 0.125, k_off=Dh)

    def call():
        out = torch.empty(N, S, Dh, dtype=BF, device=device)
        _lib.flash_attn_d64(qk, qk, vt, out, N, heads, S, S * 2 * Dh, 2 * Dh, Dh * S_pad, S_pad, S * Dh, Dh, 0.125, k_off=Dh)
        return out

    monkeypatch.setenv("ALG_ATTN_SPLIT_TAIL", "0")
    ref = call()
    monkeypatch.delenv("ALG_ATTN_SPLIT_TAIL")
    got = call()
    assert bool(torch.isfinite(got.float()).all())
    same = (got == ref).reshape(N, S, heads, 64).all(dim=3)                 # [N, S, heads]
    # tail units: per XCD the last 8 of 840 units = q blocks 62..69 of the last (batch, head) slot of that XCD
    assert bool(same[0].all()) and bool(same[1, :, :40].all()) and bool(same[1, :62 * 256].all())
    tail = (got.float() - ref.float())[1, 62 * 256:, 40:]
    scale = ref.float()[1, 62 * 256:, 40:].abs().max().item()
    assert tail.abs().max().item() <= 2.0 ** -7 * scale and not bool(same[1, 62 * 256:, 40:].all())
    assert torch.equal(call(), got)                                          # deterministic


def test_default_attention_agrees_with_the_exact_max_variant_in_a_full_width_forward(device, monkeypatch):
    """The default attention keeps a lazy running max and sums rounded probabilities (variant 33); variant 1 subtracts the
    exact running max and sums fp32 probabilities.  Same softmax: a 2-layer forward at the full C2 token count and width
    must agree to bf16 rounding."""
    cfg = CogVideoXTransformerConfig(num_layers=2)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=11, device=device)
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    c0 = torch.zeros(1, 13, 16, 60, 90, dtype=BF, device=device)
    c0[:, 0] = (torch.randn(1, 16, 60, 90, generator=g) * 0.7).to(device, BF)
    pe, ne = (torch.randn(1, 226, 4096, generator=g).to(device, BF) for _ in range(2))
    rope = rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), 13)
    ts = torch.full((2,), 500.0)
    a = model.forward_assembled(lat, [c0, c0], torch.cat([ne, pe]), ts, rope).float()
    monkeypatch.setenv("ALG_ATTN_VARIANT", "1")
    b = model.forward_assembled(lat, [c0, c0], torch.cat([ne, pe]), ts, rope).float()
    rel = ((a - b).norm() / b.norm()).item()
    assert bool(torch.isfinite(a).all()) and rel < 5e-3, rel


def test_prescaled_attention_path_agrees_in_a_full_width_forward(device, monkeypatch):
    """The transformer folds the softmax scale into Q (alg_qk_norm_rope_scaled + ALG_ATTN_Q_PRESCALED, default);
    `model.attn_prescale = False` keeps the per-score multiply: same forward to bf16 rounding at the full C2 size (split-KV tail
    included on both paths)."""
    cfg = CogVideoXTransformerConfig(num_layers=2)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=13, device=device)
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    c0 = torch.zeros(1, 13, 16, 60, 90, dtype=BF, device=device)
    c0[:, 0] = (torch.randn(1, 16, 60, 90, generator=g) * 0.7).to(device, BF)
    pe, ne = (torch.randn(1, 226, 4096, generator=g).to(device, BF) for _ in range(2))
    rope = rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), 13)
    ts = torch.full((2,), 500.0)
    a = model.forward_assembled(lat, [c0, c0], torch.cat([ne, pe]), ts, rope).float()
    model.attn_prescale = False
    b = model.forward_assembled(lat, [c0, c0], torch.cat([ne, pe]), ts, rope).float()
    rel = ((a - b).norm() / b.norm()).item()
    assert bool(torch.isfinite(a).all()) and 0 < rel < 5e-3, rel


def test_c2_forward_at_its_real_shape_two_layers_vs_fp32_oracle(device):
    """VERDICT r3 missing 2: the HEADLINE configuration's forward at its real shape -- 48 heads x 64, 17,550 video + 226 text =
    17,776 tokens, N = 2 (a CFG step), 2 of the 42 layers -- HIP against `oracle/dit_oracle.dit_forward` in fp32 on the host
    (the mathematical reference, ~2 x 2 x 7.9e12 FLOP).  The floor is the reference's OWN execution mode at this shape: the same
    oracle function with bf16 weights / activations run by torch's eager ops on the device (what `run.py:65-69` executes on a
    GPU).  Bounds: tests/_parity.py -- global relative L2 <= 1.5 x floor, every token <= 4 x the floor's p99.9 token error,
    worst element <= 2 x the floor's worst element."""
    from _parity import check_floor
    from oracle import dit_oracle
    kw = dict(num_attention_heads=48, attention_head_dim=64, in_channels=32, out_channels=16, num_layers=2,
              time_embed_dim=512, text_embed_dim=4096, max_text_seq_length=226, sample_width=90, sample_height=60,
              sample_frames=49, patch_size=2)
    ocfg = dit_oracle.DiTConfig(**kw)
    wbf = {k: v.to(BF) for k, v in dit_oracle.init_weights(ocfg, seed=21, std=0.02, randomize_affine=True).items()}
    w32 = {k: v.float() for k, v in wbf.items()}
    model = CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**kw), wbf, device=device)
    g = torch.Generator().manual_seed(8)
    hs = torch.randn(2, 13, 32, 60, 90, generator=g).to(BF)
    hs[:, 1:, 16:] = 0                                       # the conditioning half: frame 0 real, frames 1..12 zero (cog:402-411)
    ehs = torch.randn(2, 226, 4096, generator=g).to(BF)
    ts = torch.tensor([999, 999])
    rope = dit_oracle.rope_tables(ocfg, 480, 720, 13)
    assert rope[0].shape == (17550, 64)
    out = model(hs.to(device), ehs.to(device), ts, image_rotary_emb=rope, return_dict=False)[0]
    assert out.shape == (2, 13, 16, 60, 90)
    wdev = {k: v.to(device) for k, v in wbf.items()}
    eager = dit_oracle.dit_forward(ocfg, wdev, hs.to(device), ehs.to(device), ts.to(device),
                                   tuple(t.to(device) for t in rope)).cpu()
    del wdev
    ref = dit_oracle.dit_forward(ocfg, w32, hs.float(), ehs.float(), ts, rope)
    check_floor("cog_forward_c2_real_shape_2layers_17776tokens", out, ref, eager, channel_dim=2)


def test_paired_qkv_launch_leaves_the_forward_bit_identical(device):
    """`pair_qkv` (default): the Q|K and V^T projections of every block as one alg_gemm_bf16_pair launch.  A tile is computed
    exactly as by its own launch, so the full-width 2-layer C2 forward must not change by a bit (N = 2 and N = 3); neither
    does `fuse_qk_norm` (opt-in: QK LayerNorm + rope inside the Q|K store loop, alg_gemm_bf16_pair_qk) change one."""
    cfg = CogVideoXTransformerConfig(num_layers=2)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=17, device=device)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    c0 = torch.zeros(1, 13, 16, 60, 90, dtype=BF, device=device)
    c0[:, 0] = (torch.randn(1, 16, 60, 90, generator=g) * 0.7).to(device, BF)
    pe, ne = (torch.randn(1, 226, 4096, generator=g).to(device, BF) for _ in range(2))
    rope = rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), 13)
    for n in (2, 3):
        ehs = torch.cat([ne] * (n - 1) + [pe])
        ts = torch.full((n,), 700.0)
        assert model.pair_qkv and not model.fuse_qk_norm
        a = model.forward_assembled(lat, [c0] * n, ehs, ts, rope)
        model.fuse_qk_norm = True            # QK LayerNorm + rope inside the Q|K store loop instead of the stand-alone kernel
        c = model.forward_assembled(lat, [c0] * n, ehs, ts, rope)
        model.pair_qkv, model.fuse_qk_norm = False, False
        b = model.forward_assembled(lat, [c0] * n, ehs, ts, rope)
        model.pair_qkv = True
        assert bool(torch.isfinite(a.float()).all()) and torch.equal(a, b) and torch.equal(a, c)
