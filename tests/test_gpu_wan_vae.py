"""GPU parity of the Wan 2.1 VAE (alg_amd/autoencoder_kl_wan.py; reference call sites wan:426-430 encode, wan:959 decode,
wan:493-540 the pixel-space ALG branch) against oracle/wan_vae_oracle.py -- the published CHUNKED algorithm in fp32, with the
same oracle run in bf16 eager mode as the tolerance floor (tests/_parity.py).  Parity unpinned: diffusers is absent."""
import pytest
import torch

from alg_amd import _lib
from alg_amd.autoencoder_kl_wan import AutoencoderKLWan, AutoencoderKLWanConfig
from oracle import wan_vae_oracle as O
from _parity import check_floor, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def make(base_dim, z_dim=4, seed=1):
    kw = dict(base_dim=base_dim, z_dim=z_dim, latents_mean=[0.1 * i for i in range(z_dim)],
              latents_std=[1.0 + 0.2 * i for i in range(z_dim)])
    ocfg = O.WanVAEConfig(**kw)
    sd = O.init_weights(ocfg, seed=seed)                                   # bf16 values
    vae = AutoencoderKLWan(AutoencoderKLWanConfig(**kw), device=DEV).load_state_dict(sd)
    return ocfg, sd, {k: v.float() for k, v in sd.items()}, vae


def test_rms_norm_rows_and_softmax_hilo():
    g = torch.Generator().manual_seed(0)
    rows, C, Cp = 37, 96, 128
    x = torch.randn(rows, Cp, generator=g).to(BF)
    gamma = torch.zeros(Cp)
    gamma[:C] = 1 + 0.1 * torch.randn(C, generator=g)
    gamma = gamma.to(BF)
    for silu in (False, True):
        y = torch.empty(rows, Cp, dtype=BF, device=DEV)
        _lib.rms_norm_rows(x.to(DEV), gamma.to(DEV), y, rows, C, Cp, silu)
        xf = x.float()[:, :C]
        ref = torch.nn.functional.normalize(xf, dim=1) * C ** 0.5 * gamma.float()[:C]
        ref = torch.nn.functional.silu(ref) if silu else ref
        assert (y[:, :C].float().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
        assert torch.count_nonzero(y[:, C:]) == 0
    # scores split in hi + lo bf16 parts; ragged column count, zero padding
    n, cols, ld = 19, 203, 256
    s = torch.randn(n, cols, generator=g) * 6
    hi = s.to(BF)
    lo = (s - hi.float()).to(BF)
    neg_hi, lo_p = torch.full((n, ld), 7.0, dtype=BF), torch.full((n, ld), -3.0, dtype=BF)   # junk in the padding columns
    neg_hi[:, :cols], lo_p[:, :cols] = -hi, lo
    p = torch.empty(n, ld, dtype=BF, device=DEV)
    _lib.softmax_hilo(neg_hi.to(DEV), lo_p.to(DEV), p, n, cols, ld, 0.37)
    ref = torch.softmax((hi.float() + lo.float()) * 0.37, dim=1)
    assert (p[:, :cols].float().cpu() - ref).abs().max().item() <= 2.0 ** -8
    assert torch.count_nonzero(p[:, cols:]) == 0
    assert (p.float().sum(1).cpu() - 1).abs().max().item() <= 2e-2


@pytest.mark.parametrize("base_dim,L,h,w", [(32, 3, 4, 6), (24, 2, 6, 4), (24, 1, 4, 4), (32, 5, 2, 4)])
def test_decode_whole_video_vs_chunked_oracle(base_dim, L, h, w):
    """wan:959: latents -> frames.  base_dim 24 gives 24 / 48 / 96 channels (padded to 64 / 64 / 128, like 96 / 192 / 384 ->
    128 / 256 / 512 at full size); 1, 2, 3, 5 latent frames cover the 'Rep' first chunk and both upsample3d levels."""
    ocfg, sd, sd32, vae = make(base_dim)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1, 4, L, h, w, generator=g).to(BF)
    ref = O.decode(ocfg, sd32, z.float())
    eager = O.decode(ocfg, sd, z)
    out = vae.decode(z.to(DEV)).sample
    assert out.shape == ref.shape == (1, 3, 4 * (L - 1) + 1, 8 * h, 8 * w) and out.dtype == BF
    check_floor("wan_vae_decode_dim%d_L%d" % (base_dim, L), out, ref, eager)
    assert torch.equal(vae.decode(z.to(DEV)).sample, out)                      # deterministic


@pytest.mark.parametrize("base_dim,T,H,W", [(32, 9, 32, 48), (24, 5, 48, 32), (24, 1, 32, 32), (32, 17, 16, 32)])
def test_encode_whole_video_vs_chunked_oracle(base_dim, T, H, W):
    """wan:426-430: the condition video (first frame + zeros, as the pipeline builds it) -> posterior moments."""
    ocfg, sd, sd32, vae = make(base_dim, seed=2)
    g = torch.Generator().manual_seed(4)
    x = torch.zeros(1, 3, T, H, W)
    x[:, :, 0] = torch.randn(1, 3, H, W, generator=g).clamp(-1, 1)
    if T > 5:
        x[:, :, 5:] = 0.3 * torch.randn(1, 3, T - 5, H, W, generator=g)        # and a non-trivial tail
    x = x.to(BF)
    ref = O.encode(ocfg, sd32, x.float())
    eager = O.encode(ocfg, sd, x)
    dist = vae.encode(x.to(DEV)).latent_dist
    mom = dist.parameters
    assert mom.shape == ref.shape == (1, 8, 1 + (T - 1) // 4, H // 8, W // 8)
    check_floor("wan_vae_encode_dim%d_T%d" % (base_dim, T), mom, ref, eager)
    assert torch.equal(dist.mode(), mom[:, :4])
    s1 = dist.sample(generator=torch.Generator().manual_seed(1))
    # the posterior lives in float32 like the reference's float32 Wan VAE (run:51-55): the draw is the float32 stream
    assert mom.dtype == torch.float32 and s1.dtype == torch.float32
    noise = torch.randn(dist.mean.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float32)
    assert torch.equal(s1.cpu(), (dist.mean.cpu() + dist.std.cpu() * noise))


def test_mid_block_attention_matches_fp32_scores():
    """One frame, more tokens than one KV tile (h * w = 15 x 9 = 135 -> padded to 192): the two-GEMM hi + lo scores and the
    fp32 softmax against the oracle's attention block alone."""
    ocfg, sd, sd32, vae = make(32, seed=5)
    C, T, H, W = 128, 2, 15, 9
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, C, T, H, W, generator=g).to(BF)
    name = "decoder.mid_block.attentions.0"
    ref = O.attention_block(x.float(), sd32, name)
    eager = O.attention_block(x, sd, name)
    from alg_amd.autoencoder_kl_wan import _Act
    a = _Act(torch.zeros(T * (H + 2) * (W + 2) * C, dtype=BF, device=DEV), T, H, W, C)
    a.valid().copy_(x[0].permute(1, 2, 3, 0).to(DEV))
    y = vae._attention(a, name).valid().permute(3, 0, 1, 2)[None]
    check_floor("wan_vae_mid_attention", y, ref, eager)


def test_wan_pipeline_image_in_frames_out_and_pixel_space_alg():
    """wan:587-970 end to end on tiny models: PIL-free tensor image -> encode_condition -> ALG loop -> decode -> frames; then
    the pixel-space branch (lp_filter_in_latent=False: filter RGB, re-encode and SAMPLE every step)."""
    from alg_amd import UniPCMultistepScheduler, WanImageToVideoPipeline, WanTransformer3DModel, WanTransformerConfig
    from oracle import wan_oracle
    kw = dict(num_attention_heads=4, ffn_dim=1024, num_layers=1, text_dim=64, image_dim=64, added_kv_proj_dim=512)
    model = WanTransformer3DModel(WanTransformerConfig(**kw), wan_oracle.init_weights(wan_oracle.WanConfig(**kw), seed=7), device=DEV)
    vcfg = AutoencoderKLWanConfig(base_dim=24, z_dim=16)
    vae = AutoencoderKLWan.from_synthetic(vcfg, seed=3, device=DEV)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(1, 3, 64, 96, generator=g).clamp(-1, 1)
    clip_tokens = torch.randn(1, 257, 64, generator=g).to(BF).to(DEV)

    class Proc:                                   # duck-typed CLIP processor / vision tower (wan:228-234 call protocol):
        def __call__(self, images, return_tensors):   # the reference forbids `image` next to `image_embeds`
            class Batch(dict):
                def to(self, device):
                    return self
            return Batch(pixel_values=images)

    class Enc:
        def __call__(self, pixel_values, output_hidden_states):
            from types import SimpleNamespace
            return SimpleNamespace(hidden_states=[None, clip_tokens, None])

    pipe = WanImageToVideoPipeline(transformer=model, vae=vae, image_encoder=Enc(), image_processor=Proc(),
                                   scheduler=UniPCMultistepScheduler(flow_shift=3.0)).to(DEV)
    emb = dict(prompt_embeds=torch.randn(1, 512, 64, generator=g).to(BF).to(DEV),
               negative_prompt_embeds=torch.randn(1, 512, 64, generator=g).to(BF).to(DEV))
    alg = dict(use_low_pass_guidance=True, lp_filter_type="down_up", lp_resize_factor=0.5, lp_strength_schedule_type="interval",
               schedule_interval_start_time=0.0, schedule_interval_end_time=0.5)
    trace = []
    out = pipe(image=img, height=64, width=96, num_frames=9, num_inference_steps=3, guidance_scale=5.0, output_type="pt",
               lp_filter_in_latent=True, generator=torch.Generator().manual_seed(1), step_trace=trace, **emb, **alg).frames
    assert out.shape == (1, 9, 3, 64, 96) and bool(torch.isfinite(out.float()).all())
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0 and float(out.float().std()) > 0.01
    assert [n for _, n, _ in trace] == [3, 3, 2]
    # the condition the pipeline encoded = [mask4 | normalised posterior mode] of the zero-padded condition video
    cond = pipe.encode_condition(img.to(DEV), 1, 64, 96, 9, torch.float32, torch.device(DEV))
    assert cond.shape == (1, 20, 3, 8, 12) and bool((cond[:, :4, 0] == 1).all()) and bool((cond[:, :4, 1:] == 0).all())
    video = torch.cat([img[:, :, None], torch.zeros(1, 3, 8, 64, 96)], dim=2).to(BF).to(DEV)
    mode = vae.encode(video).latent_dist.mode().float()
    mean = torch.tensor(vcfg.latents_mean, device=DEV).view(1, 16, 1, 1, 1)
    std = torch.tensor(vcfg.latents_std, device=DEV).view(1, 16, 1, 1, 1)
    assert torch.allclose(cond[:, 4:], (mode - mean) * (1.0 / std), atol=1e-6)
    # pixel-space ALG: runs, consumes the generator every active step, differs from the latent-space branch
    g1 = torch.Generator().manual_seed(1)
    pix = pipe(image=img, height=64, width=96, num_frames=9, num_inference_steps=3, guidance_scale=5.0, output_type="latent",
               lp_filter_in_latent=False, generator=g1, **emb, **alg).frames
    lat = pipe(image=img, height=64, width=96, num_frames=9, num_inference_steps=3, guidance_scale=5.0, output_type="latent",
               lp_filter_in_latent=True, generator=torch.Generator().manual_seed(1), **emb, **alg).frames
    assert pix.shape == lat.shape == (1, 16, 3, 8, 12) and bool(torch.isfinite(pix).all())
    assert rel(pix, lat) > 1e-4
    with pytest.raises(_lib.AlgHipError, match="VAE"):
        WanImageToVideoPipeline(transformer=model, image_encoder=Enc(), image_processor=Proc(),
                                scheduler=UniPCMultistepScheduler()).to(DEV)(
            image=img, height=64, width=96, num_frames=9, num_inference_steps=1, **emb)
