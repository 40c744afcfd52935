"""GPU parity of the CogVideoX VAE decoder row (call site cog:427-433): every kernel against a plain
torch fp32 statement of the same op, and the whole HIP decoder (whole-video formulation) against oracle/vae_oracle.py
(the published batched decode with conv caches).  bf16 arithmetic: tolerances are written next to each check."""
import pytest
import torch
import torch.nn.functional as F

from alg_amd import _lib
from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX, AutoencoderKLCogVideoXConfig, _Level
from oracle import vae_oracle

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _padded(x, time_pad):
    """NCTHW fp32 (B = 1) -> padded channels-last bf16 flat buffer [T + time_pad][H + 2][W + 2][C] (+ slack)."""
    _, C, T, H, W = x.shape
    if time_pad:
        x = torch.cat([x[:, :, :1]] * time_pad + [x], dim=2)
    x = F.pad(x, (1, 1, 1, 1))
    flat = x[0].permute(1, 2, 3, 0).contiguous().bfloat16().reshape(-1)
    return torch.cat([flat, torch.zeros((2 * (W + 2) + 4) * C, dtype=torch.bfloat16)]).to(_dev())


def _virtual(x, fill=7.0):
    """NCTHW (B = 1) -> virtual layout [T][H + 2][W + 2][C] with a sentinel in the don't-care rows."""
    x = F.pad(x, (0, 2, 0, 2), value=fill)
    return x[0].permute(1, 2, 3, 0).contiguous().bfloat16().reshape(-1).to(_dev())


def _from_virtual(buf, T, H, W, C):
    return buf.reshape(T, H + 2, W + 2, C)[:, :H, :W].permute(3, 0, 1, 2).float().cpu()


@pytest.mark.parametrize("Cin,Cout,kt,res,pair", [
    (64, 128, 3, False, False), (128, 128, 3, True, False), (256, 128, 3, False, False), (512, 256, 3, True, False),
    (128, 4, 3, False, False), (256, 256, 1, False, False), (128, 128, 3, True, True), (256, 128, 3, False, True),
    (128, 128, 1, False, True), (128, 4, 3, False, True)])
def test_conv_cl_matches_conv3d(Cin, Cout, kt, res, pair):
    g = torch.Generator().manual_seed(Cin + Cout + kt)
    T, H, W = 5, 8 if pair else 9, 13
    x = torch.randn(1, Cin, T, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, kt, 3, 3, generator=g) / (Cin * kt * 9) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g).bfloat16().float()
    r = torch.randn(1, Cout, T, H, W, generator=g).bfloat16().float() if res else None
    xin = torch.cat([x[:, :, :1]] * (kt - 1) + [x], dim=2)
    want = F.conv3d(xin, w, b, padding=(0, 1, 1))
    if res:
        want = want + r
    wp = w.reshape(Cout, Cin, -1).permute(0, 2, 1).reshape(Cout, -1).contiguous().bfloat16().to(_dev())
    bp = b.bfloat16().to(_dev())
    if pair:
        wp, bp = _lib.pack_conv_pair(wp, bp, kt)
    y = _virtual(r) if res else torch.full((T * (H + 2) * (W + 2) * Cout,), 3.0, dtype=torch.bfloat16, device=_dev())
    _lib.conv_cl(_padded(x, kt - 1), wp, bp, y if res else None, y, T, H + 2, W + 2, Cin, Cout, kt, pair=pair)
    got = _from_virtual(y, T, H, W, Cout)
    # fp32 accumulation, one bf16 rounding of the result: 2^-8 relative of the magnitude
    assert (got - want[0]).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()


def test_conv_cl_rejects_bad_shapes():
    z = torch.zeros(1 << 16, dtype=torch.bfloat16, device=_dev())
    with pytest.raises(_lib.AlgHipError, match="power of two"):
        _lib.conv_cl(z, z, None, None, z, 1, 5, 5, 96, 128, 3)
    with pytest.raises(_lib.AlgHipError, match="Cout"):
        _lib.conv_cl(z, z, None, None, z, 1, 5, 5, 64, 3, 3)


@pytest.mark.parametrize("C", [128, 256, 512])
@pytest.mark.parametrize("L,rate", [(5, 1), (5, 4), (4, 2), (1, 1)])
def test_groupnorm_stats_and_spatial_norm(C, L, rate):
    g = torch.Generator().manual_seed(C + L + rate)
    h, w, scale = 3, 5, 2
    lv = _Level(L, h, w, rate, scale)
    T, H, W = lv.T, lv.H, lv.W
    x = (torch.randn(1, C, T, H, W, generator=g) * 2 + 0.5).bfloat16().float()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).bfloat16()
    beta = (0.2 * torch.randn(C, generator=g)).bfloat16()
    zy = torch.randn(1, C, L, h, w, generator=g).bfloat16().float()
    zb = torch.randn(1, C, L, h, w, generator=g).bfloat16().float()
    geom = _lib.vae_geom(frames=T, H=H, W=W, C=C, first_len=lv.first_len, seg_len=lv.seg_len,
                         lat_first_single=int(lv.single), lat_rate=rate, lat_scale=scale, lat_h=h, lat_w=w)
    # segments and latent-frame map as the published batched decode produces them
    first_lat = L if L < 2 else 2 + L % 2
    lat_batches = [list(range(first_lat))] + [[i, i + 1] for i in range(first_lat, L, 2)]
    frame_of = lambda l: [0] if (lv.single and l == 0) else (
        list(range(1 + (l - 1) * rate, 1 + l * rate)) if lv.single else list(range(l * rate, (l + 1) * rate)))
    want = torch.zeros(1, C, T, H, W)
    stats_want = []
    for bt in lat_batches:
        fr = [f for l in bt for f in frame_of(l)]
        seg = x[:, :, fr]
        n = F.group_norm(seg, 32, gamma.float(), beta.float(), 1e-6).bfloat16().float()
        lat_idx = torch.tensor([l for l in bt for _ in frame_of(l)])
        up = lambda z: z[:, :, lat_idx].repeat_interleave(scale, 3).repeat_interleave(scale, 4)
        a = ((n * up(zy)).bfloat16().float() + up(zb)).bfloat16().float()
        want[:, :, fr] = F.silu(a).bfloat16().float()
        sg = seg[0].reshape(32, -1)                                  # group = C/32 consecutive channels, all voxels
        stats_want.append(torch.stack([sg.mean(1), (sg.var(1, unbiased=False) + 1e-6).rsqrt()], 1))
    ws = torch.empty(_lib.vae_groupnorm_workspace(geom) // 4, device=_dev())
    stats = torch.empty(len(lat_batches) * 64, device=_dev())
    xv = _virtual(x, fill=float("nan"))
    _lib.vae_groupnorm_stats(xv, geom, 1e-6, ws, stats)
    got_stats = stats.reshape(-1, 32, 2).cpu()
    assert torch.allclose(got_stats, torch.stack(stats_want), rtol=2e-5, atol=2e-6)
    zyb = F.pad(torch.cat([zy, zb], 1), (1, 1, 1, 1))
    zyb = torch.cat([zyb[:, :, :1]] * 2 + [zyb], 2)[0].permute(1, 2, 3, 0).contiguous().bfloat16().to(_dev())
    out = torch.full(((T + 2) * (H + 2) * (W + 2) * C,), 9.0, dtype=torch.bfloat16, device=_dev())
    _lib.vae_spatial_norm(xv, stats, gamma.to(_dev()), beta.to(_dev()), zyb, out, geom, silu=True)
    got = out.reshape(T + 2, H + 2, W + 2, C).float().cpu()
    assert bool((got[:, 0] == 0).all() and (got[:, -1] == 0).all() and (got[:, :, 0] == 0).all() and (got[:, :, -1] == 0).all())
    assert torch.equal(got[0], got[2]) and torch.equal(got[1], got[2])
    inner = got[2:, 1:-1, 1:-1].permute(3, 0, 1, 2)
    # same per-op bf16 roundings; the fp32 statistics differ in the last bits -> an occasional 1-ulp flip
    err = (inner - want[0]).abs()
    assert err.max().item() <= 2.0 ** -6 * max(1.0, want.abs().max().item()) and err.mean().item() < 1e-3


@pytest.mark.parametrize("compress,single", [(True, True), (True, False), (False, True)])
def test_upsample_is_exact(compress, single):
    g = torch.Generator().manual_seed(11)
    C, T, H, W = 128, 5, 4, 6
    x = torch.randn(1, C, T, H, W, generator=g).bfloat16().float()
    if compress:
        want = F.interpolate(x, scale_factor=2.0) if not single else torch.cat(
            [F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None], F.interpolate(x[:, :, 1:], scale_factor=2.0)], 2)
    else:
        want = x.repeat_interleave(2, 3).repeat_interleave(2, 4)
    T2 = want.shape[2]
    out = torch.full((T2 * (2 * H + 2) * (2 * W + 2) * C,), 5.0, dtype=torch.bfloat16, device=_dev())
    _lib.vae_upsample(_virtual(x, fill=float("nan")), out, T2, H, W, C, compress, single)
    got = out.reshape(T2, 2 * H + 2, 2 * W + 2, C).float().cpu()
    assert torch.equal(got[:, 1:-1, 1:-1].permute(3, 0, 1, 2), want[0])
    assert bool((got[:, 0] == 0).all() and (got[:, -1] == 0).all() and (got[:, :, 0] == 0).all() and (got[:, :, -1] == 0).all())


def test_pack_latent_and_unpack_video_are_exact():
    g = torch.Generator().manual_seed(12)
    L, h, w = 3, 4, 5
    z = torch.randn(L, 16, h, w, generator=g).bfloat16()           # the sampler's [F, C, h, w]
    out = torch.full(((L + 2) * (h + 2) * (w + 2) * 64,), 3.0, dtype=torch.bfloat16, device=_dev())
    s = float(torch.tensor(1 / 0.7, dtype=torch.float32))
    _lib.vae_pack_latent(z.to(_dev()), h * w, 16 * h * w, out, L, h, w, 16, s)
    got = out.reshape(L + 2, h + 2, w + 2, 64).cpu()
    want = (1 / 0.7 * z).permute(0, 2, 3, 1)                        # bf16 tensor * python float, as cog:430
    assert torch.equal(got[2:, 1:-1, 1:-1, :16], want) and bool((got[..., 16:] == 0).all())
    assert torch.equal(got[0], got[2]) and torch.equal(got[1], got[2]) and bool((got[:, 0] == 0).all())
    T, H, W = 4, 6, 300
    v = (torch.randn(1, 4, T, H, W, generator=g) * 0.8).bfloat16()
    vv = _virtual(v.float(), fill=float("nan"))
    nc = torch.empty(3, T, H, W, dtype=torch.bfloat16, device=_dev())
    _lib.vae_unpack_video(vv, nc, T, H, W, False)
    assert torch.equal(nc.cpu(), v[0, :3])
    u8 = torch.empty(T, H, W, 3, dtype=torch.uint8, device=_dev())
    _lib.vae_unpack_video(vv, u8, T, H, W, True)
    assert torch.equal(u8.cpu(), vae_oracle.postprocess_uint8(v[0, :3]))


def _decoder_pair(cfg_kw, L, h, w, seed):
    ocfg = vae_oracle.VAEConfig(**cfg_kw)
    sd = vae_oracle.synthetic_state_dict(ocfg, seed=seed)
    vae = AutoencoderKLCogVideoX(AutoencoderKLCogVideoXConfig(**cfg_kw), device=_dev()).load_state_dict(sd)
    g = torch.Generator().manual_seed(seed + 1)
    lat = torch.randn(1, L, 16, h, w, generator=g).bfloat16()
    return ocfg, sd, vae, lat


@pytest.mark.parametrize("L", [1, 3, 4, 5])
def test_decoder_matches_the_batched_oracle(L):
    """Whole-video HIP decode == the published batched decode (conv caches, per-batch GroupNorm, first-frame-single
    upsampling), including an even latent count (all frames doubled) and a single latent frame."""
    ocfg, sd, vae, lat = _decoder_pair(dict(layers_per_block=1), L, 4, 6, seed=20 + L)
    want = vae_oracle.decode_latents(lat.float(), sd, ocfg)
    got = vae.decode_latents(lat.to(_dev())).float().cpu()
    assert got.shape == want.shape and bool(torch.isfinite(got).all())
    rel = ((got - want).norm() / want.norm()).item()
    # ~14 bf16 convolutions and 21 normalisations deep; fp32 oracle
    assert rel < 3e-2, rel
    z = lat.permute(0, 2, 1, 3, 4).contiguous()
    alt = vae.decode((1 / 0.7 * z).to(_dev())).sample.float().cpu()
    assert torch.equal(alt, got)                                   # decode() on pre-scaled [B, C, L, h, w] is the same path


def test_decoder_full_depth_and_uint8_writer():
    ocfg, sd, vae, lat = _decoder_pair(dict(), 3, 2, 3, seed=31)  # published depth: 3 layers per block
    want = vae_oracle.decode_latents(lat.float(), sd, ocfg)
    got = vae.decode_latents(lat.to(_dev()))
    rel = ((got.float().cpu() - want).norm() / want.norm()).item()
    assert rel < 4e-2, rel
    u8 = vae.decode_latents(lat.to(_dev()), to_uint8=True)
    assert u8.shape == (1, 9, 16, 24, 3) and u8.dtype == torch.uint8
    assert torch.equal(u8[0].cpu(), vae_oracle.postprocess_uint8(got[0].cpu()))
    # deterministic: two runs are bit-identical (fixed-order GroupNorm reductions)
    assert torch.equal(vae.decode_latents(lat.to(_dev())), got)


def test_decoder_batch_of_two():
    ocfg, sd, vae, lat = _decoder_pair(dict(layers_per_block=1), 3, 2, 3, seed=41)
    lat2 = torch.cat([lat, lat.flip(1)], 0).contiguous()
    both = vae.decode_latents(lat2.to(_dev()))
    assert torch.equal(both[0], vae.decode_latents(lat.to(_dev()))[0])
    assert torch.equal(both[1], vae.decode_latents(lat.flip(1).contiguous().to(_dev()))[0])


def test_pipeline_decodes_through_the_hip_vae():
    """cog:1142-1148: with a VAE attached the CogVideoX pipeline returns frames; 'pt' goes through decode_latents +
    postprocess_video, 'uint8' (extension) / 'pil' through the fused writer kernel -- same pixels."""
    from alg_amd import CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline
    from alg_amd.transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig
    from oracle import dit_oracle
    dev = _dev()
    small = dict(num_attention_heads=8, attention_head_dim=64, in_channels=32, out_channels=16, num_layers=1,
                 time_embed_dim=64, text_embed_dim=128, max_text_seq_length=10, sample_width=6, sample_height=4,
                 sample_frames=9, patch_size=2)
    w = dit_oracle.init_weights(dit_oracle.DiTConfig(**small), seed=4, std=0.05, randomize_affine=True)
    tr = CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**small), {k: v.bfloat16() for k, v in w.items()}, device=dev)
    vae = AutoencoderKLCogVideoX.from_synthetic(AutoencoderKLCogVideoXConfig(layers_per_block=0), seed=5, device=dev)
    pipe = CogVideoXImageToVideoPipeline(transformer=tr, scheduler=CogVideoXDDIMScheduler(), vae=vae).to(dev)
    g = torch.Generator().manual_seed(1)
    kw = dict(image=None, prompt_embeds=torch.randn(1, 10, 128, generator=g).bfloat16(),
              negative_prompt_embeds=torch.randn(1, 10, 128, generator=g).bfloat16(),
              image_latents=(torch.randn(1, 1, 16, 4, 6, generator=g) * 0.7).bfloat16(), height=32, width=48,
              num_frames=9, num_inference_steps=2, use_low_pass_guidance=False)
    pt = pipe(**kw, generator=torch.Generator().manual_seed(2), output_type="pt").frames
    u8 = pipe(**kw, generator=torch.Generator().manual_seed(2), output_type="uint8").frames
    pil = pipe(**kw, generator=torch.Generator().manual_seed(2), output_type="pil").frames
    assert pt.shape == (1, 9, 3, 32, 48) and u8.shape == (1, 9, 32, 48, 3) and u8.dtype == torch.uint8
    want = (pt[0].permute(0, 2, 3, 1).float().cpu().numpy() * 255).round().astype("uint8")
    assert (u8[0].cpu().numpy() == want).all()
    import numpy as np
    assert len(pil[0]) == 9 and (np.asarray(pil[0][3]) == want[3]).all()


@pytest.mark.parametrize("Cin,Cout", [(128, 128), (256, 256)])
def test_stride2_conv_and_repitch(Cin, Cout):
    """CogVideoXDownsample3D's spatial part: pad (0, 1, 0, 1) + Conv2d k3 s2 p0, as ALG_CONV_STRIDE2 + alg_vae_repitch."""
    g = torch.Generator().manual_seed(Cin)
    T, H, W = 2, 8, 12
    x = torch.randn(1, Cin, T, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g).bfloat16().float()
    xp = F.pad(x[0].permute(1, 0, 2, 3), (0, 1, 0, 1))
    want = F.conv2d(xp, w, b, stride=2).permute(1, 0, 2, 3)                       # [Cout, T, H/2, W/2]
    wp = w.reshape(Cout, Cin, 9).permute(0, 2, 1).reshape(Cout, -1).contiguous().bfloat16().to(_dev())
    m = H // 2 * (W + 2)
    wide = torch.full((T * m * Cout,), 3.0, dtype=torch.bfloat16, device=_dev())
    _lib.conv_cl(_padded(x, 0), wp, b.bfloat16().to(_dev()), None, wide, T, H + 2, W + 2, Cin, Cout, 1, stride2=True)
    out = torch.full((T * (H // 2 + 2) * (W // 2 + 2) * Cout,), 5.0, dtype=torch.bfloat16, device=_dev())
    _lib.vae_repitch(wide, out, T, H // 2, W // 2, Cout, m, W + 2)
    got = _from_virtual(out, T, H // 2, W // 2, Cout)
    assert (got - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()


def test_group_norm_pad_and_planes_kernels():
    g = torch.Generator().manual_seed(3)
    C, H, W = 128, 6, 10
    x = (torch.randn(1, C, 1, H, W, generator=g) * 1.5 - 0.3).bfloat16().float()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).bfloat16()
    beta = (0.2 * torch.randn(C, generator=g)).bfloat16()
    geom = _lib.vae_geom(frames=1, H=H, W=W, C=C, first_len=1, seg_len=2, lat_first_single=1, lat_rate=1, lat_scale=1,
                         lat_h=H, lat_w=W)
    xv = _virtual(x, fill=float("nan"))
    ws = torch.empty(_lib.vae_groupnorm_workspace(geom) // 4, device=_dev())
    stats = torch.empty(64, device=_dev())
    _lib.vae_groupnorm_stats(xv, geom, 1e-6, ws, stats)
    out = torch.full((3 * (H + 2) * (W + 2) * C,), 9.0, dtype=torch.bfloat16, device=_dev())
    _lib.vae_group_norm(xv, stats, gamma.to(_dev()), beta.to(_dev()), out, geom, silu=True)
    got = out.reshape(3, H + 2, W + 2, C).float().cpu()
    want = F.silu(F.group_norm(x, 32, gamma.float(), beta.float(), 1e-6).bfloat16().float()).bfloat16().float()
    err = (got[2, 1:-1, 1:-1].permute(2, 0, 1) - want[0, :, 0]).abs()
    assert err.max().item() <= 2.0 ** -6 * max(1.0, want.abs().max().item()) and err.mean().item() < 1e-3
    assert torch.equal(got[0], got[2]) and bool((got[:, 0] == 0).all() and (got[:, :, -1] == 0).all())
    _lib.vae_pad(xv, out, 1, H, W, C)
    got = out.reshape(3, H + 2, W + 2, C).float().cpu()
    assert torch.equal(got[2, 1:-1, 1:-1].permute(2, 0, 1), x[0, :, 0]) and bool((got[:, 0] == 0).all())
    planes = torch.empty(C, 1, H, W, dtype=torch.bfloat16, device=_dev())
    _lib.vae_unpack_planes(xv, planes, 1, H, W, C)
    assert torch.equal(planes.float().cpu(), x[0])


@pytest.mark.parametrize("layers", [1, 3])
def test_encoder_matches_the_oracle(layers):
    """encode() on single frames (cog:388-391 / cog:645) vs the fp32 restatement; the sample uses the same CPU noise."""
    kw = dict(layers_per_block=layers)
    ocfg = vae_oracle.VAEConfig(**kw)
    sd = vae_oracle.synthetic_state_dict(ocfg, seed=50 + layers, encoder=True)
    vae = AutoencoderKLCogVideoX(AutoencoderKLCogVideoXConfig(**kw), device=_dev()).load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    img = (torch.rand(2, 3, 1, 32, 48, generator=g) * 2 - 1).bfloat16()
    want = vae_oracle.encode_moments(img.float(), sd, ocfg)
    dist = vae.encode(img.to(_dev())).latent_dist
    got = dist.parameters.float().cpu()
    assert got.shape == want.shape == (2, 32, 1, 4, 6)
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 3e-2, rel
    z = dist.sample(torch.Generator().manual_seed(4))
    noise = torch.randn(2, 16, 1, 4, 6, generator=torch.Generator().manual_seed(4), dtype=torch.bfloat16)
    assert torch.equal(z.cpu(), vae_oracle.gaussian_sample(dist.parameters.cpu(), noise))   # same eager ops, same noise
    assert torch.equal(dist.mode(), dist.mean)
    with pytest.raises(ValueError, match="single frames"):
        vae.encode(torch.zeros(1, 3, 2, 32, 48, dtype=torch.bfloat16, device=_dev()))


def test_pipeline_from_image_with_pixel_space_alg_branch():
    """End to end on HIP: image tensor -> VAE encode (cog:388-391) -> loop with `lp_filter_in_latent=False` (cog:628-680:
    the RGB image is filtered and re-encoded every step, drawing fresh posterior noise) -> VAE decode -> uint8 frames."""
    from alg_amd import CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, lp_utils
    from alg_amd.transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig
    from oracle import dit_oracle
    dev = _dev()
    small = dict(num_attention_heads=8, attention_head_dim=64, in_channels=32, out_channels=16, num_layers=1,
                 time_embed_dim=64, text_embed_dim=128, max_text_seq_length=10, sample_width=6, sample_height=4,
                 sample_frames=9, patch_size=2)
    w = dit_oracle.init_weights(dit_oracle.DiTConfig(**small), seed=4, std=0.05, randomize_affine=True)
    tr = CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**small), {k: v.bfloat16() for k, v in w.items()}, device=dev)
    vae = AutoencoderKLCogVideoX.from_synthetic(AutoencoderKLCogVideoXConfig(layers_per_block=1), seed=5, device=dev,
                                                encoder=True)
    pipe = CogVideoXImageToVideoPipeline(transformer=tr, scheduler=CogVideoXDDIMScheduler(), vae=vae).to(dev)
    g = torch.Generator().manual_seed(1)
    image = (torch.rand(1, 3, 32, 48, generator=g) * 2 - 1).bfloat16()
    kw = dict(image=image, prompt_embeds=torch.randn(1, 10, 128, generator=g).bfloat16(),
              negative_prompt_embeds=torch.randn(1, 10, 128, generator=g).bfloat16(), height=32, width=48, num_frames=9,
              num_inference_steps=3, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.5)
    run = lambda **over: pipe(**dict(kw, **over), generator=torch.Generator().manual_seed(2)).frames
    pix = run(lp_filter_in_latent=False, output_type="uint8")
    assert pix.shape == (1, 9, 32, 48, 3) and pix.dtype == torch.uint8
    assert torch.equal(pix, run(lp_filter_in_latent=False, output_type="uint8"))          # same seeds, same frames
    lat_pix = run(lp_filter_in_latent=False, output_type="latent")
    lat_lat = run(lp_filter_in_latent=True, output_type="latent")
    assert bool(torch.isfinite(lat_pix.float()).all()) and not torch.equal(lat_pix, lat_lat)
    # the conditioning frame the loop starts from is the scaled posterior sample of the image (cog:388-396)
    gen = torch.Generator().manual_seed(2)
    first = vae.encode(image.to(dev).unsqueeze(2)).latent_dist.sample(gen)
    _, cond = pipe.prepare_latents(image.to(dev), 1, 16, 9, 32, 48, torch.bfloat16, dev, torch.Generator().manual_seed(2))
    assert torch.equal(cond[:, :1], (0.7 * first).permute(0, 2, 1, 3, 4)) and bool((cond[:, 1:] == 0).all())
    # and the per-step pixel-branch conditioning is the scaled sample of the filtered image, zero-padded in time
    gen = torch.Generator().manual_seed(3)
    lp = pipe.prepare_lp("down_up", 0.0, 0, 0.25, gen, 9, True, False, cond, image.to(dev))
    img_lp = lp_utils.apply_low_pass_filter(image.to(dev), "down_up", 0.0, 0, 0.25)
    want = 0.7 * vae.encode(img_lp.unsqueeze(2)).latent_dist.sample(torch.Generator().manual_seed(3))
    assert lp.shape == cond.shape and torch.equal(lp[:, :1], want.permute(0, 2, 1, 3, 4)) and bool((lp[:, 1:] == 0).all())
