"""CPU checks for the VAE decoder row: the oracle's structural properties, the host-side geometry
and the weight packing of the product class (no GPU, no compute through the C ABI)."""
import pytest
import torch
import torch.nn.functional as F

from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX, AutoencoderKLCogVideoXConfig, _Level
from oracle import vae_oracle


def test_param_tables_agree_and_match_published_counts():
    vae = AutoencoderKLCogVideoX(device="cpu")
    shapes = vae_oracle.decoder_param_shapes(vae_oracle.VAEConfig())
    assert vae.param_shapes() == shapes
    assert len(vae._resnets()) == 2 + 4 * 4
    assert shapes["decoder.conv_in.conv.weight"] == (512, 16, 3, 3, 3)
    assert shapes["decoder.up_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1, 1)
    assert shapes["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"] == (128, 256, 1, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in shapes
    assert shapes["decoder.conv_out.conv.weight"] == (3, 128, 3, 3, 3)
    n = sum(torch.Size(s).numel() for s in shapes.values())
    assert 90e6 < n < 130e6                      # the published decoder is ~0.1 B parameters


@pytest.mark.parametrize("L,frames", [(13, 49), (21, 81), (1, 1), (4, 16), (3, 9)])
def test_level_geometry(L, frames):
    top = _Level(L, 60, 90, 4, 8)
    assert (top.T, top.H, top.W) == (frames, 480, 720)
    mid = _Level(L, 60, 90, 1, 1)
    assert mid.T == L and mid.first_len == (L if L < 2 else 2 + L % 2) and mid.seg_len == 2
    if L % 2 == 1 and L > 1:
        assert top.first_len == 9 and top.seg_len == 8 and (top.T - top.first_len) % top.seg_len == 0


def test_weight_packing_is_tap_major_channels_last():
    cfg = AutoencoderKLCogVideoXConfig(layers_per_block=0)
    vae = AutoencoderKLCogVideoX(cfg, device="cpu")
    sd = vae_oracle.synthetic_state_dict(vae_oracle.VAEConfig(layers_per_block=0), seed=3)
    vae.load_state_dict(sd)
    w, b, pair = vae.w["decoder.up_blocks.2.resnets.0.conv1"]
    ref = sd["decoder.up_blocks.2.resnets.0.conv1.conv.weight"]
    assert w.shape == (256, 27 * 256) and w.dtype == torch.bfloat16 and not pair
    assert torch.equal(w.float().reshape(256, 3, 3, 3, 256)[5, 2, 0, 1], ref[5, :, 2, 0, 1])
    # 128 output channels: two neighbouring voxels per GEMM row, kernel at dx 0..2 for the first, dx 1..3 for the second
    w, b, pair = vae.w["decoder.up_blocks.3.resnets.0.conv1"]
    ref = sd["decoder.up_blocks.3.resnets.0.conv1.conv.weight"]
    assert pair and w.shape == (256, 36 * 256) and b.shape == (256,) and torch.equal(b[:128], b[128:])
    w5 = w.float().reshape(2, 128, 3, 3, 4, 256)
    assert torch.equal(w5[0, 7, 1, 2, 0:3], ref[7, :, 1, 2, :].T) and bool((w5[0, :, :, :, 3] == 0).all())
    assert torch.equal(w5[1, 7, 1, 2, 1:4], ref[7, :, 1, 2, :].T) and bool((w5[1, :, :, :, 0] == 0).all())
    w, b, _ = vae.w["decoder.conv_in"]
    assert w.shape == (512, 27 * 64) and bool((w.reshape(512, 27, 64)[:, :, 16:] == 0).all())
    w, b, _ = vae.w["decoder.conv_out"]
    assert w.shape == (4, 27 * 128) and bool((w[3] == 0).all()) and b[3] == 0
    gamma, beta, wyb, byb = vae.w["decoder.norm_out"]
    assert wyb.shape == (256, 64) and byb.shape == (256,)
    assert torch.equal(wyb[128:, :16].float(), sd["decoder.norm_out.conv_b.conv.weight"].reshape(128, 16))
    with pytest.raises(KeyError):
        vae.load_state_dict({})
    with pytest.raises(ValueError, match="powers of two"):
        AutoencoderKLCogVideoX(AutoencoderKLCogVideoXConfig(block_out_channels=(96, 128, 128, 128)), device="cpu")


def test_oracle_decode_is_causal_per_latent_batch():
    """Later latent batches cannot change earlier output frames (causal convolutions, GroupNorm per batch), and the
    output has 4(L-1)+1 frames at 8x the latent resolution."""
    cfg = vae_oracle.VAEConfig(block_out_channels=(32, 32, 32, 32), layers_per_block=0)
    sd = vae_oracle.synthetic_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 16, 5, 2, 3, generator=g)
    out = vae_oracle.decode(z, sd, cfg)
    assert out.shape == (1, 3, 17, 16, 24)
    z2 = z.clone()
    z2[:, :, 3:] += 1.0
    out2 = vae_oracle.decode(z2, sd, cfg)
    assert torch.equal(out2[:, :, :9], out[:, :, :9]) and not torch.equal(out2[:, :, 9:], out[:, :, 9:])
    # a single latent frame decodes to a single image
    assert vae_oracle.decode(z[:, :, :1], sd, cfg).shape == (1, 3, 1, 16, 24)
    # an even count doubles every frame
    assert vae_oracle.decode(z[:, :, :4], sd, cfg).shape == (1, 3, 16, 16, 24)


def test_oracle_postprocess_uint8():
    v = torch.tensor([-1.5, -1.0, 0.0, 0.5, 1.0, 2.0]).bfloat16().reshape(1, 1, 1, 6).repeat(3, 1, 1, 1)
    u = vae_oracle.postprocess_uint8(v)
    assert u.shape == (1, 1, 6, 3) and u[0, 0, :, 0].tolist() == [0, 0, 128, 191, 255, 255]


def test_decode_requires_device_tensors():
    from alg_amd import _lib
    vae = AutoencoderKLCogVideoX(device="cpu")
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        vae.decode(torch.zeros(1, 16, 3, 2, 2, dtype=torch.bfloat16))


def test_encoder_tables_and_oracle_shapes():
    vae = AutoencoderKLCogVideoX(device="cpu")
    assert vae.encoder_param_shapes() == vae_oracle.encoder_param_shapes(vae_oracle.VAEConfig())
    es = vae.encoder_param_shapes()
    assert es["encoder.conv_in.conv.weight"] == (128, 3, 3, 3, 3) and es["encoder.conv_out.conv.weight"] == (32, 512, 3, 3, 3)
    assert es["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in es
    cfg = vae_oracle.VAEConfig(block_out_channels=(32, 32, 64, 64), layers_per_block=1)
    sd = vae_oracle.synthetic_state_dict(cfg, seed=1, encoder=True)
    # synthetic conv_y biases stay near 1 (the decoder's norms must not collapse) -- guards the name classifier
    assert abs(sd["decoder.norm_out.conv_y.conv.bias"].mean().item() - 1.0) < 0.1
    g = torch.Generator().manual_seed(0)
    m1 = vae_oracle.encode_moments(torch.randn(1, 3, 1, 32, 48, generator=g), sd, cfg)
    assert m1.shape == (1, 32, 1, 4, 6)
    # 9 frames -> 1 + 8/4 = 3 latent frames (first frame kept through both temporal poolings)
    assert vae_oracle.encode_moments(torch.randn(1, 3, 9, 32, 48, generator=g), sd, cfg).shape == (1, 32, 3, 4, 6)
    noise = torch.randn(1, 16, 1, 4, 6, generator=g)
    z = vae_oracle.gaussian_sample(m1, noise)
    assert torch.allclose(z, m1[:, :16] + torch.exp(0.5 * m1[:, 16:]) * noise)


def test_product_vae_without_encoder_weights_refuses_encode():
    from alg_amd import _lib
    vae = AutoencoderKLCogVideoX.from_synthetic(AutoencoderKLCogVideoXConfig(layers_per_block=0), device="cpu")
    with pytest.raises(_lib.AlgHipError, match="without encoder"):
        vae.encode(torch.zeros(1, 3, 1, 16, 16, dtype=torch.bfloat16))
    with pytest.raises(ValueError, match="layers_per_block"):
        AutoencoderKLCogVideoX.from_synthetic(AutoencoderKLCogVideoXConfig(layers_per_block=0), device="cpu", encoder=True)
    vae = AutoencoderKLCogVideoX.from_synthetic(AutoencoderKLCogVideoXConfig(layers_per_block=1), device="cpu", encoder=True)
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        vae.encode(torch.zeros(1, 3, 1, 16, 16, dtype=torch.bfloat16))
