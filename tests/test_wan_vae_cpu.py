"""CPU checks of the Wan VAE oracle (oracle/wan_vae_oracle.py: diffusers' AutoencoderKLWan restated in its published
chunked form) and of the host-side structure of the HIP implementation -- no GPU needed."""
import torch
import torch.nn.functional as F

from oracle import wan_vae_oracle as O


def _whole_video(cfg, sd):
    """The formulation alg_amd/autoencoder_kl_wan.py launches: every layer over the whole video, zero-padded causal
    convolutions, upsample3d leaves frame 0 undoubled and out of its temporal convolution, downsample3d passes frame 0 through
    and convolves frames (2t - 2, 2t - 1, 2t)."""
    cconv = lambda x, n, pad: O.causal_conv(x, sd[n + ".weight"], sd[n + ".bias"], pad)

    def res(x, name):
        h = cconv(x, name + ".conv_shortcut", (0, 0, 0)) if name + ".conv_shortcut.weight" in sd else x
        x = cconv(F.silu(O.rms_norm(x, sd[name + ".norm1.gamma"])), name + ".conv1", (1, 1, 1))
        x = cconv(F.silu(O.rms_norm(x, sd[name + ".norm2.gamma"])), name + ".conv2", (1, 1, 1))
        return x + h

    def mid(x, p):
        return res(O.attention_block(res(x, p + ".resnets.0"), sd, p + ".attentions.0"), p + ".resnets.1")

    def spatial(x, name, up):
        b, ch, t, h, w = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(b * t, ch, h, w)
        rw, rb = sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"]
        if up:
            x = F.conv2d(F.interpolate(x, scale_factor=(2.0, 2.0), mode="nearest-exact"), rw, rb, padding=1)
        else:
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), rw, rb, stride=2)
        return x.view(b, t, x.size(1), x.size(2), x.size(3)).permute(0, 2, 1, 3, 4)

    def resample(x, name, mode):
        b, ch, t, h, w = x.shape
        if mode == "upsample3d" and t > 1:
            xx = x.clone()
            xx[:, :, 0] = 0
            y = cconv(xx, name + ".time_conv", (1, 0, 0))[:, :, 1:]
            y = y.reshape(b, 2, ch, t - 1, h, w)
            y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, ch, 2 * (t - 1), h, w)
            x = torch.cat([x[:, :, :1], y], 2)
        x = spatial(x, name, mode.startswith("up"))
        if mode == "downsample3d" and x.shape[2] > 1:
            y = F.conv3d(x, sd[name + ".time_conv.weight"], sd[name + ".time_conv.bias"], stride=(2, 1, 1))
            x = torch.cat([x[:, :, :1], y], 2)
        return x

    def enc(x):
        plan, top = O.encoder_plan(cfg)
        x = cconv(x, "encoder.conv_in", (1, 1, 1))
        for kind, name, ci, co in plan:
            x = res(x, name) if kind == "res" else resample(x, name, kind)
        x = F.silu(O.rms_norm(mid(x, "encoder.mid_block"), sd["encoder.norm_out.gamma"]))
        return cconv(cconv(x, "encoder.conv_out", (1, 1, 1)), "quant_conv", (0, 0, 0))

    def dec(z):
        plan, top, last = O.decoder_plan(cfg)
        x = mid(cconv(cconv(z, "post_quant_conv", (0, 0, 0)), "decoder.conv_in", (1, 1, 1)), "decoder.mid_block")
        for kind, name, ci, co in plan:
            x = res(x, name) if kind == "res" else resample(x, name, kind)
        x = F.silu(O.rms_norm(x, sd["decoder.norm_out.gamma"]))
        return cconv(x, "decoder.conv_out", (1, 1, 1)).clamp(-1, 1)

    return enc, dec


def test_chunked_published_form_equals_the_whole_video_form():
    cfg = O.WanVAEConfig(base_dim=8, z_dim=4, latents_mean=[0.0] * 4, latents_std=[1.0] * 4)
    sd = {k: v.float() for k, v in O.init_weights(cfg, seed=1).items()}
    enc, dec = _whole_video(cfg, sd)
    g = torch.Generator().manual_seed(0)
    for frames in (1, 9, 17):
        x = torch.randn(1, 3, frames, 16, 24, generator=g)
        m = O.encode(cfg, sd, x)
        assert m.shape == (1, 8, 1 + (frames - 1) // 4, 2, 3)
        assert (enc(x) - m).abs().max().item() <= 1e-5
    for lat in (1, 2, 5):
        z = torch.randn(1, 4, lat, 2, 3, generator=g)
        y = O.decode(cfg, sd, z)
        assert y.shape == (1, 3, 4 * (lat - 1) + 1, 16, 24) and float(y.abs().max()) <= 1.0
        assert (dec(z) - y).abs().max().item() <= 1e-5
    # causality: later latent frames cannot change earlier output frames
    z = torch.randn(1, 4, 4, 2, 3, generator=g)
    z2 = z.clone()
    z2[:, :, 3] += 1.0
    a, b = O.decode(cfg, sd, z), O.decode(cfg, sd, z2)
    assert torch.equal(a[:, :, :9], b[:, :, :9]) and not torch.equal(a[:, :, 9:], b[:, :, 9:])


def test_published_structure():
    cfg = O.WanVAEConfig()
    shapes = O.param_shapes(cfg)
    n = sum(torch.Size(s).numel() for s in shapes.values())
    assert 120e6 < n < 135e6                      # Wan 2.1 VAE: 127 M parameters
    assert shapes["encoder.down_blocks.5.resample.1.weight"] == (192, 192, 3, 3)       # level 1 -> downsample3d
    assert shapes["encoder.down_blocks.5.time_conv.weight"] == (192, 192, 3, 1, 1)
    assert shapes["decoder.up_blocks.0.upsamplers.0.time_conv.weight"] == (768, 384, 3, 1, 1)
    assert shapes["decoder.up_blocks.1.resnets.0.conv_shortcut.weight"] == (384, 192, 1, 1, 1)
    assert shapes["decoder.mid_block.attentions.0.to_qkv.weight"] == (1152, 384, 1, 1)
    from alg_amd.autoencoder_kl_wan import AutoencoderKLWan, AutoencoderKLWanConfig
    vae = AutoencoderKLWan.__new__(AutoencoderKLWan)                # structure only (the constructor needs a GPU)
    vae.config = AutoencoderKLWanConfig()
    vae.temperal_downsample = list(vae.config.temperal_downsample)
    vae.temperal_upsample = vae.temperal_downsample[::-1]
    assert vae.param_shapes() == shapes
    assert vae._encoder_plan() == O.encoder_plan(cfg) and vae._decoder_plan() == O.decoder_plan(cfg)
