"""CPU checks of oracle/hunyuan_vae_oracle.py (the restatement of diffusers' AutoencoderKLHunyuanVideo used at hy:578-582 and
hy:1291-1292) and of the host-side structure of alg_amd/autoencoder_kl_hunyuan_video.py.  Parity unpinned (diffusers absent):
these pin the restatement to the published structure -- state-dict names / shapes of the released checkpoint layout,
replicate padding, stride placement, the block-causal mask and the temporal tiling arithmetic."""
import pytest
import torch

from oracle import hunyuan_vae_oracle as O

SMALL = dict(block_out_channels=[32, 32, 64, 64], latent_channels=4)


def test_published_layout_names_shapes_and_strides():
    cfg = O.HunyuanVAEConfig()
    s = O.param_shapes(cfg)
    # spatial x8 in blocks 0-2, temporal x4 in blocks 1-2 (encoder) -- mirrored in the decoder
    assert [st for *_, st in O.encoder_plan(cfg)] == [(1, 2, 2), (2, 2, 2), (2, 2, 2), None]
    assert [f for *_, f in O.decoder_plan(cfg)] == [(1, 2, 2), (2, 2, 2), (2, 2, 2), None]
    assert s["encoder.conv_in.conv.weight"] == (128, 3, 3, 3, 3)
    assert s["encoder.down_blocks.1.resnets.0.conv_shortcut.conv.weight"] == (256, 128, 1, 1, 1)
    assert "encoder.down_blocks.0.resnets.0.conv_shortcut.conv.weight" not in s
    assert s["encoder.down_blocks.2.downsamplers.0.conv.conv.weight"] == (512, 512, 3, 3, 3)
    assert "encoder.down_blocks.3.downsamplers.0.conv.conv.weight" not in s
    assert s["encoder.mid_block.attentions.0.to_out.0.weight"] == (512, 512)
    assert s["encoder.conv_out.conv.weight"] == (32, 512, 3, 3, 3) and s["quant_conv.weight"] == (32, 32, 1, 1, 1)
    assert s["decoder.conv_in.conv.weight"] == (512, 16, 3, 3, 3) and s["post_quant_conv.weight"] == (16, 16, 1, 1, 1)
    assert s["decoder.up_blocks.2.resnets.0.conv_shortcut.conv.weight"] == (256, 512, 1, 1, 1)
    assert s["decoder.up_blocks.3.resnets.2.conv2.conv.weight"] == (128, 128, 3, 3, 3)
    assert s["decoder.up_blocks.2.upsamplers.0.conv.conv.weight"] == (256, 256, 3, 3, 3)
    assert s["decoder.conv_out.conv.weight"] == (3, 128, 3, 3, 3)
    # the host module declares exactly the same state dict
    import inspect
    from alg_amd import autoencoder_kl_hunyuan_video as M
    vae = M.AutoencoderKLHunyuanVideo.__new__(M.AutoencoderKLHunyuanVideo)
    vae.config = M.AutoencoderKLHunyuanVideoConfig()
    assert vae.param_shapes() == s
    assert "oracle" not in inspect.getsource(M).replace("oracle/hunyuan_vae_oracle.py", "")


def test_causal_conv_is_replicate_padded_and_strided_on_the_padded_tensor():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 5, 4, 6, generator=g)
    sd = {"c.conv.weight": torch.randn(3, 2, 3, 3, 3, generator=g), "c.conv.bias": torch.randn(3, generator=g)}
    y = O.causal_conv(x, sd, "c")
    # naive: index clamping = replicate; taps reach two frames BACK
    w, b = sd["c.conv.weight"], sd["c.conv.bias"]
    ref = torch.zeros(1, 3, 5, 4, 6)
    for t in range(5):
        for yy in range(4):
            for xx in range(6):
                acc = b.clone()
                for dt in range(3):
                    for dy in range(3):
                        for dx in range(3):
                            tt = max(t + dt - 2, 0)
                            sy, sx = min(max(yy + dy - 1, 0), 3), min(max(xx + dx - 1, 0), 5)
                            acc = acc + w[:, :, dt, dy, dx] @ x[0, :, tt, sy, sx]
                ref[0, :, t, yy, xx] = acc
    assert torch.allclose(y, ref, atol=1e-4)
    # a stride-2 convolution = the stride-1 result sub-sampled from index 0 (what the HIP module relies on)
    assert torch.equal(O.causal_conv(x, sd, "c", stride=(2, 2, 2)), y[:, :, ::2, ::2, ::2])
    assert torch.equal(O.causal_conv(x, sd, "c", stride=(1, 2, 2)), y[:, :, :, ::2, ::2])
    assert O.causal_conv(x[:, :, :1], sd, "c", stride=(2, 2, 2)).shape == (1, 3, 1, 2, 3)


def test_block_causal_mask_and_upsample_rule():
    m = O.causal_attention_mask(3, 2, torch.float32)
    assert m.shape == (6, 6)
    for i in range(6):
        for j in range(6):
            assert m[i, j] == (0.0 if j // 2 <= i // 2 else float("-inf"))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 2, 3, 2, 2, generator=g)
    sd = {"u.conv.conv.weight": torch.zeros(2, 2, 3, 3, 3), "u.conv.conv.bias": torch.zeros(2)}
    sd["u.conv.conv.weight"][0, 0, 2, 1, 1] = 1.0                        # identity on channel 0 (tap = current voxel)
    y = O.upsample(x, sd, "u", (2, 2, 2))
    assert y.shape == (1, 2, 5, 4, 4)                                    # 1 + 2 * 2 frames: the first is not doubled
    assert torch.equal(y[0, 0, 0], x[0, 0, 0].repeat_interleave(2, 0).repeat_interleave(2, 1))
    assert torch.equal(y[0, 0, 1], y[0, 0, 2]) and torch.equal(y[0, 0, 3, ::2, ::2], x[0, 0, 2])
    assert O.upsample(x[:, :, :1], sd, "u", (2, 2, 2)).shape == (1, 2, 1, 4, 4)


@pytest.mark.parametrize("L", [1, 4, 5, 7, 8])
def test_temporal_tiling_arithmetic(L):
    cfg = O.HunyuanVAEConfig(**SMALL)
    sd = {k: v.float() for k, v in O.init_weights(cfg, seed=3).items()}
    z = torch.randn(1, 4, L, 2, 2, generator=torch.Generator().manual_seed(L))
    out = O.decode(z, sd, cfg)
    assert out.shape == (1, 3, 4 * (L - 1) + 1, 16, 16) and bool(torch.isfinite(out).all())
    if L <= 4:
        assert torch.equal(out, O._decode_tile(z, sd, cfg))
        return
    t0 = O._decode_tile(z[:, :, :5], sd, cfg)                            # 17 frames; 13 kept, none of them blended
    assert torch.equal(out[:, :, :13], t0[:, :, :13])
    t1 = O._decode_tile(z[:, :, 3:8], sd, cfg)[:, :, 1:]                 # its frames 0-3 cross-fade with t0's last four
    n = min(4, out.shape[2] - 13)
    for x in range(n):
        want = t0[:, :, 13 + x] * (1 - x / 4) + t1[:, :, x] * (x / 4)
        assert torch.allclose(out[:, :, 13 + x], want, atol=1e-6)
    if out.shape[2] > 17:
        assert torch.equal(out[:, :, 17: 25], t1[:, :, 4: 12][:, :, : out.shape[2] - 17])


def test_encode_shapes_and_limits():
    cfg = O.HunyuanVAEConfig(**SMALL)
    sd = {k: v.float() for k, v in O.init_weights(cfg, seed=4).items()}
    assert O.encode(torch.zeros(1, 3, 1, 16, 24), sd, cfg).shape == (1, 8, 1, 2, 3)
    assert O.encode(torch.zeros(1, 3, 9, 16, 16), sd, cfg).shape == (1, 8, 3, 2, 2)
    with pytest.raises(NotImplementedError):
        O.encode(torch.zeros(1, 3, 17, 16, 16), sd, cfg)
