"""CPU restatement of diffusers' `AutoencoderKLHunyuanVideo` (the component
`pipeline_hunyuan_video_image2video_lowpass.py:578-582` encodes the conditioning image with and `:1291-1292` decodes the
final latents with; third-party, diffusers @ be2fb77, NOT in the reference tree -- **parity unpinned**: restated from the
published module structure).

TEST INFRASTRUCTURE -- never imported by ``alg_amd``.

What is restated:
  * `HunyuanVideoCausalConv3d`: REPLICATE padding -- k - 1 copies of the first frame in front, one replicated row / column
    on every spatial side -- then a plain Conv3d (stride on the padded tensor);
  * `HunyuanVideoResnetBlockCausal3D` (GroupNorm 32 / SiLU / conv, twice; 1x1x1 shortcut when the width changes);
  * `HunyuanVideoMidBlock3D`: resnet, one-head attention over ALL T * H * W tokens with the block-causal frame mask
    (`prepare_causal_attention_mask`: a token sees the tokens of its own and of earlier frames), GroupNorm in front,
    residual behind, resnet;
  * `HunyuanVideoDownsampleCausal3D` (stride-(1|2, 2, 2) causal convolution) and `HunyuanVideoUpsampleCausal3D` (nearest:
    the first frame x (2, 2) in space only, the other frames x (2, 2, 2); then a causal convolution);
  * `encode` of up to 16 frames (no temporal tiling below that; the reference encodes ONE frame);
  * `decode` with the defaults the reference runs with (`use_framewise_decoding` on, spatial tiling off): latents of more
    than 4 frames go through `_temporal_tiled_decode` -- tiles of 5 latent frames every 3, first decoded frame of every
    later tile dropped, 4-frame linear cross-fade `blend_t`, 12 frames kept per tile (13 of the first).
Weights use the diffusers state-dict names.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F


@dataclass
class HunyuanVAEConfig:
    """Defaults = hunyuanvideo-community/HunyuanVideo-I2V vae/config.json."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    block_out_channels: List[int] = field(default_factory=lambda: [128, 256, 512, 512])
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.476986
    spatial_compression_ratio: int = 8
    temporal_compression_ratio: int = 4
    mid_block_add_attention: bool = True
    # tiling constants of the published class (not config entries)
    tile_sample_min_num_frames: int = 16
    tile_sample_stride_num_frames: int = 12


def encoder_plan(cfg):
    """[(block index, in, out, downsample stride or None)] of `HunyuanVideoEncoder3D.down_blocks`."""
    boc = list(cfg.block_out_channels)
    n = len(boc)
    n_space = (cfg.spatial_compression_ratio).bit_length() - 1
    n_time = (cfg.temporal_compression_ratio).bit_length() - 1
    plan, ci = [], boc[0]
    for i, co in enumerate(boc):
        stride = None
        if i < n - 1:
            space = i < n_space
            time = i >= n - 1 - n_time
            stride = (2 if time else 1, 2 if space else 1, 2 if space else 1)
        plan.append((i, ci, co, stride))
        ci = co
    return plan


def decoder_plan(cfg):
    """[(block index, in, out, upsample factor or None)] of `HunyuanVideoDecoder3D.up_blocks`."""
    boc = list(cfg.block_out_channels)[::-1]
    n = len(boc)
    n_space = (cfg.spatial_compression_ratio).bit_length() - 1
    n_time = (cfg.temporal_compression_ratio).bit_length() - 1
    plan, ci = [], boc[0]
    for i, co in enumerate(boc):
        factor = None
        if i < n - 1:
            space = i < n_space
            time = i >= n - 1 - n_time
            factor = (2 if time else 1, 2 if space else 1, 2 if space else 1)
        plan.append((i, ci, co, factor))
        ci = co
    return plan


def param_shapes(cfg):
    s = {}

    def conv(name, ci, co, k=3):
        s[name + ".conv.weight"], s[name + ".conv.bias"] = (co, ci, k, k, k), (co,)

    def norm(name, c):
        s[name + ".weight"], s[name + ".bias"] = (c,), (c,)

    def res(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", ci, co)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co)
        if ci != co:
            conv(name + ".conv_shortcut", ci, co, 1)

    def mid(prefix, c):
        res(prefix + ".resnets.0", c, c)
        if cfg.mid_block_add_attention:
            a = prefix + ".attentions.0"
            norm(a + ".group_norm", c)
            for p in ("to_q", "to_k", "to_v", "to_out.0"):
                s["%s.%s.weight" % (a, p)], s["%s.%s.bias" % (a, p)] = (c, c), (c,)
        res(prefix + ".resnets.1", c, c)

    boc = list(cfg.block_out_channels)
    conv("encoder.conv_in", cfg.in_channels, boc[0])
    for i, ci, co, stride in encoder_plan(cfg):
        for j in range(cfg.layers_per_block):
            res("encoder.down_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
        if stride is not None:
            conv("encoder.down_blocks.%d.downsamplers.0.conv" % i, co, co)
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1])
    conv("encoder.conv_out", boc[-1], 2 * cfg.latent_channels)
    s["quant_conv.weight"], s["quant_conv.bias"] = (2 * cfg.latent_channels,) * 2 + (1, 1, 1), (2 * cfg.latent_channels,)
    s["post_quant_conv.weight"], s["post_quant_conv.bias"] = (cfg.latent_channels,) * 2 + (1, 1, 1), (cfg.latent_channels,)
    conv("decoder.conv_in", cfg.latent_channels, boc[-1])
    mid("decoder.mid_block", boc[-1])
    for i, ci, co, factor in decoder_plan(cfg):
        for j in range(cfg.layers_per_block + 1):
            res("decoder.up_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
        if factor is not None:
            conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, co, co)
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cfg.out_channels)
    return s


def init_weights(cfg, seed=0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) == 1 and name.endswith(".weight"):                 # GroupNorm scale
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            t = torch.randn(shape, generator=g) * (1.2 / fan ** 0.5)
        sd[name] = t.to(dtype)
    return sd


# ---- modules ---------------------------------------------------------------------------------------------------------
def causal_conv(x, sd, name, stride=(1, 1, 1)):
    """`HunyuanVideoCausalConv3d.forward`: F.pad(mode="replicate") by (k//2, k//2, k//2, k//2, k - 1, 0), then Conv3d."""
    w, b = sd[name + ".conv.weight"].to(x.dtype), sd[name + ".conv.bias"].to(x.dtype)
    k = w.shape[-1]
    if k > 1:
        x = F.pad(x, (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return F.conv3d(x, w, b, stride=stride)


def group_norm(x, sd, name, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[name + ".weight"].to(x.dtype), sd[name + ".bias"].to(x.dtype), eps)


def resnet(x, sd, name, cfg):
    h = F.silu(group_norm(x, sd, name + ".norm1", cfg.norm_num_groups))
    h = causal_conv(h, sd, name + ".conv1")
    h = F.silu(group_norm(h, sd, name + ".norm2", cfg.norm_num_groups))
    h = causal_conv(h, sd, name + ".conv2")                              # dropout 0
    if name + ".conv_shortcut.conv.weight" in sd:
        x = causal_conv(x, sd, name + ".conv_shortcut")
    return h + x


def causal_attention_mask(frames, tokens_per_frame, dtype):
    """`prepare_causal_attention_mask`: 0 where key frame <= query frame, -inf elsewhere."""
    idx = torch.arange(frames).repeat_interleave(tokens_per_frame)
    return torch.where(idx[None, :] <= idx[:, None], 0.0, float("-inf")).to(dtype)


def mid_attention(x, sd, name, cfg):
    """`HunyuanVideoMidBlock3D.forward` attention step: tokens = (t, h, w) flattened; `Attention(heads=1, dim_head=C,
    norm_num_groups=32, residual_connection=True, bias=True)` through `AttnProcessor2_0`."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, C)
    mask = causal_attention_mask(T, H * W, x.dtype)
    res = tok
    h = group_norm(tok.transpose(1, 2), sd, name + ".group_norm", cfg.norm_num_groups).transpose(1, 2)
    lin = lambda t, p: F.linear(t, sd["%s.%s.weight" % (name, p)].to(t.dtype), sd["%s.%s.bias" % (name, p)].to(t.dtype))
    q, k, v = lin(h, "to_q"), lin(h, "to_k"), lin(h, "to_v")
    s = (q @ k.transpose(1, 2)) * (C ** -0.5) + mask
    o = torch.softmax(s.float(), dim=-1).to(x.dtype) @ v
    o = lin(o, "to_out.0") + res
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def mid_block(x, sd, prefix, cfg):
    x = resnet(x, sd, prefix + ".resnets.0", cfg)
    if cfg.mid_block_add_attention:
        x = mid_attention(x, sd, prefix + ".attentions.0", cfg)
    return resnet(x, sd, prefix + ".resnets.1", cfg)


def upsample(x, sd, name, factor):
    """`HunyuanVideoUpsampleCausal3D.forward`: first frame nearest x factor[1:], other frames nearest x factor, concat, conv."""
    first = F.interpolate(x[:, :, 0], scale_factor=tuple(float(f) for f in factor[1:]), mode="nearest").unsqueeze(2)
    if x.shape[2] > 1:
        other = F.interpolate(x[:, :, 1:].contiguous(), scale_factor=tuple(float(f) for f in factor), mode="nearest")
        x = torch.cat([first, other], dim=2)
    else:
        x = first
    return causal_conv(x, sd, name + ".conv")


def encoder(x, sd, cfg):
    h = causal_conv(x, sd, "encoder.conv_in")
    for i, ci, co, stride in encoder_plan(cfg):
        for j in range(cfg.layers_per_block):
            h = resnet(h, sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), cfg)
        if stride is not None:
            h = causal_conv(h, sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, stride=stride)
    h = mid_block(h, sd, "encoder.mid_block", cfg)
    h = F.silu(group_norm(h, sd, "encoder.conv_norm_out", cfg.norm_num_groups))
    return causal_conv(h, sd, "encoder.conv_out")


def decoder(z, sd, cfg):
    h = causal_conv(z, sd, "decoder.conv_in")
    h = mid_block(h, sd, "decoder.mid_block", cfg)
    for i, ci, co, factor in decoder_plan(cfg):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(h, sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), cfg)
        if factor is not None:
            h = upsample(h, sd, "decoder.up_blocks.%d.upsamplers.0" % i, factor)
    h = F.silu(group_norm(h, sd, "decoder.conv_norm_out", cfg.norm_num_groups))
    return causal_conv(h, sd, "decoder.conv_out")


def encode(x, sd, cfg):
    """`AutoencoderKLHunyuanVideo._encode` below the temporal tiling threshold -> moments [B, 2 z, T', H / 8, W / 8]."""
    if x.shape[2] > cfg.tile_sample_min_num_frames:
        raise NotImplementedError("temporal tiled ENCODE is not on the reference's path (it encodes one frame)")
    h = encoder(x, sd, cfg)
    return F.conv3d(h, sd["quant_conv.weight"].to(h.dtype), sd["quant_conv.bias"].to(h.dtype))


def _decode_tile(z, sd, cfg):
    z = F.conv3d(z, sd["post_quant_conv.weight"].to(z.dtype), sd["post_quant_conv.bias"].to(z.dtype))
    return decoder(z, sd, cfg)


def blend_t(a, b, extent):
    extent = min(a.shape[2], b.shape[2], extent)
    for x in range(extent):
        b[:, :, x] = a[:, :, -extent + x] * (1 - x / extent) + b[:, :, x] * (x / extent)
    return b


def decode(z, sd, cfg):
    """`AutoencoderKLHunyuanVideo._decode` with framewise decoding on and spatial tiling off."""
    ratio = cfg.temporal_compression_ratio
    lat_min = cfg.tile_sample_min_num_frames // ratio
    L = z.shape[2]
    if L <= lat_min:
        return _decode_tile(z, sd, cfg)
    lat_stride = cfg.tile_sample_stride_num_frames // ratio
    blend = cfg.tile_sample_min_num_frames - cfg.tile_sample_stride_num_frames
    row = []
    for i in range(0, L, lat_stride):
        d = _decode_tile(z[:, :, i: i + lat_min + 1], sd, cfg)
        row.append(d[:, :, 1:] if i > 0 else d)
    out = []
    for i, tile in enumerate(row):
        if i > 0:
            tile = blend_t(row[i - 1], tile, blend)
            out.append(tile[:, :, : cfg.tile_sample_stride_num_frames])
        else:
            out.append(tile[:, :, : cfg.tile_sample_stride_num_frames + 1])
    return torch.cat(out, dim=2)[:, :, : (L - 1) * ratio + 1]
