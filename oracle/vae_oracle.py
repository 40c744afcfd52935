"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CogVideoX VAE *decoder* the reference calls at cog:428-433
(`frames = self.vae.decode(latents).sample`) and of the output conversion after it (cog:1148 `postprocess_video`,
run:121-125).  **Parity unpinned**: the model lives in a third-party dependency that is absent from /root/reference
(diffusers, `AutoencoderKLCogVideoX`, pinned by the reference's requirements at 0.33.x); neither the package nor a
checkpoint is available here, so this file restates the published module structure and is anchored only on the reference's
call sites (`decode_latents` cog:427-433, scaling cog:430, `vae_scale_factor_*` cog:217-223).

Restated as published (module by module, so the product's whole-video formulation is checked against the batched one):
  * `_decode`: latent frames are decoded in batches of `num_latent_frames_batch_size` = 2 (the first batch also takes the
    remainder, i.e. 3 frames for an odd count), carrying a per-convolution `conv_cache` of the last two input frames;
  * `CogVideoXCausalConv3d`: the first frame repeated kernel-1 times in front (first batch) or the cache (later ones),
    zero padding in height / width;
  * `CogVideoXSpatialNorm3D`: GroupNorm(32, eps 1e-6)(f) * conv_y(zq') + conv_b(zq'), zq' = nearest interpolation of the
    batch's latent to f's size, with the first frame interpolated on its own when f has an odd frame count > 1;
  * `CogVideoXResnetBlock3D`: norm1 - silu - conv1 - norm2 - silu - conv2 (+ 1x1x1 `conv_shortcut` when channels change);
  * `CogVideoXUpsample3D`: nearest x2 in space, and in time too (`compress_time`) with the first frame of an odd batch
    kept single; then a per-frame Conv2d 3x3;
  * decoder: conv_in, mid block (2 resnets), 4 up blocks (layers_per_block + 1 resnets each; upsample on all but the last,
    `compress_time` on the first log2(temporal_compression_ratio)), norm_out, silu, conv_out.
  * encoder (`encode`, used by the reference on single frames only: cog:388-391, cog:645): conv_in, 4 down blocks
    (layers_per_block resnets with plain GroupNorm(32, eps 1e-6); `CogVideoXDownsample3D` = optional temporal average
    pooling with the first frame kept, pad (0,1,0,1), per-frame Conv2d k3 s2 p0), mid block, GroupNorm, silu, conv_out
    to 2 x latent channels; frames go through in batches of `num_sample_frames_batch_size` = 8 with conv caches;
    `DiagonalGaussianDistribution`: logvar clamped to [-30, 20], sample = mean + exp(0.5 logvar) * noise.
State-dict names are diffusers' (`decoder.up_blocks.0.resnets.1.norm1.conv_y.conv.weight` ...).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


class VAEConfig:
    def __init__(self, block_out_channels=(128, 256, 256, 512), latent_channels=16, layers_per_block=3, out_channels=3,
                 norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4, scaling_factor=0.7,
                 invert_scale_latents=False):
        self.block_out_channels = tuple(block_out_channels)
        self.latent_channels = latent_channels
        self.layers_per_block = layers_per_block
        self.out_channels = out_channels
        self.norm_eps = norm_eps
        self.norm_num_groups = norm_num_groups
        self.temporal_compression_ratio = temporal_compression_ratio
        self.scaling_factor = scaling_factor
        self.invert_scale_latents = invert_scale_latents


def decoder_param_shapes(cfg):
    """name -> shape of every decoder parameter, in diffusers' naming."""
    zc, rev = cfg.latent_channels, list(reversed(cfg.block_out_channels))
    out = {}

    def conv3(name, ci, co, k=3):
        out[name + ".conv.weight"], out[name + ".conv.bias"] = (co, ci, k, k, k), (co,)

    def snorm(name, c):
        out[name + ".norm_layer.weight"], out[name + ".norm_layer.bias"] = (c,), (c,)
        conv3(name + ".conv_y", zc, c, 1)
        conv3(name + ".conv_b", zc, c, 1)

    def resnet(name, ci, co):
        snorm(name + ".norm1", ci)
        conv3(name + ".conv1", ci, co)
        snorm(name + ".norm2", co)
        conv3(name + ".conv2", co, co)
        if ci != co:
            out[name + ".conv_shortcut.weight"], out[name + ".conv_shortcut.bias"] = (co, ci, 1, 1, 1), (co,)

    conv3("decoder.conv_in", zc, rev[0])
    for j in range(2):
        resnet("decoder.mid_block.resnets.%d" % j, rev[0], rev[0])
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else c, c)
        if i != len(rev) - 1:
            out["decoder.up_blocks.%d.upsamplers.0.conv.weight" % i] = (c, c, 3, 3)
            out["decoder.up_blocks.%d.upsamplers.0.conv.bias" % i] = (c,)
        prev = c
    snorm("decoder.norm_out", rev[-1])
    conv3("decoder.conv_out", rev[-1], cfg.out_channels)
    return out


def encoder_param_shapes(cfg, in_channels=3):
    boc = list(cfg.block_out_channels)
    out = {}

    def conv3(name, ci, co):
        out[name + ".conv.weight"], out[name + ".conv.bias"] = (co, ci, 3, 3, 3), (co,)

    def resnet(name, ci, co):
        out[name + ".norm1.weight"], out[name + ".norm1.bias"] = (ci,), (ci,)
        conv3(name + ".conv1", ci, co)
        out[name + ".norm2.weight"], out[name + ".norm2.bias"] = (co,), (co,)
        conv3(name + ".conv2", co, co)
        if ci != co:
            out[name + ".conv_shortcut.weight"], out[name + ".conv_shortcut.bias"] = (co, ci, 1, 1, 1), (co,)

    prev = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet("encoder.down_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else c, c)
        if i != len(boc) - 1:
            out["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (c, c, 3, 3)
            out["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (c,)
        prev = c
    for j in range(2):
        resnet("encoder.mid_block.resnets.%d" % j, boc[-1], boc[-1])
    conv3("encoder.conv_in", in_channels, boc[0])
    out["encoder.norm_out.weight"], out["encoder.norm_out.bias"] = (boc[-1],), (boc[-1],)
    conv3("encoder.conv_out", boc[-1], 2 * cfg.latent_channels)
    return out


def synthetic_state_dict(cfg, seed=0, encoder=False):
    """Seeded random decoder weights (bf16-representable, stored as float32) at the published shapes: variance-preserving
    convolutions, norm scales near 1, conv_y near 1 and conv_b near 0 so activations stay O(1) through 40 layers."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    shapes = dict(decoder_param_shapes(cfg))
    if encoder:
        shapes.update(encoder_param_shapes(cfg))
    for name, shape in shapes.items():
        if name.split(".")[-2] in ("norm_layer", "norm1", "norm2", "norm_out") and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.split(".")[-2] in ("norm_layer", "norm1", "norm2", "norm_out"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
            if ".conv_y." in name:
                t = t + 1.0
        else:
            fan_in = math.prod(shape[1:])
            gain = 0.3 if (".conv_y." in name or ".conv_b." in name) else (1.0 if "conv_shortcut" in name or "conv_out" in name else 1.4)
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        sd[name] = t.bfloat16().float()
    return sd


def _causal_conv(x, sd, name, cache):
    w, b = sd[name + ".conv.weight"], sd[name + ".conv.bias"]
    k = w.shape[2]
    if k > 1:
        front = [cache[name]] if name in cache else [x[:, :, :1]] * (k - 1)
        x = torch.cat(front + [x], dim=2)
        cache[name] = x[:, :, -(k - 1):].clone()
    return F.conv3d(x, w, b, padding=(0, w.shape[3] // 2, w.shape[4] // 2))


def _spatial_norm(f, zq, sd, name, cfg, cache):
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=f[:, :, :1].shape[-3:])
        z_rest = F.interpolate(zq[:, :, 1:], size=f[:, :, 1:].shape[-3:])
        zq = torch.cat([z_first, z_rest], dim=2)
    else:
        zq = F.interpolate(zq, size=f.shape[-3:])
    conv_y = _causal_conv(zq, sd, name + ".conv_y", cache)
    conv_b = _causal_conv(zq, sd, name + ".conv_b", cache)
    norm_f = F.group_norm(f, cfg.norm_num_groups, sd[name + ".norm_layer.weight"], sd[name + ".norm_layer.bias"], 1e-6)
    return norm_f * conv_y + conv_b


def _resnet(x, zq, sd, name, cfg, cache):
    h = _spatial_norm(x, zq, sd, name + ".norm1", cfg, cache)
    h = _causal_conv(F.silu(h), sd, name + ".conv1", cache)
    h = _spatial_norm(h, zq, sd, name + ".norm2", cfg, cache)
    h = _causal_conv(F.silu(h), sd, name + ".conv2", cache)
    if name + ".conv_shortcut.weight" in sd:
        x = F.conv3d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
    return h + x


def _upsample(x, sd, name, compress_time):
    if compress_time:
        if x.shape[2] > 1 and x.shape[2] % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
            rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
            x = torch.cat([first, rest], dim=2)
        elif x.shape[2] > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
    else:
        b, c, t, h, w = x.shape
        x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), scale_factor=2.0)
        x = x.reshape(b, t, c, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)
    b, c, t, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), sd[name + ".conv.weight"], sd[name + ".conv.bias"],
                 padding=1)
    return y.reshape(b, t, -1, h, w).permute(0, 2, 1, 3, 4)


def _decoder(z, sd, cfg, cache):
    rev = list(reversed(cfg.block_out_channels))
    h = _causal_conv(z, sd, "decoder.conv_in", cache)
    for j in range(2):
        h = _resnet(h, z, sd, "decoder.mid_block.resnets.%d" % j, cfg, cache)
    levels = int(math.log2(cfg.temporal_compression_ratio))
    for i in range(len(rev)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(h, z, sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), cfg, cache)
        if i != len(rev) - 1:
            h = _upsample(h, sd, "decoder.up_blocks.%d.upsamplers.0" % i, i < levels)
    h = _spatial_norm(h, z, sd, "decoder.norm_out", cfg, cache)
    return _causal_conv(F.silu(h), sd, "decoder.conv_out", cache)


def decode(z, sd, cfg, frame_batch_size=2):
    """AutoencoderKLCogVideoX._decode on float32: z [B, C, L, h, w] -> [B, 3, T, 8h, 8w]."""
    L = z.shape[2]
    num_batches = max(L // frame_batch_size, 1)
    remaining = L % frame_batch_size
    cache, dec = {}, []
    for i in range(num_batches):
        start = frame_batch_size * i + (0 if i == 0 else remaining)
        end = frame_batch_size * (i + 1) + remaining
        dec.append(_decoder(z[:, :, start:end], sd, cfg, cache))
    return torch.cat(dec, dim=2)


def decode_latents(latents, sd, cfg):
    """cog:427-433: latents [B, F, C, h, w] as the sampler returns them."""
    z = latents.permute(0, 2, 1, 3, 4)
    z = 1 / cfg.scaling_factor * z
    return decode(z.float(), sd, cfg)


def _enc_resnet(x, sd, name, cfg, cache):
    h = F.group_norm(x, cfg.norm_num_groups, sd[name + ".norm1.weight"], sd[name + ".norm1.bias"], 1e-6)
    h = _causal_conv(F.silu(h), sd, name + ".conv1", cache)
    h = F.group_norm(h, cfg.norm_num_groups, sd[name + ".norm2.weight"], sd[name + ".norm2.bias"], 1e-6)
    h = _causal_conv(F.silu(h), sd, name + ".conv2", cache)
    if name + ".conv_shortcut.weight" in sd:
        x = F.conv3d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
    return h + x


def _downsample(x, sd, name, compress_time):
    if compress_time:
        b, c, t, h, w = x.shape
        x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
        if t % 2 == 1:
            first, rest = x[..., 0], x[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            x = torch.cat([first[..., None], rest], dim=-1)
        else:
            x = F.avg_pool1d(x, kernel_size=2, stride=2)
        x = x.reshape(b, h, w, c, x.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    b, c, t, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), sd[name + ".conv.weight"], sd[name + ".conv.bias"],
                 stride=2)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def _encoder(x, sd, cfg, cache):
    boc = list(cfg.block_out_channels)
    levels = int(math.log2(cfg.temporal_compression_ratio))
    h = _causal_conv(x, sd, "encoder.conv_in", cache)
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            h = _enc_resnet(h, sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), cfg, cache)
        if i != len(boc) - 1:
            h = _downsample(h, sd, "encoder.down_blocks.%d.downsamplers.0" % i, i < levels)
    for j in range(2):
        h = _enc_resnet(h, sd, "encoder.mid_block.resnets.%d" % j, cfg, cache)
    h = F.group_norm(h, cfg.norm_num_groups, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6)
    return _causal_conv(F.silu(h), sd, "encoder.conv_out", cache)


def encode_moments(x, sd, cfg, frame_batch_size=8):
    """AutoencoderKLCogVideoX._encode on float32: x [B, 3, T, H, W] -> moments [B, 2 * latent, T', H/8, W/8]."""
    T = x.shape[2]
    num_batches = max(T // frame_batch_size, 1)
    remaining = T % frame_batch_size
    cache, enc = {}, []
    for i in range(num_batches):
        start = frame_batch_size * i + (0 if i == 0 else remaining)
        end = frame_batch_size * (i + 1) + remaining
        enc.append(_encoder(x[:, :, start:end], sd, cfg, cache))
    return torch.cat(enc, dim=2)


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution.sample with the noise given: per-op arithmetic in the moments' dtype."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mean + std * noise


def postprocess_uint8(video):
    """VideoProcessor.postprocess_video(output_type="pil") followed by run:121-125, on a [3, T, H, W] tensor in the VAE's
    dtype: (x * 0.5 + 0.5).clamp(0, 1) per op in that dtype, float32, (x * 255).round() -> uint8 [T, H, W, 3] (to_tensor
    and the writer's `* 255` are the identity on uint8 values)."""
    x = (video * 0.5 + 0.5).clamp(0, 1)
    x = x.permute(1, 2, 3, 0).float().numpy()
    return torch.from_numpy((x * 255).round().astype("uint8"))
