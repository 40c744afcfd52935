"""CPU restatement of diffusers' `AutoencoderKLWan` (the component `pipeline_wan_image2video_lowpass.py:426-430` encodes
the condition video with and `:959` decodes the final latents with; third-party, diffusers @ be2fb77, NOT in the reference
tree -- **parity unpinned**: restated from the published module structure).

TEST INFRASTRUCTURE -- never imported by ``alg_amd``.

What is restated, in the PUBLISHED CHUNKED FORM (so that the whole-video HIP formulation is checked against it):
  * `WanCausalConv3d`: zero padding in time (2 * pad frames in front), or the cached last frames of the previous chunk;
  * `WanRMS_norm`: F.normalize over channels * sqrt(C) * gamma;  `WanResidualBlock`;  `WanAttentionBlock` (per frame,
    one head of width C);  `WanResample` in its four modes with the feature cache ("Rep" marker on the first chunk of an
    upsample3d: the first latent frame is not doubled and never enters the temporal convolution);
  * `_encode`: chunks of 1, 4, 4, ... frames;  `_decode`: one latent frame per chunk;  clamp to [-1, 1].
Weights use the diffusers state-dict names.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F

CACHE_T = 2


@dataclass
class WanVAEConfig:
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    temperal_downsample: List[bool] = field(default_factory=lambda: [False, True, True])
    latents_mean: List[float] = field(default_factory=lambda: [
        -0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
        -0.9497, 0.2503, -0.2921])
    latents_std: List[float] = field(default_factory=lambda: [
        2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253,
        2.8251, 1.9160])


def encoder_plan(cfg):
    """[(kind, name, in_dim, out_dim)] of `WanEncoder3d.down_blocks` (a flat ModuleList in the published module)."""
    dims = [cfg.base_dim * u for u in [1] + list(cfg.dim_mult)]
    plan, k = [], 0
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            plan.append(("res", "encoder.down_blocks.%d" % k, ci, co))
            ci = co
            k += 1
        if i != len(cfg.dim_mult) - 1:
            plan.append(("downsample3d" if cfg.temperal_downsample[i] else "downsample2d", "encoder.down_blocks.%d" % k, co, co))
            k += 1
    return plan, dims[-1]


def decoder_plan(cfg):
    """[(kind, name, in_dim, out_dim)] of `WanDecoder3d.up_blocks` (WanUpBlock: resnets + optional upsampler)."""
    dims = [cfg.base_dim * u for u in [cfg.dim_mult[-1]] + list(cfg.dim_mult[::-1])]
    up = list(cfg.temperal_downsample[::-1])
    plan = []
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            ci = ci // 2
        for j in range(cfg.num_res_blocks + 1):
            plan.append(("res", "decoder.up_blocks.%d.resnets.%d" % (i, j), ci, co))
            ci = co
        if i != len(cfg.dim_mult) - 1:
            plan.append(("upsample3d" if up[i] else "upsample2d", "decoder.up_blocks.%d.upsamplers.0" % i, co, co // 2))
    return plan, dims[0], dims[-1]


def param_shapes(cfg):
    s = {}

    def conv3(name, ci, co, k=(3, 3, 3)):
        s[name + ".weight"], s[name + ".bias"] = (co, ci) + tuple(k), (co,)

    def res(name, ci, co):
        s[name + ".norm1.gamma"] = (ci, 1, 1, 1)
        conv3(name + ".conv1", ci, co)
        s[name + ".norm2.gamma"] = (co, 1, 1, 1)
        conv3(name + ".conv2", co, co)
        if ci != co:
            conv3(name + ".conv_shortcut", ci, co, (1, 1, 1))

    def mid(prefix, dim):
        res(prefix + ".resnets.0", dim, dim)
        s[prefix + ".attentions.0.norm.gamma"] = (dim, 1, 1)
        s[prefix + ".attentions.0.to_qkv.weight"], s[prefix + ".attentions.0.to_qkv.bias"] = (3 * dim, dim, 1, 1), (3 * dim,)
        s[prefix + ".attentions.0.proj.weight"], s[prefix + ".attentions.0.proj.bias"] = (dim, dim, 1, 1), (dim,)
        res(prefix + ".resnets.1", dim, dim)

    plan, top = encoder_plan(cfg)
    conv3("encoder.conv_in", 3, cfg.base_dim)
    for kind, name, ci, co in plan:
        if kind == "res":
            res(name, ci, co)
        else:
            s[name + ".resample.1.weight"], s[name + ".resample.1.bias"] = (co, ci, 3, 3), (co,)
            if kind == "downsample3d":
                conv3(name + ".time_conv", ci, ci, (3, 1, 1))
    mid("encoder.mid_block", top)
    s["encoder.norm_out.gamma"] = (top, 1, 1, 1)
    conv3("encoder.conv_out", top, 2 * cfg.z_dim)
    conv3("quant_conv", 2 * cfg.z_dim, 2 * cfg.z_dim, (1, 1, 1))
    conv3("post_quant_conv", cfg.z_dim, cfg.z_dim, (1, 1, 1))
    plan, top, last = decoder_plan(cfg)
    conv3("decoder.conv_in", cfg.z_dim, top)
    mid("decoder.mid_block", top)
    for kind, name, ci, co in plan:
        if kind == "res":
            res(name, ci, co)
        else:
            s[name + ".resample.1.weight"], s[name + ".resample.1.bias"] = (co, ci, 3, 3), (co,)
            if kind == "upsample3d":
                conv3(name + ".time_conv", ci, 2 * ci, (3, 1, 1))
    s["decoder.norm_out.gamma"] = (last, 1, 1, 1)
    conv3("decoder.conv_out", last, 3)
    return s


def init_weights(cfg, seed=0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".gamma"):
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
def causal_conv(x, w, b, pad, stride=(1, 1, 1), cache_x=None):
    """WanCausalConv3d.forward: pad = the module's (t, h, w) padding; time padding 2 * pad_t in front (minus the cache)."""
    pt, ph, pw = pad
    front = 2 * pt
    if cache_x is not None and front > 0:
        x = torch.cat([cache_x, x], dim=2)
        front -= cache_x.shape[2]
    x = F.pad(x, (pw, pw, ph, ph, front, 0))
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x, gamma, dim=1):
    c = x.shape[dim]
    return F.normalize(x, dim=dim) * (c ** 0.5) * gamma


class _Cache:
    """feat_cache / feat_idx of the published forward passes."""

    def __init__(self):
        self.map, self.idx = {}, 0

    def conv(self, x, w, b, pad):
        """A cached causal convolution as WanResidualBlock / WanEncoder3d / WanDecoder3d call it."""
        i = self.idx
        cache_x = x[:, :, -CACHE_T:].clone()
        prev = self.map.get(i)
        if cache_x.shape[2] < 2 and prev is not None:
            cache_x = torch.cat([prev[:, :, -1:], cache_x], dim=2)
        y = causal_conv(x, w, b, pad, cache_x=prev)
        self.map[i] = cache_x
        self.idx += 1
        return y


def res_block(x, sd, name, c: _Cache):
    h = x
    if name + ".conv_shortcut.weight" in sd:
        h = causal_conv(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"], (0, 0, 0))
    x = F.silu(rms_norm(x, sd[name + ".norm1.gamma"]))
    x = c.conv(x, sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], (1, 1, 1))
    x = F.silu(rms_norm(x, sd[name + ".norm2.gamma"]))
    x = c.conv(x, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], (1, 1, 1))
    return x + h


def attention_block(x, sd, name):
    ident = x
    b, ch, t, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, ch, h, w)
    x = rms_norm(x, sd[name + ".norm.gamma"])
    qkv = F.conv2d(x, sd[name + ".to_qkv.weight"], sd[name + ".to_qkv.bias"])
    qkv = qkv.reshape(b * t, 1, ch * 3, -1).permute(0, 1, 3, 2).contiguous()
    q, k, v = qkv.chunk(3, dim=-1)
    x = F.scaled_dot_product_attention(q, k, v)
    x = x.squeeze(1).permute(0, 2, 1).reshape(b * t, ch, h, w)
    x = F.conv2d(x, sd[name + ".proj.weight"], sd[name + ".proj.bias"])
    x = x.view(b, t, ch, h, w).permute(0, 2, 1, 3, 4)
    return x + ident


def mid_block(x, sd, prefix, c):
    x = res_block(x, sd, prefix + ".resnets.0", c)
    x = attention_block(x, sd, prefix + ".attentions.0")
    return res_block(x, sd, prefix + ".resnets.1", c)


def resample(x, sd, name, mode, c: _Cache):
    b, ch, t, h, w = x.shape
    if mode == "upsample3d":
        i = c.idx
        if c.map.get(i) is None:
            c.map[i] = "Rep"
            c.idx += 1
        else:
            cache_x = x[:, :, -CACHE_T:].clone()
            prev = c.map[i]
            if cache_x.shape[2] < 2 and not isinstance(prev, str):
                cache_x = torch.cat([prev[:, :, -1:], cache_x], dim=2)
            if cache_x.shape[2] < 2 and isinstance(prev, str):
                cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
            tw, tb = sd[name + ".time_conv.weight"], sd[name + ".time_conv.bias"]
            x = causal_conv(x, tw, tb, (1, 0, 0), cache_x=None if isinstance(prev, str) else prev)
            c.map[i] = cache_x
            c.idx += 1
            x = x.reshape(b, 2, ch, t, h, w)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, ch, t * 2, h, w)
    t = x.shape[2]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, ch, h, w)
    rw, rb = sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"]
    if mode.startswith("upsample"):
        x = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(x)
        x = F.conv2d(x, rw, rb, padding=1)
    else:
        x = F.conv2d(F.pad(x, (0, 1, 0, 1)), rw, rb, stride=2)
    x = x.view(b, t, x.size(1), x.size(2), x.size(3)).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d":
        i = c.idx
        if c.map.get(i) is None:
            c.map[i] = x.clone()
            c.idx += 1
        else:
            cache_x = x[:, :, -1:].clone()
            x = causal_conv(torch.cat([c.map[i][:, :, -1:], x], 2), sd[name + ".time_conv.weight"],
                            sd[name + ".time_conv.bias"], (0, 0, 0), stride=(2, 1, 1))
            c.map[i] = cache_x
            c.idx += 1
    return x


def encoder_chunk(cfg, sd, x, c: _Cache):
    plan, top = encoder_plan(cfg)
    x = c.conv(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], (1, 1, 1))
    for kind, name, ci, co in plan:
        x = res_block(x, sd, name, c) if kind == "res" else resample(x, sd, name, kind, c)
    x = mid_block(x, sd, "encoder.mid_block", c)
    x = F.silu(rms_norm(x, sd["encoder.norm_out.gamma"]))
    return c.conv(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], (1, 1, 1))


def decoder_chunk(cfg, sd, x, c: _Cache):
    plan, top, last = decoder_plan(cfg)
    x = c.conv(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], (1, 1, 1))
    x = mid_block(x, sd, "decoder.mid_block", c)
    for kind, name, ci, co in plan:
        x = res_block(x, sd, name, c) if kind == "res" else resample(x, sd, name, kind, c)
    x = F.silu(rms_norm(x, sd["decoder.norm_out.gamma"]))
    return c.conv(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], (1, 1, 1))


def encode(cfg, sd, x):
    """AutoencoderKLWan._encode: x [B, 3, T, H, W] (T = 4k + 1) -> moments [B, 2 z, 1 + (T - 1) / 4, H / 8, W / 8]."""
    t = x.shape[2]
    cache = _Cache()
    out = None
    for i in range(1 + (t - 1) // 4):
        cache.idx = 0
        piece = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        o = encoder_chunk(cfg, sd, piece, cache)
        out = o if out is None else torch.cat([out, o], 2)
    return causal_conv(out, sd["quant_conv.weight"], sd["quant_conv.bias"], (0, 0, 0))


def decode(cfg, sd, z):
    """AutoencoderKLWan._decode: z [B, z, L, h, w] -> [B, 3, 4 (L - 1) + 1, 8 h, 8 w], clamped to [-1, 1]."""
    x = causal_conv(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"], (0, 0, 0))
    cache = _Cache()
    out = None
    for i in range(z.shape[2]):
        cache.idx = 0
        o = decoder_chunk(cfg, sd, x[:, :, i:i + 1], cache)
        out = o if out is None else torch.cat([out, o], 2)
    return torch.clamp(out, min=-1.0, max=1.0)
