"""CogVideoX VAE decoder on MI355X -- the component the reference calls right after the denoising loop
(`pipeline_cogvideox_image2video_lowpass.py:427-433`: `latents = 1 / scaling_factor * latents;
frames = self.vae.decode(latents).sample`), i.e. diffusers' `AutoencoderKLCogVideoX.decode` (third-party, not vendored in
the reference; restated from the published module structure -- see oracle/vae_oracle.py for what is restated and why
parity is unpinned).

MI355X-first formulation (nothing here is a module tree):
  * channels-last bf16 activations over a zero-padded grid, so every 3x3x3 causal convolution and every upsampler Conv2d
    is ONE launch of the ping-pong MFMA GEMM with constant per-tap row offsets (`alg_conv_cl_bf16`): no im2col buffer, no
    per-frame launches, the 2-frame "conv cache" of the published batched decode is simply the two frames in front;
  * the whole video goes through each layer at once (288 GB of HBM: the 49 x 480 x 720 x 256-channel activation is
    8.7 GB); the published decoder works in batches of 2 latent frames to save memory, which only changes GroupNorm (its
    statistics are per batch) -- the stats kernel takes the batch segments, so the numbers are the batched ones;
  * conv_y / conv_b of CogVideoXSpatialNorm3D are 1x1x1, so they run at latent resolution (one small GEMM per norm) and
    the fused GroupNorm * y + b (+ SiLU) kernel reads them through the nearest-neighbour index map.
The HIP extension is mandatory: there is no torch fallback.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib


@dataclass
class AutoencoderKLCogVideoXConfig:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 256, 512)
    latent_channels: int = 16
    layers_per_block: int = 3
    norm_eps: float = 1e-6
    norm_num_groups: int = 32
    temporal_compression_ratio: float = 4
    scaling_factor: float = 0.7          # CogVideoX-5B-I2V vae/config.json (2B: 1.15258426)
    invert_scale_latents: bool = False   # True for CogVideoX 1.5 (cog:395-400)
    use_post_quant_conv: bool = False


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """diffusers' posterior object: moments [B, 2C, ...] -> mean, logvar clamped to [-30, 20], std = exp(0.5 logvar);
    `sample` draws the noise like `randn_tensor` (on the generator's device, then moved), in the moments' dtype."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


class _Level:
    """Geometry of one resolution level for a video of L latent frames."""

    def __init__(self, L, h, w, rate, scale):
        first_lat = L if L < 2 else 2 + L % 2          # latent frames in the first decode batch (cog vae `_decode`)
        self.single = L % 2 == 1                        # an odd first batch keeps its first frame single when doubled
        self.rate, self.scale = rate, scale
        self.T = (1 + (L - 1) * rate) if self.single else L * rate
        self.first_len = (1 + (first_lat - 1) * rate) if self.single else first_lat * rate
        self.seg_len = 2 * rate
        self.H, self.W = h * scale, w * scale
        self.Hp, self.Wp = self.H + 2, self.W + 2
        self.rows = self.Hp * self.Wp


class AutoencoderKLCogVideoX:
    num_latent_frames_batch_size = 2

    def __init__(self, config: Optional[AutoencoderKLCogVideoXConfig] = None, device="cuda", dtype=torch.bfloat16):
        self.config = config or AutoencoderKLCogVideoXConfig()
        c = self.config
        if dtype != torch.bfloat16:
            raise ValueError("the HIP decoder computes in bfloat16")
        if c.norm_num_groups != 32 or c.latent_channels > 64 or c.use_post_quant_conv:
            raise ValueError("unsupported AutoencoderKLCogVideoX configuration")
        for ch in c.block_out_channels:
            if ch < 128 or ch & (ch - 1):
                raise ValueError("block_out_channels must be powers of two >= 128 (32 groups x 16-byte channel chunks)")
        self.device, self.dtype = torch.device(device), dtype
        self.w = {}
        self.profile = None  # optional dict name -> list of (start, end) events

    # ---- weights -------------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda", encoder=False):
        """Seeded random weights at the published shapes (no checkpoint is reachable from this environment)."""
        self = cls(config, device=device)
        c = self.config
        g = torch.Generator().manual_seed(seed)
        sd = {}
        shapes = dict(self.param_shapes())
        if encoder:
            shapes.update(self.encoder_param_shapes())
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
                gain = 0.3 if (".conv_y." in name or ".conv_b." in name) else (
                    1.0 if ("conv_shortcut" in name or "conv_out" in name) else 1.4)
                t = torch.randn(shape, generator=g) * (gain / math.sqrt(math.prod(shape[1:])))
            sd[name] = t.bfloat16()
        self.load_state_dict(sd)
        return self

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", torch_dtype=torch.bfloat16, device="cuda", **_):
        """diffusers-format directory on local disk (`vae/config.json` + safetensors): decoder, and the encoder if present."""
        from .weights import component_from_pretrained
        return component_from_pretrained(cls, AutoencoderKLCogVideoXConfig, path, subfolder, device=device)

    def encoder_param_shapes(self):
        """diffusers names / shapes of the encoder (plain GroupNorm, stride-2 downsamplers, 2 x latent moments out)."""
        c = self.config
        boc = list(c.block_out_channels)
        out = {}

        def conv3(name, ci, co):
            out[name + ".conv.weight"], out[name + ".conv.bias"] = (co, ci, 3, 3, 3), (co,)

        for name, ci, co in self._encoder_resnets():
            out[name + ".norm1.weight"], out[name + ".norm1.bias"] = (ci,), (ci,)
            conv3(name + ".conv1", ci, co)
            out[name + ".norm2.weight"], out[name + ".norm2.bias"] = (co,), (co,)
            conv3(name + ".conv2", co, co)
            if ci != co:
                out[name + ".conv_shortcut.weight"], out[name + ".conv_shortcut.bias"] = (co, ci, 1, 1, 1), (co,)
        conv3("encoder.conv_in", c.in_channels, boc[0])
        for i, ch in enumerate(boc[:-1]):
            out["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (ch, ch, 3, 3)
            out["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (ch,)
        out["encoder.norm_out.weight"], out["encoder.norm_out.bias"] = (boc[-1],), (boc[-1],)
        conv3("encoder.conv_out", boc[-1], 2 * c.latent_channels)
        return out

    def _encoder_resnets(self):
        c = self.config
        boc = list(c.block_out_channels)
        out, prev = [], boc[0]
        for i, ch in enumerate(boc):
            for j in range(c.layers_per_block):
                out.append(("encoder.down_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else ch, ch))
            prev = ch
        return out + [("encoder.mid_block.resnets.%d" % j, boc[-1], boc[-1]) for j in range(2)]

    def param_shapes(self):
        c = self.config
        zc, rev = c.latent_channels, list(reversed(c.block_out_channels))
        out = {}

        def conv3(name, ci, co, k=3):
            out[name + ".conv.weight"], out[name + ".conv.bias"] = (co, ci, k, k, k), (co,)

        def snorm(name, ch):
            out[name + ".norm_layer.weight"], out[name + ".norm_layer.bias"] = (ch,), (ch,)
            conv3(name + ".conv_y", zc, ch, 1)
            conv3(name + ".conv_b", zc, ch, 1)

        for name, ci, co in self._resnets():
            snorm(name + ".norm1", ci)
            conv3(name + ".conv1", ci, co)
            snorm(name + ".norm2", co)
            conv3(name + ".conv2", co, co)
            if ci != co:
                out[name + ".conv_shortcut.weight"], out[name + ".conv_shortcut.bias"] = (co, ci, 1, 1, 1), (co,)
        conv3("decoder.conv_in", zc, rev[0])
        for i, ch in enumerate(rev[:-1]):
            out["decoder.up_blocks.%d.upsamplers.0.conv.weight" % i] = (ch, ch, 3, 3)
            out["decoder.up_blocks.%d.upsamplers.0.conv.bias" % i] = (ch,)
        snorm("decoder.norm_out", rev[-1])
        conv3("decoder.conv_out", rev[-1], c.out_channels)
        return out

    def _resnets(self):
        c = self.config
        rev = list(reversed(c.block_out_channels))
        out = [("decoder.mid_block.resnets.%d" % j, rev[0], rev[0]) for j in range(2)]
        prev = rev[0]
        for i, ch in enumerate(rev):
            for j in range(c.layers_per_block + 1):
                out.append(("decoder.up_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else ch, ch))
            prev = ch
        return out

    def load_state_dict(self, sd, strict=True):
        """diffusers-named decoder tensors -> packed device weights.  Convolution weights become [Cout][taps][Cin]
        (tap-major, channels innermost; the latent's 16 channels are padded to 64, conv_out's 3 outputs to 4), conv_y and
        conv_b of a norm are stacked into one [2C][64] matrix."""
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if missing and strict:
            raise KeyError("missing decoder weights: %s ..." % missing[:3])
        enc_shapes = self.encoder_param_shapes()
        has_encoder = any(k.startswith("encoder.") for k in sd)
        if has_encoder and self.config.layers_per_block < 1:
            raise ValueError("the encoder changes channel width in its resnets: layers_per_block must be >= 1")
        if has_encoder:
            missing = [k for k in enc_shapes if k not in sd]
            if missing:
                raise KeyError("missing encoder weights: %s ..." % missing[:3])
            shapes = dict(shapes, **enc_shapes)
        dev, bf = self.device, torch.bfloat16

        def conv_w(name, cin_pad=None, cout_pad=None, pair_ok=False):
            w = sd[name + ".weight"].to(torch.float32)
            if tuple(w.shape) != tuple(shapes[name + ".weight"]):
                raise ValueError("%s: shape %s, expected %s" % (name, tuple(w.shape), shapes[name + ".weight"]))
            co, ci = w.shape[:2]
            w = w.reshape(co, ci, -1).permute(0, 2, 1)                        # [Cout][taps][Cin]
            if cin_pad and cin_pad > ci:
                w = torch.nn.functional.pad(w, (0, cin_pad - ci))
            b = sd[name + ".bias"].to(torch.float32)
            if cout_pad and cout_pad > co:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cout_pad - co))
                b = torch.nn.functional.pad(b, (0, cout_pad - co))
            w, b = w.reshape(w.shape[0], -1).contiguous().to(dev, bf), b.contiguous().to(dev, bf)
            if pair_ok and w.shape[0] == 128:
                # 128 output channels fill half of the 256-column GEMM tile: pack two neighbouring voxels per GEMM row
                return _lib.pack_conv_pair(w, b, 3) + (True,)
            return w, b, False

        W = {}

        def snorm(name):
            wy, by, _ = conv_w(name + ".conv_y.conv", cin_pad=64)
            wb, bb, _ = conv_w(name + ".conv_b.conv", cin_pad=64)
            W[name] = (sd[name + ".norm_layer.weight"].to(dev, bf).contiguous(),
                       sd[name + ".norm_layer.bias"].to(dev, bf).contiguous(),
                       torch.cat([wy, wb], 0).contiguous(), torch.cat([by, bb], 0).contiguous())

        for name, ci, co in self._resnets():
            snorm(name + ".norm1")
            snorm(name + ".norm2")
            W[name + ".conv1"] = conv_w(name + ".conv1.conv", pair_ok=True)
            W[name + ".conv2"] = conv_w(name + ".conv2.conv", pair_ok=True)
            if ci != co:
                W[name + ".conv_shortcut"] = conv_w(name + ".conv_shortcut")
        W["decoder.conv_in"] = conv_w("decoder.conv_in.conv", cin_pad=64)
        for i in range(len(self.config.block_out_channels) - 1):
            W["decoder.up_blocks.%d.upsamplers.0" % i] = conv_w("decoder.up_blocks.%d.upsamplers.0.conv" % i)
        snorm("decoder.norm_out")
        W["decoder.conv_out"] = conv_w("decoder.conv_out.conv", cout_pad=4)
        if has_encoder:
            for name, ci, co in self._encoder_resnets():
                for n in (".norm1", ".norm2"):
                    W[name + n] = (sd[name + n + ".weight"].to(dev, bf).contiguous(),
                                   sd[name + n + ".bias"].to(dev, bf).contiguous())
                W[name + ".conv1"] = conv_w(name + ".conv1.conv", pair_ok=True)
                W[name + ".conv2"] = conv_w(name + ".conv2.conv", pair_ok=True)
                if ci != co:
                    W[name + ".conv_shortcut"] = conv_w(name + ".conv_shortcut")
            W["encoder.conv_in"] = conv_w("encoder.conv_in.conv", cin_pad=64, pair_ok=True)
            for i in range(len(self.config.block_out_channels) - 1):
                W["encoder.down_blocks.%d.downsamplers.0" % i] = conv_w("encoder.down_blocks.%d.downsamplers.0.conv" % i)
            W["encoder.norm_out"] = (sd["encoder.norm_out.weight"].to(dev, bf).contiguous(),
                                     sd["encoder.norm_out.bias"].to(dev, bf).contiguous())
            W["encoder.conv_out"] = conv_w("encoder.conv_out.conv")
        self.has_encoder = has_encoder
        self.w = W
        return self

    # ---- launches ------------------------------------------------------------------------------------------------
    def _buf(self, rows, ch, slack_rows=0):
        return torch.empty((rows + slack_rows) * ch, device=self.device, dtype=torch.bfloat16)

    def _mark(self, name):
        if self.profile is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.profile.setdefault(name, []).append(ev)
        return ev

    def _geom(self, lv, C, lat):
        return _lib.vae_geom(frames=lv.T, H=lv.H, W=lv.W, C=C, first_len=lv.first_len, seg_len=lv.seg_len,
                             lat_first_single=int(lv.single), lat_rate=lv.rate, lat_scale=lv.scale, lat_h=lat[1],
                             lat_w=lat[2])

    def _spatial_norm(self, x, C, lv, name, zpad, lat):
        """virtual x [T][Hp][Wp][C] -> padded silu(GroupNorm(x) * conv_y(zq) + conv_b(zq)) [T + 2][Hp][Wp][C]; without a
        latent (`zpad` None: the encoder) plain silu(GroupNorm(x))."""
        if zpad is None:
            gamma, beta = self.w[name]
            lat = (1, lv.H, lv.W)
        else:
            gamma, beta, wyb, byb = self.w[name]
            L, h, w = lat
            zrows = (L + 2) * (h + 2) * (w + 2)
            zyb = self._buf(zrows, 2 * C)
            self._mark("zq_gemm")
            _lib.gemm(zpad, wyb, zyb, zrows, 2 * C, 64, 64, 64, 2 * C, bias=byb)
        g = self._geom(lv, C, lat)
        ws = torch.empty(_lib.vae_groupnorm_workspace(g) // 4, device=self.device, dtype=torch.float32)
        nseg = 1 if lv.T <= lv.first_len else 1 + -(-(lv.T - lv.first_len) // lv.seg_len)
        stats = torch.empty(nseg * 64, device=self.device, dtype=torch.float32)
        self._mark("gn_stats")
        _lib.vae_groupnorm_stats(x, g, self.config.norm_eps, ws, stats)
        out = self._buf((lv.T + 2) * lv.rows, C, slack_rows=2 * lv.Wp + 4)
        self._mark("spatial_norm")
        if zpad is None:
            _lib.vae_group_norm(x, stats, gamma, beta, out, g, silu=True)
        else:
            _lib.vae_spatial_norm(x, stats, gamma, beta, zyb, out, g, silu=True)
        return out

    def _conv(self, xpad, name, lv, Cin, Cout, res=None, out=None, kt=3, frames=None):
        w, b, pair = self.w[name]
        frames = lv.T if frames is None else frames
        out = self._buf(frames * lv.rows, Cout) if out is None else out
        self._mark("conv%d_%d" % (Cin, Cout))
        _lib.conv_cl(xpad, w, b, res, out, frames, lv.Hp, lv.Wp, Cin, Cout, kt, pair=pair)
        return out

    def _resnet(self, h, name, ci, co, lv, zpad, lat):
        n = self._spatial_norm(h, ci, lv, name + ".norm1", zpad, lat)
        c1 = self._conv(n, name + ".conv1", lv, ci, co)
        del n
        n = self._spatial_norm(c1, co, lv, name + ".norm2", zpad, lat)
        del c1
        if ci != co:
            w, b, _ = self.w[name + ".conv_shortcut"]
            res = self._buf(lv.T * lv.rows, co)
            self._mark("shortcut")
            _lib.gemm(h, w, res, lv.rows, co, ci, ci, ci, co, bias=b, batch=lv.T, strideA=lv.rows * ci,
                      strideC=lv.rows * co)
        else:
            res = h
        return self._conv(n, name + ".conv2", lv, co, co, res=res, out=res)

    def _decode_one(self, z, c_stride, frame_stride, z_off, L, h, w, scale, to_uint8):
        c = self.config
        rev = list(reversed(c.block_out_channels))
        lat = (L, h, w)
        lv = _Level(L, h, w, 1, 1)
        zpad = self._buf((L + 2) * lv.rows, 64, slack_rows=2 * lv.Wp + 4)
        self._mark("pack")
        _lib.vae_pack_latent(z, c_stride, frame_stride, zpad, L, h, w, c.latent_channels, scale, z_off=z_off)
        hcur = self._conv(zpad, "decoder.conv_in", lv, 64, rev[0])
        resnets = self._resnets()
        for name, ci, co in resnets[:2]:
            hcur = self._resnet(hcur, name, ci, co, lv, zpad, lat)
        levels = int(math.log2(c.temporal_compression_ratio))
        k = 2
        for i, ch in enumerate(rev):
            for _ in range(c.layers_per_block + 1):
                name, ci, co = resnets[k]
                k += 1
                hcur = self._resnet(hcur, name, ci, co, lv, zpad, lat)
            if i != len(rev) - 1:
                compress = i < levels
                nxt = _Level(L, h, w, lv.rate * (2 if compress else 1), lv.scale * 2)
                up = self._buf(nxt.T * nxt.rows, ch, slack_rows=2 * nxt.Wp + 4)
                self._mark("upsample")
                _lib.vae_upsample(hcur, up, nxt.T, lv.H, lv.W, ch, compress, lv.single)
                hcur = self._conv(up, "decoder.up_blocks.%d.upsamplers.0" % i, nxt, ch, ch, kt=1)
                del up
                lv = nxt
        n = self._spatial_norm(hcur, rev[-1], lv, "decoder.norm_out", zpad, lat)
        rgb = self._conv(n, "decoder.conv_out", lv, rev[-1], 4)
        del n
        self._mark("unpack")
        if to_uint8:
            out = torch.empty(lv.T, lv.H, lv.W, 3, device=self.device, dtype=torch.uint8)
        else:
            out = torch.empty(3, lv.T, lv.H, lv.W, device=self.device, dtype=torch.bfloat16)
        _lib.vae_unpack_video(rgb, out, lv.T, lv.H, lv.W, to_uint8)
        self._mark("end")
        return out

    def _encode_one(self, x, x_off, H, W):
        """One image [3][1][H][W] -> moments planes [2 * latent][1][H/8][W/8] (AutoencoderKLCogVideoX._encode, one frame)."""
        c = self.config
        boc = list(c.block_out_channels)
        lv = _Level(1, H, W, 1, 1)
        xpad = self._buf(3 * lv.rows, 64, slack_rows=2 * lv.Wp + 4)
        _lib.vae_pack_latent(x, H * W, H * W, xpad, 1, H, W, c.in_channels, 1.0, z_off=x_off)
        h = self._conv(xpad, "encoder.conv_in", lv, 64, boc[0])
        resnets = self._encoder_resnets()
        k = 0
        for i, ch in enumerate(boc):
            for _ in range(c.layers_per_block):
                name, ci, co = resnets[k]
                k += 1
                h = self._resnet(h, name, ci, co, lv, None, None)
            if i != len(boc) - 1:
                # CogVideoXDownsample3D on one frame: no temporal pooling; pad (0, 1, 0, 1) + Conv2d k3 s2
                pad = self._buf(3 * lv.rows, ch, slack_rows=2 * lv.Wp + 4)
                _lib.vae_pad(h, pad, 1, lv.H, lv.W, ch)
                w, b, _ = self.w["encoder.down_blocks.%d.downsamplers.0" % i]
                m = lv.H // 2 * lv.Wp
                wide = self._buf(m, ch)
                _lib.conv_cl(pad, w, b, None, wide, 1, lv.Hp, lv.Wp, ch, ch, 1, stride2=True, x_off=2 * lv.rows * ch)
                nxt = _Level(1, lv.H // 2, lv.W // 2, 1, 1)
                h = self._buf(nxt.rows, ch)
                _lib.vae_repitch(wide, h, 1, nxt.H, nxt.W, ch, m, lv.Wp)
                lv = nxt
        for name, ci, co in resnets[k:]:
            h = self._resnet(h, name, ci, co, lv, None, None)
        n = self._spatial_norm(h, boc[-1], lv, "encoder.norm_out", None, None)
        mom = self._conv(n, "encoder.conv_out", lv, boc[-1], 2 * c.latent_channels)
        out = torch.empty(2 * c.latent_channels, 1, lv.H, lv.W, device=self.device, dtype=torch.bfloat16)
        _lib.vae_unpack_planes(mom, out, 1, lv.H, lv.W, 2 * c.latent_channels)
        return out

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLCogVideoX.encode` for single frames -- the only use the reference makes of it (cog:388-391 the
        conditioning image, cog:645 the filtered image of the pixel-space ALG branch): x [B, 3, 1, H, W] bf16 in [-1, 1]
        -> `.latent_dist` (DiagonalGaussianDistribution over [B, 16, 1, H/8, W/8])."""
        if not getattr(self, "has_encoder", False):
            raise _lib.AlgHipError("this AutoencoderKLCogVideoX was loaded without encoder weights")
        if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 5 and x.is_contiguous()):
            raise _lib.AlgHipError("AutoencoderKLCogVideoX.encode: a contiguous 5-D bfloat16 device tensor is required "
                                   "(HIP-only path)")
        B, C, T, H, W = x.shape
        down = 2 ** (len(self.config.block_out_channels) - 1)
        if T != 1 or C != self.config.in_channels or H % down or W % down:
            raise ValueError("encode() takes single frames [B, %d, 1, H, W] with H, W multiples of %d (video encoding -- "
                             "temporal pooling, 8-frame batches -- is not used by the ALG pipelines and not built)"
                             % (self.config.in_channels, down))
        mom = torch.stack([self._encode_one(x, b * C * H * W, H, W) for b in range(B)])
        dist = DiagonalGaussianDistribution(mom)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def _run(self, z, scale, to_uint8, layout):
        if not (z.is_cuda and z.dtype == torch.bfloat16 and z.dim() == 5 and z.is_contiguous()):
            raise _lib.AlgHipError("AutoencoderKLCogVideoX.decode: a contiguous 5-D bfloat16 device tensor is required "
                                   "(HIP-only path)")
        if layout == "BCTHW":
            B, C, L, h, w = z.shape
            cs, fs = L * h * w, h * w
        else:                               # "BTCHW": the sampler's own layout (cog:427-429 permutes it)
            B, L, C, h, w = z.shape
            cs, fs = h * w, C * h * w
        if C != self.config.latent_channels:
            raise ValueError("latent has %d channels, the VAE expects %d" % (C, self.config.latent_channels))
        return torch.stack([self._decode_one(z, cs, fs, b * C * L * h * w, L, h, w, scale, to_uint8)
                            for b in range(B)])

    # ---- public surface (diffusers names) -------------------------------------------------------------------------
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLCogVideoX.decode`: z [B, 16, L, h, w] -> sample [B, 3, T, 8h, 8w]."""
        out = self._run(z, 1.0, False, "BCTHW")
        return DecoderOutput(sample=out) if return_dict else (out,)

    def decode_latents(self, latents: torch.Tensor, to_uint8: bool = False):
        """cog:427-433 in one pass: sampler latents [B, F, 16, h, w] -> frames [B, 3, T, H, W] (or, with `to_uint8`, the
        writer's uint8 [B, T, H, W, 3]: postprocess_video + run:121-125 fused into the last kernel); the
        1 / scaling_factor product is applied while packing (one bf16 rounding, as the tensor op)."""
        c = self.config
        return self._run(latents, float(torch.tensor(1 / c.scaling_factor, dtype=torch.float32)), to_uint8, "BTCHW")
