"""HunyuanVideo VAE on MI355X -- the component the reference's HunyuanVideo pipeline encodes the conditioning image with
(`pipeline_hunyuan_video_image2video_lowpass.py:578-582`: `retrieve_latents(self.vae.encode(img.unsqueeze(0)), generator,
"argmax")`) and decodes the final latents with (`:1291-1292`), i.e. diffusers' `AutoencoderKLHunyuanVideo` (third-party,
not vendored in the reference; restated from the published module structure -- see oracle/hunyuan_vae_oracle.py for what
is restated and why parity is unpinned).

MI355X-first formulation (shares the machinery of the CogVideoX and Wan VAEs):
  * activations are channels-last bf16 in the implicit-GEMM convolution's layouts; every 3x3x3 `HunyuanVideoCausalConv3d`
    is ONE `alg_conv_cl_bf16` launch over a border-REPLICATED grid (the published module pads with mode="replicate": two
    copies of the first frame in front -- which `alg_vae_group_norm` / `alg_vae_pad` already write -- and one replicated
    row / column per side, four strided byte copies);
  * GroupNorm(32) + SiLU is `alg_vae_groupnorm_stats` + `alg_vae_group_norm` (deterministic two-level reduction, one
    segment = the whole tile);
  * strided downsampler convolutions are the stride-1 launch sub-sampled (the stride applies to the padded tensor, so
    output (t, y, x) is the stride-1 output at (2t, 2y, 2x)); only the encoder has them and the reference encodes ONE
    image, so the 4-8x surplus is a few ms;
  * the mid-block attention (one head of width 512 over all T * H * W tokens, block-causal over frames) runs frame by
    frame: the queries of frame f against the keys of frames 0..f -- the masked blocks are never computed -- as two score
    GEMMs that together hold q k^T to 2^-17 (hi + lo bf16 parts), the fp32 row softmax `alg_softmax_hilo` and the P V GEMM;
  * decode follows the published defaults the reference runs with (framewise decoding on, spatial tiling off): tiles of
    5 latent frames every 3, first decoded frame of later tiles dropped, 4-frame cross-fade (`alg_lincomb`).
PyTorch moves bytes between layouts (borders, nearest-neighbour duplication, sub-sampling); arithmetic is HIP.  There is
no torch fallback.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from .autoencoder_kl_cogvideox import AutoencoderKLOutput, DecoderOutput, DiagonalGaussianDistribution

BF = torch.bfloat16


@dataclass
class AutoencoderKLHunyuanVideoConfig:
    """Defaults = hunyuanvideo-community/HunyuanVideo-I2V vae/config.json."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    down_block_types: List[str] = field(default_factory=lambda: ["HunyuanVideoDownBlock3D"] * 4)
    up_block_types: List[str] = field(default_factory=lambda: ["HunyuanVideoUpBlock3D"] * 4)
    block_out_channels: List[int] = field(default_factory=lambda: [128, 256, 512, 512])
    layers_per_block: int = 2
    act_fn: str = "silu"
    norm_num_groups: int = 32
    scaling_factor: float = 0.476986
    spatial_compression_ratio: int = 8
    temporal_compression_ratio: int = 4
    mid_block_add_attention: bool = True


class _Act:
    """A video in the 'virtual' layout the convolution writes: flat bf16 [T][H + 2][W + 2][C], valid for y < H, x < W."""

    def __init__(self, buf, T, H, W, C):
        self.buf, self.T, self.H, self.W, self.C = buf, T, H, W, C
        self.Hp, self.Wp = H + 2, W + 2
        self.rows = self.Hp * self.Wp

    def view(self):
        return self.buf[: self.T * self.rows * self.C].view(self.T, self.Hp, self.Wp, self.C)

    def valid(self):
        return self.view()[:, : self.H, : self.W]


class AutoencoderKLHunyuanVideo:
    # tiling constants of the published class (attributes there too, not config entries)
    tile_sample_min_num_frames = 16
    tile_sample_stride_num_frames = 12

    def __init__(self, config: Optional[AutoencoderKLHunyuanVideoConfig] = None, device="cuda", dtype=BF):
        self.config = config or AutoencoderKLHunyuanVideoConfig()
        c = self.config
        if dtype != BF:
            raise ValueError("the HIP VAE computes in bfloat16")
        if c.norm_num_groups != 32 or c.act_fn != "silu":
            raise NotImplementedError("GroupNorm(32) + SiLU is what every published HunyuanVideo VAE config uses")
        if c.temporal_compression_ratio != 4 or c.spatial_compression_ratio != 8:
            raise NotImplementedError("only the published 4 x 8 x 8 compression is built")
        for ch in c.block_out_channels:
            if ch < 128 or ch & (ch - 1):
                raise NotImplementedError("block widths must be powers of two >= 128 (GroupNorm / convolution kernels)")
        self.spatial_compression_ratio = c.spatial_compression_ratio     # the pipeline reads these (hy:278-279)
        self.temporal_compression_ratio = c.temporal_compression_ratio
        self.use_framewise_decoding = True
        self.device, self.dtype = torch.device(device), dtype
        if self.device.type != "cuda":
            raise _lib.AlgHipError("AutoencoderKLHunyuanVideo runs on the GPU only (HIP kernels, no CPU fallback)")
        _lib.load_library()
        self.w = {}

    # ---- structure (diffusers module order and names) ---------------------------------------------------------------
    def _plan(self, reverse):
        """[(block, in, out, stride / factor or None)]: spatial resampling in blocks 0-2, temporal in blocks 1-2."""
        boc = list(self.config.block_out_channels)
        boc = boc[::-1] if reverse else boc
        n, plan, ci = len(boc), [], boc[0]
        for i, co in enumerate(boc):
            f = None
            if i < n - 1:
                f = (2 if i >= n - 3 else 1, 2 if i < 3 else 1, 2 if i < 3 else 1)
            plan.append((i, ci, co, f))
            ci = co
        return plan

    def param_shapes(self):
        c = self.config
        s = {}

        def conv(name, ci, co, k=3):
            s[name + ".conv.weight"], s[name + ".conv.bias"] = (co, ci, k, k, k), (co,)

        def norm(name, ch):
            s[name + ".weight"], s[name + ".bias"] = (ch,), (ch,)

        def res(name, ci, co):
            norm(name + ".norm1", ci)
            conv(name + ".conv1", ci, co)
            norm(name + ".norm2", co)
            conv(name + ".conv2", co, co)
            if ci != co:
                conv(name + ".conv_shortcut", ci, co, 1)

        def mid(prefix, ch):
            res(prefix + ".resnets.0", ch, ch)
            if c.mid_block_add_attention:
                a = prefix + ".attentions.0"
                norm(a + ".group_norm", ch)
                for p in ("to_q", "to_k", "to_v", "to_out.0"):
                    s["%s.%s.weight" % (a, p)], s["%s.%s.bias" % (a, p)] = (ch, ch), (ch,)
            res(prefix + ".resnets.1", ch, ch)

        boc = list(c.block_out_channels)
        conv("encoder.conv_in", c.in_channels, boc[0])
        for i, ci, co, stride in self._plan(False):
            for j in range(c.layers_per_block):
                res("encoder.down_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if stride is not None:
                conv("encoder.down_blocks.%d.downsamplers.0.conv" % i, co, co)
        mid("encoder.mid_block", boc[-1])
        norm("encoder.conv_norm_out", boc[-1])
        conv("encoder.conv_out", boc[-1], 2 * c.latent_channels)
        s["quant_conv.weight"], s["quant_conv.bias"] = (2 * c.latent_channels,) * 2 + (1, 1, 1), (2 * c.latent_channels,)
        s["post_quant_conv.weight"], s["post_quant_conv.bias"] = (c.latent_channels,) * 2 + (1, 1, 1), (c.latent_channels,)
        conv("decoder.conv_in", c.latent_channels, boc[-1])
        mid("decoder.mid_block", boc[-1])
        for i, ci, co, factor in self._plan(True):
            for j in range(c.layers_per_block + 1):
                res("decoder.up_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if factor is not None:
                conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, co, co)
        norm("decoder.conv_norm_out", boc[0])
        conv("decoder.conv_out", boc[0], c.out_channels)
        return s

    # ---- weights ---------------------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda"):
        self = cls(config, device=device)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.param_shapes().items():
            if len(shape) == 1 and name.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif name.endswith(".bias"):
                t = 0.05 * torch.randn(shape, generator=g)
            else:
                fan = 1
                for d in shape[1:]:
                    fan *= d
                t = torch.randn(shape, generator=g) * (1.2 / fan ** 0.5)
            sd[name] = t.to(BF)
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", torch_dtype=BF, device="cuda", **_):
        """diffusers-format directory on local disk (`vae/config.json` + safetensors)."""
        from .weights import component_from_pretrained
        return component_from_pretrained(cls, AutoencoderKLHunyuanVideoConfig, path, subfolder, device=device)

    def load_state_dict(self, sd, strict=True):
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if missing and strict:
            raise KeyError("missing HunyuanVideo VAE weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        dev = self.device
        W = {}
        pad = torch.nn.functional.pad

        def conv_w(name, pair_ok=False, cout_pad=None):
            """[Cout][Cin][taps...] -> ([Cout_p][taps * Cin_p] bf16, [Cout_p] bias, pair flag): tap-major, channels innermost."""
            w = sd[name + ".weight"].float()
            co, ci = w.shape[:2]
            cip, cop = max(ci, 64), cout_pad or max(co, 64)
            w = pad(w.reshape(co, ci, -1).permute(0, 2, 1), (0, cip - ci, 0, 0, 0, cop - co))
            b = pad(sd[name + ".bias"].float(), (0, cop - co))
            w, b = w.reshape(cop, -1).contiguous().to(dev, BF), b.contiguous().to(dev, BF)
            if pair_ok and cop == 128 and w.shape[1] // cip == 27:
                return _lib.pack_conv_pair(w, b, 3) + (True,)
            return w, b, False

        def norm(name):
            return (sd[name + ".weight"].to(dev, BF).contiguous(), sd[name + ".bias"].to(dev, BF).contiguous())

        def res(name, ci, co):
            W[name + ".norm1"], W[name + ".norm2"] = norm(name + ".norm1"), norm(name + ".norm2")
            W[name + ".conv1"] = conv_w(name + ".conv1.conv", pair_ok=True)
            W[name + ".conv2"] = conv_w(name + ".conv2.conv", pair_ok=True)
            if ci != co:
                W[name + ".conv_shortcut"] = conv_w(name + ".conv_shortcut.conv")[:2]

        def mid(prefix, ch):
            res(prefix + ".resnets.0", ch, ch)
            if self.config.mid_block_add_attention:
                a = prefix + ".attentions.0"
                W[a + ".group_norm"] = norm(a + ".group_norm")
                wq, wk = sd[a + ".to_q.weight"].float(), sd[a + ".to_k.weight"].float()
                bq, bk = sd[a + ".to_q.bias"].float(), sd[a + ".to_k.bias"].float()
                # [q | -q | k] columns of one GEMM (the negated copy feeds the first score GEMM: see _attention)
                W[a + ".qk"] = (torch.cat([wq, -wq, wk]).contiguous().to(dev, BF), torch.cat([bq, -bq, bk]).contiguous().to(dev, BF))
                W[a + ".v"] = (sd[a + ".to_v.weight"].to(dev, BF).contiguous(), sd[a + ".to_v.bias"].to(dev, BF).contiguous())
                W[a + ".out"] = (sd[a + ".to_out.0.weight"].to(dev, BF).contiguous(), sd[a + ".to_out.0.bias"].to(dev, BF).contiguous())
            res(prefix + ".resnets.1", ch, ch)

        c = self.config
        boc = list(c.block_out_channels)
        W["encoder.conv_in"] = conv_w("encoder.conv_in.conv", pair_ok=True)
        for i, ci, co, stride in self._plan(False):
            for j in range(c.layers_per_block):
                res("encoder.down_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if stride is not None:
                W["encoder.down_blocks.%d.downsamplers.0" % i] = conv_w("encoder.down_blocks.%d.downsamplers.0.conv.conv" % i,
                                                                        pair_ok=True)
        mid("encoder.mid_block", boc[-1])
        W["encoder.conv_norm_out"] = norm("encoder.conv_norm_out")
        W["encoder.conv_out"] = conv_w("encoder.conv_out.conv")
        W["quant_conv"] = conv_w("quant_conv")[:2]
        W["post_quant_conv"] = conv_w("post_quant_conv")[:2]
        W["decoder.conv_in"] = conv_w("decoder.conv_in.conv")
        mid("decoder.mid_block", boc[-1])
        for i, ci, co, factor in self._plan(True):
            for j in range(c.layers_per_block + 1):
                res("decoder.up_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if factor is not None:
                W["decoder.up_blocks.%d.upsamplers.0" % i] = conv_w("decoder.up_blocks.%d.upsamplers.0.conv.conv" % i, pair_ok=True)
        W["decoder.conv_norm_out"] = norm("decoder.conv_norm_out")
        W["decoder.conv_out"] = conv_w("decoder.conv_out.conv", cout_pad=4)
        self.w = W
        return self

    # ---- layout plumbing (bytes only) ----------------------------------------------------------------------------------
    def _empty(self, n):
        return torch.empty(n, device=self.device, dtype=BF)

    def _pbuf(self, T, H, W, C):
        """Convolution input [T + 2][H + 2][W + 2][C] + the slack rows the implicit GEMM may touch."""
        return self._empty(((T + 2) * (H + 2) * (W + 2) + 2 * (W + 2) + 4) * C)

    @staticmethod
    def _replicate_borders(buf, T, H, W, C):
        """mode="replicate" in space on a padded buffer whose interior and leading frames are written."""
        v = buf[: (T + 2) * (H + 2) * (W + 2) * C].view(T + 2, H + 2, W + 2, C)
        v[:, 0] = v[:, 1]
        v[:, H + 1] = v[:, H]
        v[:, :, 0] = v[:, :, 1]
        v[:, :, W + 1] = v[:, :, W]
        return buf

    def _padded(self, a: _Act):
        """virtual -> replicate-padded, no arithmetic (`alg_vae_pad` repeats the first frame in front)."""
        p = self._pbuf(a.T, a.H, a.W, a.C)
        _lib.vae_pad(a.buf, p, a.T, a.H, a.W, a.C)
        return self._replicate_borders(p, a.T, a.H, a.W, a.C)

    def _padded_from(self, src, T, H, W, C):
        """`src` [T, H, W, C] (any strides) -> replicate-padded convolution input."""
        p = self._pbuf(T, H, W, C)
        v = p[: (T + 2) * (H + 2) * (W + 2) * C].view(T + 2, H + 2, W + 2, C)
        v[2:, 1:H + 1, 1:W + 1] = src
        v[0, 1:H + 1, 1:W + 1] = src[0]
        v[1, 1:H + 1, 1:W + 1] = src[0]
        return self._replicate_borders(p, T, H, W, C)

    # ---- launches --------------------------------------------------------------------------------------------------------
    def _geom(self, a: _Act):
        return _lib.vae_geom(frames=a.T, H=a.H, W=a.W, C=a.C, first_len=a.T, seg_len=a.T, lat_first_single=0, lat_rate=1,
                             lat_scale=1, lat_h=a.H, lat_w=a.W)

    def _norm(self, a: _Act, name, silu=True):
        """virtual -> replicate-padded act(GroupNorm(x)); statistics over the whole (T, H, W) extent of each group."""
        gamma, beta = self.w[name]
        g = self._geom(a)
        ws = torch.empty(_lib.vae_groupnorm_workspace(g) // 4, device=self.device, dtype=torch.float32)
        stats = torch.empty(64, device=self.device, dtype=torch.float32)
        _lib.vae_groupnorm_stats(a.buf, g, 1e-6, ws, stats)
        p = self._pbuf(a.T, a.H, a.W, a.C)
        _lib.vae_group_norm(a.buf, stats, gamma, beta, p, g, silu=silu)
        return p

    def _conv(self, pbuf, name, T, H, W, Cin, Cout, res=None, cout_pad=None):
        w, b, pair = self.w[name]
        cip, cop = max(Cin, 64), cout_pad or max(Cout, 64)
        if pair and ((H + 2) * (W + 2)) % 2:
            raise _lib.AlgHipError("internal: two-voxel packing needs an even padded plane")
        out = res.buf if res is not None else self._empty(T * (H + 2) * (W + 2) * cop)
        _lib.conv_cl(pbuf, w, b, None if res is None else res.buf, out, T, H + 2, W + 2, cip, cop, 3, pair=pair)
        return _Act(out, T, H, W, cop) if cout_pad is None else out

    def _res_block(self, x: _Act, name, ci, co):
        p = self._replicate_borders(self._norm(x, name + ".norm1"), x.T, x.H, x.W, ci)
        c1 = self._conv(p, name + ".conv1", x.T, x.H, x.W, ci, co)
        del p
        p = self._replicate_borders(self._norm(c1, name + ".norm2"), x.T, x.H, x.W, co)
        del c1
        if ci != co:
            w, b = self.w[name + ".conv_shortcut"]
            h = _Act(self._empty(x.T * x.rows * co), x.T, x.H, x.W, co)
            _lib.gemm(x.buf, w, h.buf, x.T * x.rows, co, ci, ci, ci, co, bias=b)
        else:
            h = x
        return self._conv(p, name + ".conv2", x.T, x.H, x.W, co, co, res=h)

    def _attention(self, x: _Act, name):
        """`Attention(heads=1, dim_head=C, norm_num_groups=32, residual_connection=True)` over the (t, h, w) tokens with the
        block-causal frame mask of `prepare_causal_attention_mask`: x + to_out(softmax(q k^T / sqrt(C) + mask) v)."""
        T, n, C = x.T, x.H * x.W, x.C
        dense = x.valid().reshape(T * n, C).contiguous()
        p = self._norm(x, name + ".group_norm", silu=False)
        xn = p[: (T + 2) * x.rows * C].view(T + 2, x.Hp, x.Wp, C)[2:, 1:x.H + 1, 1:x.W + 1].reshape(T * n, C).contiguous()
        del p
        wqk, bqk = self.w[name + ".qk"]
        qk = self._empty(T * n * 3 * C)                                # columns: [q | -q | k]
        _lib.gemm(xn, wqk, qk, T * n, 3 * C, C, C, C, 3 * C, bias=bqk)
        ldv = (T * n + 63) // 64 * 64
        wv, bv = self.w[name + ".v"]
        vt = torch.zeros(C * ldv, device=self.device, dtype=BF)        # V^T, written by a GEMM with swapped operands
        _lib.gemm(wv, xn, vt, C, T * n, C, C, C, ldv, bias=bv, flags=_lib.GEMM_BIAS_PER_ROW)
        del xn
        o = self._empty(T * n * C)
        for f in range(T):
            keys = (f + 1) * n                                         # frames 0..f are visible to frame f
            ld = (keys + 63) // 64 * 64
            neg_hi, lo, prob = self._empty(n * ld), self._empty(n * ld), self._empty(n * ld)
            # neg_hi = bf16((-q) k^T);  lo = bf16(q k^T + neg_hi): together the fp32 scores to ~2^-17
            _lib.gemm(qk, qk, neg_hi, n, keys, C, 3 * C, 3 * C, ld, a_off=f * n * 3 * C + C, b_off=2 * C)
            _lib.gemm(qk, qk, lo, n, keys, C, 3 * C, 3 * C, ld, R=neg_hi, ldr=ld, a_off=f * n * 3 * C, b_off=2 * C)
            _lib.softmax_hilo(neg_hi, lo, prob, n, keys, ld, float(C) ** -0.5)
            del neg_hi, lo
            _lib.gemm(prob, vt, o, n, C, ld, ld, ldv, C, c_off=f * n * C)
            del prob
        del qk, vt
        wo, bo = self.w[name + ".out"]
        out = self._empty(T * n * C)
        _lib.gemm(o, wo, out, T * n, C, C, C, C, C, bias=bo, R=dense, ldr=C)
        y = _Act(self._empty(T * x.rows * C), T, x.H, x.W, C)
        y.valid().copy_(out.view(T, x.H, x.W, C))
        return y

    def _mid(self, x, prefix, ch):
        x = self._res_block(x, prefix + ".resnets.0", ch, ch)
        if self.config.mid_block_add_attention:
            x = self._attention(x, prefix + ".attentions.0")
        return self._res_block(x, prefix + ".resnets.1", ch, ch)

    def _downsample(self, x: _Act, name, stride):
        full = self._conv(self._padded(x), name, x.T, x.H, x.W, x.C, x.C)
        st, sy, sx = stride
        sub = full.valid()[::st, ::sy, ::sx]
        T, H, W = sub.shape[:3]
        y = _Act(self._empty(T * (H + 2) * (W + 2) * x.C), T, H, W, x.C)
        y.valid().copy_(sub)
        return y

    def _upsample(self, x: _Act, name, factor):
        ft, fy, fx = factor
        v = x.valid()
        if ft == 2 and x.T > 1:
            v = torch.cat([v[:1], v[1:].repeat_interleave(2, dim=0)], dim=0)   # the first frame is never doubled in time
        v = v.repeat_interleave(fy, dim=1).repeat_interleave(fx, dim=2)
        T, H, W = v.shape[:3]
        p = self._padded_from(v, T, H, W, x.C)
        del v
        return self._conv(p, name, T, H, W, x.C, x.C)

    # ---- encoder / decoder --------------------------------------------------------------------------------------------------
    def _encode_one(self, x):
        """x [3, T, H, W] bf16 -> moments [2 z, T', H / 8, W / 8]."""
        c = self.config
        boc = list(c.block_out_channels)
        _, T, H, W = x.shape
        p = self._padded_from(torch.nn.functional.pad(x.permute(1, 2, 3, 0), (0, 64 - c.in_channels)), T, H, W, 64)
        h = self._conv(p, "encoder.conv_in", T, H, W, c.in_channels, boc[0])
        del p
        for i, ci, co, stride in self._plan(False):
            for j in range(c.layers_per_block):
                h = self._res_block(h, "encoder.down_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if stride is not None:
                h = self._downsample(h, "encoder.down_blocks.%d.downsamplers.0" % i, stride)
        h = self._mid(h, "encoder.mid_block", boc[-1])
        p = self._replicate_borders(self._norm(h, "encoder.conv_norm_out"), h.T, h.H, h.W, h.C)
        mom = self._conv(p, "encoder.conv_out", h.T, h.H, h.W, boc[-1], 2 * c.latent_channels)
        del p
        w, b = self.w["quant_conv"]
        q = _Act(self._empty(mom.T * mom.rows * mom.C), mom.T, mom.H, mom.W, mom.C)
        _lib.gemm(mom.buf, w, q.buf, mom.T * mom.rows, mom.C, mom.C, mom.C, mom.C, mom.C, bias=b)
        return q.valid()[..., : 2 * c.latent_channels].permute(3, 0, 1, 2).contiguous()

    def _decode_tile(self, z):
        """z [z_dim, L, h, w] bf16 -> frames [3, 4 (L - 1) + 1, 8 h, 8 w] bf16: post_quant_conv + `HunyuanVideoDecoder3D`."""
        c = self.config
        boc = list(c.block_out_channels)
        zc, L, h, w = z.shape
        zin = torch.zeros(L * h * w * 64, device=self.device, dtype=BF)
        zin.view(L, h, w, 64)[..., :zc] = z.permute(1, 2, 3, 0)
        wq, bq = self.w["post_quant_conv"]
        x = self._empty(L * h * w * 64)
        _lib.gemm(zin, wq, x, L * h * w, 64, 64, 64, 64, 64, bias=bq)
        p = self._padded_from(x.view(L, h, w, 64), L, h, w, 64)
        cur = self._conv(p, "decoder.conv_in", L, h, w, zc, boc[-1])
        del p, x, zin
        cur = self._mid(cur, "decoder.mid_block", boc[-1])
        for i, ci, co, factor in self._plan(True):
            for j in range(c.layers_per_block + 1):
                cur = self._res_block(cur, "decoder.up_blocks.%d.resnets.%d" % (i, j), ci if j == 0 else co, co)
            if factor is not None:
                cur = self._upsample(cur, "decoder.up_blocks.%d.upsamplers.0" % i, factor)
        p = self._replicate_borders(self._norm(cur, "decoder.conv_norm_out"), cur.T, cur.H, cur.W, cur.C)
        rgb = self._conv(p, "decoder.conv_out", cur.T, cur.H, cur.W, boc[0], c.out_channels, cout_pad=4)
        del p
        v = rgb.view(cur.T, cur.H + 2, cur.W + 2, 4)[:, : cur.H, : cur.W, : c.out_channels]
        return v.permute(3, 0, 1, 2).contiguous()

    def _decode_one(self, z):
        """`_decode` with framewise decoding: more than 4 latent frames go through `_temporal_tiled_decode`."""
        ratio = self.temporal_compression_ratio
        lat_min = self.tile_sample_min_num_frames // ratio
        L = z.shape[1]
        if not self.use_framewise_decoding or L <= lat_min:
            return self._decode_tile(z)
        lat_stride = self.tile_sample_stride_num_frames // ratio
        keep = self.tile_sample_stride_num_frames
        blend = self.tile_sample_min_num_frames - keep
        out, prev = [], None
        for i in range(0, L, lat_stride):
            d = self._decode_tile(z[:, i: i + lat_min + 1].contiguous())
            if i > 0:
                d = d[:, 1:].contiguous()
                e = min(prev.shape[1], d.shape[1], blend)
                for x in range(e):                                    # blend_t: a[-e + x] * (1 - x / e) + b[x] * (x / e)
                    a = prev[:, prev.shape[1] - e + x].contiguous()
                    b = d[:, x].contiguous()
                    d[:, x] = _lib.lincomb([(1.0 - x / e, a), (x / e, b)], BF)
                out.append(d[:, :keep])
            else:
                out.append(d[:, : keep + 1])
            prev = d
        return torch.cat(out, dim=1)[:, : (L - 1) * ratio + 1].contiguous()

    # ---- public surface (diffusers names) -------------------------------------------------------------------------------------
    def _check(self, t, what, ch):
        if not (t.is_cuda and t.dim() == 5):
            raise _lib.AlgHipError("AutoencoderKLHunyuanVideo.%s: a 5-D device tensor is required (HIP-only path)" % what)
        if t.shape[1] != ch:
            raise ValueError("%s input has %d channels, expected %d" % (what, t.shape[1], ch))

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLHunyuanVideo.encode`: x [B, 3, T, H, W] (T = 4k + 1 <= 16: no temporal tiling below 17 frames; the
        reference encodes ONE frame, hy:578-582) -> `.latent_dist` over [B, 16, 1 + (T - 1) / 4, H / 8, W / 8]."""
        self._check(x, "encode", self.config.in_channels)
        B, _, T, H, W = x.shape
        if T > self.tile_sample_min_num_frames:
            raise NotImplementedError("temporal tiled ENCODE (more than 16 frames) is not on the reference's path")
        if (T - 1) % 4 or H % 8 or W % 8:
            raise ValueError("encode() takes 4k + 1 frames with H, W multiples of 8 (got %d x %d x %d)" % (T, H, W))
        mom = torch.stack([self._encode_one(x[b].to(BF)) for b in range(B)])
        dist = DiagonalGaussianDistribution(mom)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLHunyuanVideo.decode`: z [B, 16, L, h, w] -> sample [B, 3, 4 (L - 1) + 1, 8 h, 8 w] (not clamped: the
        video processor clamps)."""
        self._check(z, "decode", self.config.latent_channels)
        out = torch.stack([self._decode_one(z[b].to(BF)) for b in range(z.shape[0])])
        return DecoderOutput(sample=out) if return_dict else (out,)
