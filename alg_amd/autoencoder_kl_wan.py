"""Wan 2.1 VAE on MI355X -- the component the reference's Wan pipeline encodes the condition video with
(`pipeline_wan_image2video_lowpass.py:426-430`: `retrieve_latents(self.vae.encode(video_condition), sample_mode="argmax")`,
and again every step in the pixel-space ALG branch, `:526`) and decodes the final latents with (`:959`), i.e. diffusers'
`AutoencoderKLWan` (third-party, not vendored in the reference; restated from the published module structure -- see
oracle/wan_vae_oracle.py for what is restated and why parity is unpinned).

MI355X-first formulation:
  * the WHOLE video goes through each layer at once.  The published module walks the video in chunks (1 + 4 + 4 ... frames
    when encoding, one latent frame at a time when decoding) with a per-convolution cache of the last two frames -- a
    memory-saving device that is arithmetically the zero-padded causal convolution over the whole sequence, with two
    twists that are reproduced here: an `upsample3d` leaves the FIRST frame undoubled and never feeds it to its temporal
    convolution, a `downsample3d` passes the first frame through and convolves frames (2t-2, 2t-1, 2t) afterwards
    (oracle/wan_vae_oracle.py keeps the chunked form; tests check one against the other);
  * activations are channels-last bf16 over a zero-padded grid, channel counts padded to the implicit-GEMM convolution's
    power of two (96 / 192 / 384 -> 128 / 256 / 512, zero weights in the padding), so every 3x3x3 causal convolution and
    every 3x3 resampling convolution is ONE `alg_conv_cl_bf16` launch; 1x1 convolutions, the (3, 1, 1) temporal convolutions
    (K = 3 C over three shifted frame views) and the mid-block attention are `alg_gemm_bf16` launches;
  * the mid-block attention (one head of width 384 over the H/8 x W/8 tokens of each frame) is two score GEMMs that
    together hold q k^T to 2^-17 (hi + lo bf16 parts, the second GEMM takes the first as its residual), the fp32 row
    softmax `alg_softmax_hilo`, and the P V GEMM; `WanRMS_norm` + SiLU is `alg_rms_norm_rows`.
PyTorch moves bytes between layouts (zero borders, frame shifts, nearest-neighbour duplication); arithmetic is HIP.
There is no torch fallback.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from .autoencoder_kl_cogvideox import AutoencoderKLOutput, DecoderOutput, DiagonalGaussianDistribution

BF = torch.bfloat16


@dataclass
class AutoencoderKLWanConfig:
    """Defaults = Wan-AI/Wan2.1-I2V-14B-*-Diffusers vae/config.json."""
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    attn_scales: List[float] = field(default_factory=list)
    temperal_downsample: List[bool] = field(default_factory=lambda: [False, True, True])
    dropout: float = 0.0
    latents_mean: List[float] = field(default_factory=lambda: [
        -0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
        -0.9497, 0.2503, -0.2921])
    latents_std: List[float] = field(default_factory=lambda: [
        2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253,
        2.8251, 1.9160])


def _cp(c):
    """Channel count as the convolution GEMM wants it: a power of two >= 64."""
    p = 64
    while p < c:
        p *= 2
    return p


class _Act:
    """A video in the 'virtual' layout the convolution writes: flat bf16 [T][H + 2][W + 2][Cp], valid for y < H, x < W."""

    def __init__(self, buf, T, H, W, C):
        self.buf, self.T, self.H, self.W, self.C = buf, T, H, W, C
        self.Cp, self.Hp, self.Wp = _cp(C), H + 2, W + 2
        self.rows = self.Hp * self.Wp

    def view(self):
        return self.buf[: self.T * self.rows * self.Cp].view(self.T, self.Hp, self.Wp, self.Cp)

    def valid(self):
        return self.view()[:, : self.H, : self.W]


class AutoencoderKLWan:
    def __init__(self, config: Optional[AutoencoderKLWanConfig] = None, device="cuda", dtype=BF,
                 posterior_dtype=torch.float32):
        self.config = config or AutoencoderKLWanConfig()
        c = self.config
        # The reference loads this VAE in float32 (run:51-55), so `latent_dist.sample(generator)` of the pixel-space ALG
        # branch (wan:526) draws float32 noise; torch's generator stream depends on the dtype, so the posterior keeps the
        # reference's dtype although the convolutions compute in bf16.
        self.posterior_dtype = posterior_dtype
        if dtype != BF:
            raise ValueError("the HIP VAE computes in bfloat16")
        if c.attn_scales:
            raise NotImplementedError("attn_scales is empty in every published Wan VAE config")
        self.temperal_downsample = list(c.temperal_downsample)   # the pipeline reads it (wan:176-177)
        self.temperal_upsample = self.temperal_downsample[::-1]
        self.device, self.dtype = torch.device(device), dtype
        if self.device.type != "cuda":
            raise _lib.AlgHipError("AutoencoderKLWan runs on the GPU only (HIP kernels, no CPU fallback)")
        _lib.load_library()
        self.w = {}

    # ---- structure (diffusers module order and names) ---------------------------------------------------------------
    def _encoder_plan(self):
        c = self.config
        dims = [c.base_dim * u for u in [1] + list(c.dim_mult)]
        plan, k = [], 0
        for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(c.num_res_blocks):
                plan.append(("res", "encoder.down_blocks.%d" % k, ci, co))
                ci = co
                k += 1
            if i != len(c.dim_mult) - 1:
                plan.append(("downsample3d" if c.temperal_downsample[i] else "downsample2d", "encoder.down_blocks.%d" % k, co, co))
                k += 1
        return plan, dims[-1]

    def _decoder_plan(self):
        c = self.config
        dims = [c.base_dim * u for u in [c.dim_mult[-1]] + list(c.dim_mult[::-1])]
        plan = []
        for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
            if i > 0:
                ci = ci // 2
            for j in range(c.num_res_blocks + 1):
                plan.append(("res", "decoder.up_blocks.%d.resnets.%d" % (i, j), ci, co))
                ci = co
            if i != len(c.dim_mult) - 1:
                plan.append(("upsample3d" if self.temperal_upsample[i] else "upsample2d",
                             "decoder.up_blocks.%d.upsamplers.0" % i, co, co // 2))
        return plan, dims[0], dims[-1]

    def param_shapes(self):
        c = self.config
        s = {}

        def conv(name, ci, co, k):
            s[name + ".weight"], s[name + ".bias"] = (co, ci) + tuple(k), (co,)

        def res(name, ci, co):
            s[name + ".norm1.gamma"] = (ci, 1, 1, 1)
            conv(name + ".conv1", ci, co, (3, 3, 3))
            s[name + ".norm2.gamma"] = (co, 1, 1, 1)
            conv(name + ".conv2", co, co, (3, 3, 3))
            if ci != co:
                conv(name + ".conv_shortcut", ci, co, (1, 1, 1))

        def mid(prefix, dim):
            res(prefix + ".resnets.0", dim, dim)
            s[prefix + ".attentions.0.norm.gamma"] = (dim, 1, 1)
            conv(prefix + ".attentions.0.to_qkv", dim, 3 * dim, (1, 1))
            conv(prefix + ".attentions.0.proj", dim, dim, (1, 1))
            res(prefix + ".resnets.1", dim, dim)

        plan, top = self._encoder_plan()
        conv("encoder.conv_in", 3, c.base_dim, (3, 3, 3))
        for kind, name, ci, co in plan:
            if kind == "res":
                res(name, ci, co)
            else:
                conv(name + ".resample.1", ci, co, (3, 3))
                if kind == "downsample3d":
                    conv(name + ".time_conv", ci, ci, (3, 1, 1))
        mid("encoder.mid_block", top)
        s["encoder.norm_out.gamma"] = (top, 1, 1, 1)
        conv("encoder.conv_out", top, 2 * c.z_dim, (3, 3, 3))
        conv("quant_conv", 2 * c.z_dim, 2 * c.z_dim, (1, 1, 1))
        conv("post_quant_conv", c.z_dim, c.z_dim, (1, 1, 1))
        plan, top, last = self._decoder_plan()
        conv("decoder.conv_in", c.z_dim, top, (3, 3, 3))
        mid("decoder.mid_block", top)
        for kind, name, ci, co in plan:
            if kind == "res":
                res(name, ci, co)
            else:
                conv(name + ".resample.1", ci, co, (3, 3))
                if kind == "upsample3d":
                    conv(name + ".time_conv", ci, 2 * ci, (3, 1, 1))
        s["decoder.norm_out.gamma"] = (last, 1, 1, 1)
        conv("decoder.conv_out", last, 3, (3, 3, 3))
        return s

    # ---- weights ---------------------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda"):
        self = cls(config, device=device)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.param_shapes().items():
            if name.endswith(".gamma"):
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
        return component_from_pretrained(cls, AutoencoderKLWanConfig, path, subfolder, device=device)

    def load_state_dict(self, sd, strict=True):
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if missing and strict:
            raise KeyError("missing Wan VAE weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        dev = self.device
        W = {}

        def conv_w(name, pair_ok=False, cout_pad=None):
            """[Cout][Cin][taps...] -> ([Cout_p][taps * Cin_p] bf16, [Cout_p] bias, pair flag): tap-major, channels innermost."""
            w = sd[name + ".weight"].float()
            co, ci = w.shape[:2]
            cip = _cp(ci)
            cop = cout_pad or _cp(co)
            w = w.reshape(co, ci, -1).permute(0, 2, 1)
            w = torch.nn.functional.pad(w, (0, cip - ci, 0, 0, 0, cop - co))
            b = torch.nn.functional.pad(sd[name + ".bias"].float(), (0, cop - co))
            w, b = w.reshape(cop, -1).contiguous().to(dev, BF), b.contiguous().to(dev, BF)
            if pair_ok and cop == 128 and w.shape[1] // cip == 27:
                return _lib.pack_conv_pair(w, b, 3) + (True,)
            return w, b, False

        def gamma(name):
            g = sd[name].float().reshape(-1)
            return torch.nn.functional.pad(g, (0, _cp(g.numel()) - g.numel())).to(dev, BF).contiguous()

        def res(name, ci, co):
            W[name + ".norm1"], W[name + ".norm2"] = gamma(name + ".norm1.gamma"), gamma(name + ".norm2.gamma")
            W[name + ".conv1"] = conv_w(name + ".conv1", pair_ok=True)
            W[name + ".conv2"] = conv_w(name + ".conv2", pair_ok=True)
            if ci != co:
                W[name + ".conv_shortcut"] = conv_w(name + ".conv_shortcut")

        def time_w(name, ci, co_groups):
            """(3, 1, 1) convolution as a GEMM over K = 3 * Cin_p (tap-major); the upsampler's 2 C outputs become two
            Cp-wide groups (first / second frame of the pair)."""
            w = sd[name + ".weight"].float()[:, :, :, 0, 0]                     # [Cout][Cin][3]
            b = sd[name + ".bias"].float()
            cip = _cp(ci)
            co = w.shape[0] // co_groups
            cop = _cp(co)
            wp = w.new_zeros(co_groups, cop, 3, cip)
            wp[:, :co, :, :ci] = w.reshape(co_groups, co, ci, 3).permute(0, 1, 3, 2)
            bp = b.new_zeros(co_groups, cop)
            bp[:, :co] = b.reshape(co_groups, co)
            return wp.reshape(co_groups * cop, 3 * cip).contiguous().to(dev, BF), bp.reshape(-1).contiguous().to(dev, BF)

        def mid(prefix, dim):
            res(prefix + ".resnets.0", dim, dim)
            a = prefix + ".attentions.0"
            cp = _cp(dim)
            W[a + ".norm"] = gamma(a + ".norm.gamma")
            wq = sd[a + ".to_qkv.weight"].float().reshape(3, dim, dim)
            bq = sd[a + ".to_qkv.bias"].float().reshape(3, dim)
            # [q | -q | k] rows of one GEMM (the negated copy feeds the first score GEMM: see _attention); v separately
            wqk = wq.new_zeros(3, cp, cp)
            bqk = bq.new_zeros(3, cp)
            for slot, (src, sign) in enumerate(((0, 1.0), (0, -1.0), (1, 1.0))):
                wqk[slot, :dim, :dim] = sign * wq[src]
                bqk[slot, :dim] = sign * bq[src]
            W[a + ".qk"] = (wqk.reshape(3 * cp, cp).contiguous().to(dev, BF), bqk.reshape(-1).contiguous().to(dev, BF))
            wv, bv = wq.new_zeros(cp, cp), bq.new_zeros(cp)
            wv[:dim, :dim], bv[:dim] = wq[2], bq[2]
            W[a + ".v"] = (wv.contiguous().to(dev, BF), bv.contiguous().to(dev, BF))
            W[a + ".proj"] = conv_w(a + ".proj")[:2]
            res(prefix + ".resnets.1", dim, dim)

        c = self.config
        plan, top = self._encoder_plan()
        W["encoder.conv_in"] = conv_w("encoder.conv_in", pair_ok=True)
        for kind, name, ci, co in plan:
            if kind == "res":
                res(name, ci, co)
            else:
                W[name + ".resample"] = conv_w(name + ".resample.1")
                if kind == "downsample3d":
                    W[name + ".time_conv"] = time_w(name + ".time_conv", ci, 1)
        mid("encoder.mid_block", top)
        W["encoder.norm_out"] = gamma("encoder.norm_out.gamma")
        W["encoder.conv_out"] = conv_w("encoder.conv_out")
        W["quant_conv"] = conv_w("quant_conv")[:2]
        W["post_quant_conv"] = conv_w("post_quant_conv")[:2]
        plan, top, last = self._decoder_plan()
        W["decoder.conv_in"] = conv_w("decoder.conv_in")
        mid("decoder.mid_block", top)
        for kind, name, ci, co in plan:
            if kind == "res":
                res(name, ci, co)
            else:
                W[name + ".resample"] = conv_w(name + ".resample.1", pair_ok=False)
                if kind == "upsample3d":
                    W[name + ".time_conv"] = time_w(name + ".time_conv", ci, 2)
        W["decoder.norm_out"] = gamma("decoder.norm_out.gamma")
        W["decoder.conv_out"] = conv_w("decoder.conv_out", cout_pad=4)
        self.w = W
        return self

    # ---- layout plumbing (bytes only) ----------------------------------------------------------------------------------
    def _zeros(self, n):
        return torch.zeros(n, device=self.device, dtype=BF)

    def _empty(self, n):
        return torch.empty(n, device=self.device, dtype=BF)

    def _padded(self, src, T, H, W, Cp, lead):
        """Zero-bordered convolution input [T + lead][H + 2][W + 2][Cp] (+ slack rows) with `src` [T, H, W, Cp] inside;
        lead = 2 zero frames in front for a causal 3x3x3 convolution, 0 for a per-frame 3x3 one."""
        Hp, Wp = H + 2, W + 2
        buf = self._zeros(((T + lead) * Hp * Wp + 2 * Wp + 4) * Cp)
        buf[: (T + lead) * Hp * Wp * Cp].view(T + lead, Hp, Wp, Cp)[lead:, 1:H + 1, 1:W + 1] = src
        return buf

    # ---- launches --------------------------------------------------------------------------------------------------------
    def _conv(self, pbuf, name, T, H, W, Cin, Cout, kt, res=None, cout_pad=None):
        w, b, pair = self.w[name]
        cop = cout_pad or _cp(Cout)
        if pair and ((H + 2) * (W + 2)) % 2:
            raise _lib.AlgHipError("internal: two-voxel packing needs an even padded plane")
        out = res.buf if res is not None else self._empty(T * (H + 2) * (W + 2) * cop)
        _lib.conv_cl(pbuf, w, b, None if res is None else res.buf, out, T, H + 2, W + 2, _cp(Cin), cop, kt, pair=pair)
        return _Act(out, T, H, W, Cout) if cout_pad is None else out

    def _norm(self, a: _Act, gname, silu=True):
        y = self._empty(a.T * a.rows * a.Cp)
        _lib.rms_norm_rows(a.buf, self.w[gname], y, a.T * a.rows, a.C, a.Cp, silu)
        return _Act(y, a.T, a.H, a.W, a.C)

    def _res_block(self, x: _Act, name, ci, co):
        n = self._norm(x, name + ".norm1")
        p = self._padded(n.valid(), x.T, x.H, x.W, x.Cp, 2)
        del n
        c1 = self._conv(p, name + ".conv1", x.T, x.H, x.W, ci, co, 3)
        del p
        n = self._norm(c1, name + ".norm2")
        del c1
        p = self._padded(n.valid(), x.T, x.H, x.W, n.Cp, 2)
        del n
        if ci != co:
            w, b, _ = self.w[name + ".conv_shortcut"]
            h = _Act(self._empty(x.T * x.rows * _cp(co)), x.T, x.H, x.W, co)
            _lib.gemm(x.buf, w, h.buf, x.T * x.rows, h.Cp, x.Cp, x.Cp, x.Cp, h.Cp, bias=b)
        else:
            h = x
        return self._conv(p, name + ".conv2", x.T, x.H, x.W, co, co, 3, res=h)

    def _attention(self, x: _Act, name):
        """WanAttentionBlock: per frame, one head of width C over the H * W tokens; x + proj(softmax(q k^T / sqrt(C)) v)."""
        T, n, C, Cp = x.T, x.H * x.W, x.C, x.Cp
        dense = x.valid().reshape(T * n, Cp).contiguous()
        xn = self._empty(T * n * Cp)
        _lib.rms_norm_rows(dense, self.w[name + ".norm"], xn, T * n, C, Cp, False)
        wqk, bqk = self.w[name + ".qk"]
        qk = self._empty(T * n * 3 * Cp)                              # rows: [q | -q | k]
        _lib.gemm(xn, wqk, qk, T * n, 3 * Cp, Cp, Cp, Cp, 3 * Cp, bias=bqk)
        n_pad = (n + 63) // 64 * 64
        wv, bv = self.w[name + ".v"]
        vt = self._zeros(T * Cp * n_pad)                               # V^T per frame, written by a GEMM with swapped operands
        _lib.gemm(wv, xn, vt, Cp, n, Cp, Cp, Cp, n_pad, bias=bv, batch=T, strideB=n * Cp, strideC=Cp * n_pad,
                  flags=_lib.GEMM_BIAS_PER_ROW)
        del xn
        neg_hi = self._empty(T * n * n_pad)
        lo = self._empty(T * n * n_pad)
        # neg_hi = bf16((-q) k^T);  lo = bf16(q k^T + neg_hi): together the fp32 scores to ~2^-17
        _lib.gemm(qk, qk, neg_hi, n, n, Cp, 3 * Cp, 3 * Cp, n_pad, batch=T, strideA=n * 3 * Cp, strideB=n * 3 * Cp,
                  strideC=n * n_pad, a_off=Cp, b_off=2 * Cp)
        _lib.gemm(qk, qk, lo, n, n, Cp, 3 * Cp, 3 * Cp, n_pad, R=neg_hi, ldr=n_pad, batch=T, strideA=n * 3 * Cp,
                  strideB=n * 3 * Cp, strideC=n * n_pad, strideR=n * n_pad, b_off=2 * Cp)
        del qk
        p = self._empty(T * n * n_pad)
        _lib.softmax_hilo(neg_hi, lo, p, T * n, n, n_pad, float(C) ** -0.5)
        del neg_hi, lo
        o = self._empty(T * n * Cp)
        _lib.gemm(p, vt, o, n, Cp, n_pad, n_pad, n_pad, Cp, batch=T, strideA=n * n_pad, strideB=Cp * n_pad, strideC=n * Cp)
        del p, vt
        wp, bp = self.w[name + ".proj"]
        out = self._empty(T * n * Cp)
        _lib.gemm(o, wp, out, T * n, Cp, Cp, Cp, Cp, Cp, bias=bp, R=dense, ldr=Cp)
        y = _Act(self._empty(T * x.rows * Cp), T, x.H, x.W, C)
        y.valid().copy_(out.view(T, x.H, x.W, Cp))
        return y

    def _mid(self, x, prefix, dim):
        x = self._res_block(x, prefix + ".resnets.0", dim, dim)
        x = self._attention(x, prefix + ".attentions.0")
        return self._res_block(x, prefix + ".resnets.1", dim, dim)

    def _time_gemm(self, frames3, name, rows_per_frame, Cp, groups):
        """(3, 1, 1) convolution: `frames3` = the three tap views [Tout, rows, Cp] (oldest first) -> [Tout, rows, groups * Cout_p]."""
        w, b = self.w[name]
        a = torch.cat(frames3, dim=-1).contiguous()                  # [Tout][rows][3 Cp]
        tout = a.shape[0]
        nout = w.shape[0]
        y = self._empty(tout * rows_per_frame * nout)
        _lib.gemm(a, w, y, tout * rows_per_frame, nout, 3 * Cp, 3 * Cp, 3 * Cp, nout, bias=b)
        return y.view(tout, rows_per_frame, nout)

    def _resample(self, x: _Act, name, mode, ci, co):
        T, H, W, Cp = x.T, x.H, x.W, x.Cp
        if mode == "upsample3d" and T > 1:
            v = x.view().reshape(T, x.rows, Cp)
            z = torch.zeros_like(v[:1])
            seq = torch.cat([z, z, z, v[1:]], dim=0)                  # frame 0 never enters the temporal convolution
            y = self._time_gemm([seq[1:T], seq[2:T + 1], seq[3:T + 2]], name + ".time_conv", x.rows, Cp, 2)
            y = y.view(T - 1, x.rows, 2, Cp).permute(0, 2, 1, 3).reshape(2 * (T - 1), x.rows, Cp)
            x = _Act(torch.cat([v[:1], y], dim=0).reshape(-1), 2 * T - 1, H, W, x.C)
            T = x.T
        if mode.startswith("upsample"):
            up = x.valid().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)       # nearest-exact x2
            p = self._padded(up, T, 2 * H, 2 * W, Cp, 0)
            del up
            return self._conv(p, name + ".resample", T, 2 * H, 2 * W, ci, co, 1)
        # downsample: ZeroPad2d((0, 1, 0, 1)) + Conv2d(k 3, stride 2) = the stride-2 form of the implicit GEMM over the
        # zero-bordered grid, starting at padded (1, 1); its output rows sit at the input's pitch
        p = self._padded(x.valid(), T, H, W, Cp, 0)
        w, b, _ = self.w[name + ".resample"]
        Hp, Wp = H + 2, W + 2
        m = H // 2 * Wp
        wide = self._empty(T * m * Cp)
        _lib.conv_cl(p, w, b, None, wide, T, Hp, Wp, Cp, _cp(co), 1, stride2=True)
        del p
        y = _Act(self._empty(T * (H // 2 + 2) * (W // 2 + 2) * Cp), T, H // 2, W // 2, co)
        y.valid().copy_(wide.view(T, H // 2, Wp, Cp)[:, :, : W // 2])
        del wide
        if mode == "downsample3d" and T > 1:
            v = y.view().reshape(T, y.rows, Cp)
            out = self._time_gemm([v[0:T - 2:2], v[1:T - 1:2], v[2:T:2]], name + ".time_conv", y.rows, Cp, 1)
            y = _Act(torch.cat([v[:1], out], dim=0).reshape(-1), 1 + (T - 1) // 2, y.H, y.W, co)
        return y

    # ---- encoder / decoder --------------------------------------------------------------------------------------------------
    def _encode_one(self, x):
        """x [3, T, H, W] bf16 -> moments [2 z, L, H / 8, W / 8]."""
        c = self.config
        _, T, H, W = x.shape
        p = self._padded(torch.nn.functional.pad(x.permute(1, 2, 3, 0), (0, 61)), T, H, W, 64, 2)
        h = self._conv(p, "encoder.conv_in", T, H, W, 3, c.base_dim, 3)
        del p
        plan, top = self._encoder_plan()
        for kind, name, ci, co in plan:
            h = self._res_block(h, name, ci, co) if kind == "res" else self._resample(h, name, kind, ci, co)
        h = self._mid(h, "encoder.mid_block", top)
        n = self._norm(h, "encoder.norm_out")
        p = self._padded(n.valid(), h.T, h.H, h.W, h.Cp, 2)
        del n
        mom = self._conv(p, "encoder.conv_out", h.T, h.H, h.W, top, 2 * c.z_dim, 3)
        del p
        w, b = self.w["quant_conv"]
        q = self._empty(mom.T * mom.rows * mom.Cp)
        _lib.gemm(mom.buf, w, q, mom.T * mom.rows, mom.Cp, mom.Cp, mom.Cp, mom.Cp, mom.Cp, bias=b)
        out = _Act(q, mom.T, mom.H, mom.W, 2 * c.z_dim)
        return out.valid()[..., : 2 * c.z_dim].permute(3, 0, 1, 2).contiguous()

    def _decode_one(self, z):
        """z [z_dim, L, h, w] bf16 -> frames [3, 4 (L - 1) + 1, 8 h, 8 w] bf16, clamped to [-1, 1]."""
        c = self.config
        zc, L, h, w = z.shape
        zin = _Act(self._zeros(L * (h + 2) * (w + 2) * 64), L, h, w, zc)
        zin.valid()[..., :zc] = z.permute(1, 2, 3, 0)
        wq, bq = self.w["post_quant_conv"]
        x = _Act(self._empty(L * zin.rows * 64), L, h, w, zc)
        _lib.gemm(zin.buf, wq, x.buf, L * zin.rows, 64, 64, 64, 64, 64, bias=bq)
        plan, top, last = self._decoder_plan()
        p = self._padded(x.valid(), L, h, w, 64, 2)
        hcur = self._conv(p, "decoder.conv_in", L, h, w, zc, top, 3)
        del p, x, zin
        hcur = self._mid(hcur, "decoder.mid_block", top)
        for kind, name, ci, co in plan:
            hcur = self._res_block(hcur, name, ci, co) if kind == "res" else self._resample(hcur, name, kind, ci, co)
        n = self._norm(hcur, "decoder.norm_out")
        p = self._padded(n.valid(), hcur.T, hcur.H, hcur.W, hcur.Cp, 2)
        del n
        rgb = self._conv(p, "decoder.conv_out", hcur.T, hcur.H, hcur.W, last, 3, 3, cout_pad=4)
        del p
        v = rgb.view(hcur.T, hcur.H + 2, hcur.W + 2, 4)[:, : hcur.H, : hcur.W, :3]
        return torch.clamp(v.permute(3, 0, 1, 2), min=-1.0, max=1.0).contiguous()

    # ---- public surface (diffusers names) -------------------------------------------------------------------------------------
    def _check(self, t, what, ch):
        if not (t.is_cuda and t.dim() == 5):
            raise _lib.AlgHipError("AutoencoderKLWan.%s: a 5-D device tensor is required (HIP-only path)" % what)
        if t.shape[1] != ch:
            raise ValueError("%s input has %d channels, expected %d" % (what, t.shape[1], ch))

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLWan.encode`: x [B, 3, T, H, W] (T = 4k + 1, H and W multiples of 8) -> `.latent_dist` over
        [B, 16, 1 + (T - 1) / 4, H / 8, W / 8]; the Wan pipeline takes its mode (wan:430, `sample_mode="argmax"`)."""
        self._check(x, "encode", 3)
        B, _, T, H, W = x.shape
        if (T - 1) % 4 or H % 8 or W % 8:
            raise ValueError("encode() takes 4k + 1 frames with H, W multiples of 8 (got %d x %d x %d)" % (T, H, W))
        mom = torch.stack([self._encode_one(x[b].to(BF)) for b in range(B)]).to(self.posterior_dtype)
        dist = DiagonalGaussianDistribution(mom)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """`AutoencoderKLWan.decode`: z [B, 16, L, h, w] -> sample [B, 3, 4 (L - 1) + 1, 8 h, 8 w] in [-1, 1]."""
        self._check(z, "decode", self.config.z_dim)
        out = torch.stack([self._decode_one(z[b].to(BF)) for b in range(z.shape[0])])
        return DecoderOutput(sample=out) if return_dict else (out,)
