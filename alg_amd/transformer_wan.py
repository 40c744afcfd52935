"""Wan 2.1 image-to-video DiT forward, MI355X-native.

Drop-in for the component the reference's Wan sampler calls at pipeline_wan_image2video_lowpass.py:910-917
(``transformer(hidden_states=, timestep=, encoder_hidden_states=, encoder_hidden_states_image=, attention_kwargs=,
return_dict=False)[0]``) and reads ``.dtype`` (wan:799) and ``.config.patch_size`` (wan:550) from.  The arithmetic follows
diffusers' WanTransformer3DModel (diffusers @ be2fb77, not in the reference tree -- parity unpinned, see
oracle/wan_oracle.py); state-dict names are diffusers', so a local checkpoint loads 1:1.

The forward is nothing but launch order over the C ABI of libalg_hip.so:
    alg_patchify3d -> alg_gemm_bf16 (patch embed) ; alg_timestep_embedding_f32 -> alg_linear_f32 x2 (fp32 time embedder)
    -> alg_gemm_bf16 (time_proj) -> alg_wan_modulation (scale_shift_table + temb for all 40 blocks in one launch)
    text / image embedders: alg_gemm_bf16 (+GELU-tanh), alg_layernorm_mod_f32, alg_gelu_erf
    per block:  alg_layernorm_mod_f32 -> alg_gemm_bf16 (fused QK; V written transposed + permuted) -> alg_rmsnorm_rope x2
                -> alg_flash_attn_d128 -> alg_gemm_bf16 (out proj, fp32-gate residual epilogue)
                alg_layernorm_mod_f32 -> q GEMM + rmsnorm ; text / image K (GEMM + rmsnorm) and V^T (GEMM)
                -> alg_flash_attn_d128 x2 -> alg_lincomb (text + image) -> alg_gemm_bf16 (out proj + residual)
                alg_layernorm_mod_f32 -> alg_gemm_bf16 (GELU-tanh) -> alg_gemm_bf16 (fp32-gate residual)
    alg_layernorm_mod_f32 -> alg_gemm_bf16 (proj_out) -> alg_unpatchify3d
PyTorch owns device memory and the stream only; there is no CPU fallback.
"""
from __future__ import annotations

import glob
import json
import math
import os
from dataclasses import asdict, dataclass
from types import SimpleNamespace

import torch

from . import _lib

BF = torch.bfloat16


@dataclass
class WanTransformerConfig:
    patch_size: tuple = (1, 2, 2)
    num_attention_heads: int = 40
    attention_head_dim: int = 128
    in_channels: int = 36
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 13824
    num_layers: int = 40
    cross_attn_norm: bool = True
    qk_norm: str = "rms_norm_across_heads"
    eps: float = 1e-6
    image_dim: int = 1280
    added_kv_proj_dim: int = 5120
    rope_max_seq_len: int = 1024
    pos_embed_seq_len: int = None

    @property
    def dim(self):
        return self.num_attention_heads * self.attention_head_dim

    def to_dict(self):
        return asdict(self)


def parameter_shapes(cfg):
    """diffusers state-dict name -> (shape, dtype kept after from_pretrained(torch_dtype=bf16))."""
    D, Ff = cfg.dim, cfg.ffn_dim
    pt, ph, pw = cfg.patch_size
    f32 = torch.float32
    s = {"patch_embedding.weight": ((D, cfg.in_channels, pt, ph, pw), BF), "patch_embedding.bias": ((D,), BF)}
    ce = "condition_embedder."
    s[ce + "time_embedder.linear_1.weight"] = ((D, cfg.freq_dim), f32)
    s[ce + "time_embedder.linear_1.bias"] = ((D,), f32)
    s[ce + "time_embedder.linear_2.weight"] = ((D, D), f32)
    s[ce + "time_embedder.linear_2.bias"] = ((D,), f32)
    for n, shp in (("time_proj", (6 * D, D)), ("text_embedder.linear_1", (D, cfg.text_dim)),
                   ("text_embedder.linear_2", (D, D))):
        s[ce + n + ".weight"] = (shp, BF)
        s[ce + n + ".bias"] = ((shp[0],), BF)
    if cfg.image_dim is not None:
        I = cfg.image_dim
        for n, shp in (("image_embedder.ff.net.0.proj", (I, I)), ("image_embedder.ff.net.2", (D, I))):
            s[ce + n + ".weight"] = (shp, BF)
            s[ce + n + ".bias"] = ((shp[0],), BF)
        for n, d in (("image_embedder.norm1", I), ("image_embedder.norm2", D)):
            s[ce + n + ".weight"] = ((d,), BF)
            s[ce + n + ".bias"] = ((d,), BF)
        if cfg.pos_embed_seq_len is not None:   # first-last-frame checkpoints (FLF2V): learned position embedding of the 2 x 257 image tokens
            s[ce + "image_embedder.pos_embed"] = ((1, cfg.pos_embed_seq_len, I), BF)
    for l in range(cfg.num_layers):
        b = f"blocks.{l}."
        s[b + "scale_shift_table"] = ((1, 6, D), f32)
        for a in ("attn1", "attn2"):
            names = ["to_q", "to_k", "to_v", "to_out.0"]
            if a == "attn2" and cfg.added_kv_proj_dim is not None:
                names += ["add_k_proj", "add_v_proj"]
            for n in names:
                s[b + f"{a}.{n}.weight"] = ((D, D), BF)
                s[b + f"{a}.{n}.bias"] = ((D,), BF)
            s[b + f"{a}.norm_q.weight"] = ((D,), BF)
            s[b + f"{a}.norm_k.weight"] = ((D,), BF)
            if a == "attn2" and cfg.added_kv_proj_dim is not None:
                s[b + f"{a}.norm_added_k.weight"] = ((D,), BF)
        if cfg.cross_attn_norm:
            s[b + "norm2.weight"] = ((D,), f32)
            s[b + "norm2.bias"] = ((D,), f32)
        s[b + "ffn.net.0.proj.weight"] = ((Ff, D), BF)
        s[b + "ffn.net.0.proj.bias"] = ((Ff,), BF)
        s[b + "ffn.net.2.weight"] = ((D, Ff), BF)
        s[b + "ffn.net.2.bias"] = ((D,), BF)
    s["scale_shift_table"] = ((1, 2, D), f32)
    n_out = cfg.out_channels * pt * ph * pw
    s["proj_out.weight"] = ((n_out, D), BF)
    s["proj_out.bias"] = ((n_out,), BF)
    return s


def synthetic_state_dict(cfg, seed=1234, device="cuda"):
    """Seeded synthetic weights at the configured shapes (no network here): 1/sqrt(fan_in) matrices, small biases,
    gains around 1, generated on the device tensor by tensor."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    sd = {}
    for name, (shape, dt) in parameter_shapes(cfg).items():
        r = lambda: torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        if "scale_shift_table" in name:
            t = r() / shape[-1] ** 0.5
        elif name.endswith("weight") and len(shape) == 1:
            t = 1.0 + 0.1 * r()
        elif name.endswith("bias"):
            t = 0.02 * r()
        elif name.endswith("pos_embed"):
            t = 0.5 * r()
        else:
            t = r() / math.prod(shape[1:]) ** 0.5
        sd[name] = t.to(dt)
    return sd


class WanTransformer3DModel:
    dtype = BF

    def __init__(self, config: WanTransformerConfig, weights: dict, device="cuda", fp8=False):
        """``fp8=True`` (BASELINE config 5): the seven large linears of every block run on the fp8 MFMA -- weights are
        quantised once to OCP e4m3 with one scale per output channel, activations per token right before each GEMM
        (alg_quantize_fp8_rows); norms, attention, embedders and the residual stream stay bf16 / fp32."""
        self.fp8 = bool(fp8)
        self.pair_qkv = True     # bf16: Q|K and V^T projections of a block as one alg_gemm_bf16_pair launch (bit-identical)
        self.dual_cross = True   # I2V: text + image cross-attention of a block as one alg_flash_attn_d128_dual launch (bit-identical to two launches + add)
        self.fuse_quant = True   # fp8: the modulated LayerNorm writes the e4m3 tokens + row scales itself (bit-identical to the quantiser pass)
        # bf16: the five weight-times-token linears of a block outside the Q|K / V^T pair (out, cross q, cross out, ff1, ff2) keep a copy
        # packed in MFMA-fragment order (alg_pack_b_p11) and run GEMM schedule 11 (bit-identical to schedule 10); False, or
        # ALG_GEMM_PIPE set to another schedule than 10: the row-major weights
        self.packed_weights = True
        if config.qk_norm != "rms_norm_across_heads" or config.attention_head_dim != 128:
            raise NotImplementedError("the Wan DiT path is built for rms_norm_across_heads and head_dim 128")
        if tuple(config.patch_size)[0] != 1:
            raise NotImplementedError("temporal patch size 1 only (every Wan 2.1 checkpoint)")
        if config.pos_embed_seq_len is not None and config.image_dim is None:
            raise ValueError("pos_embed_seq_len (FLF2V) needs an image embedder (image_dim)")
        self.config = config
        self.device = torch.device(device)
        dev = self.device
        if dev.type != "cuda":
            raise _lib.AlgHipError("WanTransformer3DModel runs on the GPU only (HIP kernels); there is no CPU fallback")
        _lib.load_library()
        missing = [k for k in parameter_shapes(config) if k not in weights]
        if missing:
            raise KeyError("state dict is missing %d tensors, e.g. %s" % (len(missing), missing[:3]))
        D = config.dim
        bf = lambda n: weights[n].to(device=dev, dtype=BF).contiguous()
        f32 = lambda n: weights[n].to(device=dev, dtype=torch.float32).contiguous()
        w = SimpleNamespace()
        # patch embed: Conv3d weight [D, C, 1, ph, pw] -> [D, Kpad] rows (c, py, px), zero padded to K % 64 == 0
        kin = config.in_channels * config.patch_size[1] * config.patch_size[2]
        self.k_patch = (kin + 63) // 64 * 64
        wp = torch.zeros(D, self.k_patch, dtype=BF, device=dev)
        wp[:, :kin] = bf("patch_embedding.weight").reshape(D, kin)
        w.patch_w, w.patch_b = wp, bf("patch_embedding.bias")
        ce = "condition_embedder."
        w.t1_w, w.t1_b = f32(ce + "time_embedder.linear_1.weight"), f32(ce + "time_embedder.linear_1.bias")
        w.t2_w, w.t2_b = f32(ce + "time_embedder.linear_2.weight"), f32(ce + "time_embedder.linear_2.bias")
        w.tp_w, w.tp_b = bf(ce + "time_proj.weight"), bf(ce + "time_proj.bias")
        w.x1_w, w.x1_b = bf(ce + "text_embedder.linear_1.weight"), bf(ce + "text_embedder.linear_1.bias")
        w.x2_w, w.x2_b = bf(ce + "text_embedder.linear_2.weight"), bf(ce + "text_embedder.linear_2.bias")
        self.has_image = config.image_dim is not None
        if self.has_image:
            ie = ce + "image_embedder."
            w.in1_w, w.in1_b = f32(ie + "norm1.weight"), f32(ie + "norm1.bias")   # FP32LayerNorm: weight.float()
            w.in2_w, w.in2_b = f32(ie + "norm2.weight"), f32(ie + "norm2.bias")
            w.if1_w, w.if1_b = bf(ie + "ff.net.0.proj.weight"), bf(ie + "ff.net.0.proj.bias")
            w.if2_w, w.if2_b = bf(ie + "ff.net.2.weight"), bf(ie + "ff.net.2.bias")
            # FLF2V (wan:805-812 hands [first, last] CLIP embeddings as 2 x 257 tokens): WanImageEmbedding.pos_embed [1, 514, I]
            w.img_pos = bf(ie + "pos_embed").reshape(-1, config.image_dim) if config.pos_embed_seq_len is not None else None
        w.tables = torch.stack([f32(f"blocks.{l}.scale_shift_table").reshape(6, D)
                                for l in range(config.num_layers)]).contiguous()      # [L, 6, D]
        w.table_out = f32("scale_shift_table").reshape(1, 2, D).contiguous()
        w.out_w, w.out_b = bf("proj_out.weight"), bf("proj_out.bias")
        self.blocks = []
        for l in range(config.num_layers):
            b = f"blocks.{l}."
            L = SimpleNamespace()
            L.wqk = torch.cat([bf(b + "attn1.to_q.weight"), bf(b + "attn1.to_k.weight")], dim=0).contiguous()
            L.bqk = torch.cat([bf(b + "attn1.to_q.bias"), bf(b + "attn1.to_k.bias")]).contiguous()
            L.wv, L.bv = bf(b + "attn1.to_v.weight"), bf(b + "attn1.to_v.bias")
            L.wo, L.bo = bf(b + "attn1.to_out.0.weight"), bf(b + "attn1.to_out.0.bias")
            L.nq, L.nk = bf(b + "attn1.norm_q.weight"), bf(b + "attn1.norm_k.weight")
            L.n2w = f32(b + "norm2.weight") if config.cross_attn_norm else None
            L.n2b = f32(b + "norm2.bias") if config.cross_attn_norm else None
            L.cq_w, L.cq_b = bf(b + "attn2.to_q.weight"), bf(b + "attn2.to_q.bias")
            L.ck_w, L.ck_b = bf(b + "attn2.to_k.weight"), bf(b + "attn2.to_k.bias")
            L.cv_w, L.cv_b = bf(b + "attn2.to_v.weight"), bf(b + "attn2.to_v.bias")
            L.co_w, L.co_b = bf(b + "attn2.to_out.0.weight"), bf(b + "attn2.to_out.0.bias")
            L.cnq, L.cnk = bf(b + "attn2.norm_q.weight"), bf(b + "attn2.norm_k.weight")
            if config.added_kv_proj_dim is not None:
                L.ak_w, L.ak_b = bf(b + "attn2.add_k_proj.weight"), bf(b + "attn2.add_k_proj.bias")
                L.av_w, L.av_b = bf(b + "attn2.add_v_proj.weight"), bf(b + "attn2.add_v_proj.bias")
                L.cnak = bf(b + "attn2.norm_added_k.weight")
            L.f1_w, L.f1_b = bf(b + "ffn.net.0.proj.weight"), bf(b + "ffn.net.0.proj.bias")
            L.f2_w, L.f2_b = bf(b + "ffn.net.2.weight"), bf(b + "ffn.net.2.bias")
            if self.fp8:
                for name in ("wqk", "wv", "wo", "cq_w", "co_w", "f1_w", "f2_w"):
                    wt = getattr(L, name)
                    q = torch.empty(wt.shape, dtype=torch.uint8, device=dev)
                    sc = torch.empty(wt.shape[0], dtype=torch.float32, device=dev)
                    _lib.quantize_fp8_rows(wt, q, sc, wt.shape[0], wt.shape[1])
                    setattr(L, name, (q, sc))          # the bf16 copy is dropped: half the weight memory
            else:
                L.packed = {name: _lib.PackedB(getattr(L, name)) for name in ("wo", "cq_w", "co_w", "f1_w", "f2_w")}
            self.blocks.append(L)
        self.w = w
        self._ws = {}
        self._rope_cache = {}
        self.profile = None  # dict name -> [ms] when set (bench / tests)

    # ---- construction ------------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config=None, seed=1234, device="cuda", fp8=False):
        config = config or WanTransformerConfig()
        return cls(config, synthetic_state_dict(config, seed=seed, device=device), device=device, fp8=fp8)

    @classmethod
    def from_pretrained(cls, path, subfolder="transformer", torch_dtype=BF, device="cuda", fp8=False, **_):
        """Load a diffusers-format checkpoint directory (config.json + *.safetensors) from local disk."""
        root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        cfg_path = os.path.join(root, "config.json")
        if not os.path.exists(cfg_path):
            raise FileNotFoundError("%s not found: weights must be on local disk (no network access in this build; use "
                                    "WanTransformer3DModel.from_synthetic for shape-faithful synthetic weights)" % cfg_path)
        with open(cfg_path) as f:
            raw = json.load(f)
        fields = WanTransformerConfig.__dataclass_fields__
        cfg = WanTransformerConfig(**{k: (tuple(v) if k == "patch_size" else v) for k, v in raw.items() if k in fields})
        from .weights import read_shards
        sd = read_shards(root)
        return cls(cfg, sd, device=device, fp8=fp8)

    def to(self, *args, **kwargs):
        return self

    # ---- helpers -----------------------------------------------------------------------------------------------------
    def _timed(self, name, fn, *a, **k):
        if self.profile is None:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.profile.setdefault(name, []).append((e0, e1))
        return r

    def rope_tables(self, F_, H, W):
        """WanRotaryPosEmbed as (cos, sin) fp32 [S, 64]: axis split (44, 42, 42) of the 128-wide head, float64 angles."""
        key = (F_, H, W)
        hit = self._rope_cache.get(key)
        if hit is None:
            cfg = self.config
            d = cfg.attention_head_dim
            pt, ph, pw = cfg.patch_size
            h_dim = w_dim = 2 * (d // 6)
            t_dim = d - h_dim - w_dim
            tabs = []
            for dim in (t_dim, h_dim, w_dim):
                freqs = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
                tabs.append(torch.outer(torch.arange(cfg.rope_max_seq_len, dtype=torch.float64), freqs))
            ppf, pph, ppw = F_ // pt, H // ph, W // pw
            ang = torch.cat([tabs[0][:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1),
                             tabs[1][:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1),
                             tabs[2][:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)], dim=-1)
            ang = ang.reshape(ppf * pph * ppw, -1)
            hit = (torch.cos(ang).float().to(self.device).contiguous(), torch.sin(ang).float().to(self.device).contiguous())
            self._rope_cache[key] = hit
        return hit

    def _workspace(self, N, S, n_txt, n_img):
        key = (N, S, n_txt, n_img)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()  # one shape at a time: the buffers are large
            cfg, dev = self.config, self.device
            D, Ff = cfg.dim, cfg.ffn_dim
            e = lambda *s, dt=BF: torch.empty(*s, dtype=dt, device=dev)
            z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
            ws = SimpleNamespace()
            ws.S_pad = (S + 63) // 64 * 64
            ws.patches = e(N, S, self.k_patch)
            ws.x, ws.y, ws.att, ws.qc = e(N, S, D), e(N, S, D), e(N, S, D), e(N, S, D)
            ws.qk = e(N, S, 2 * D)
            ws.vt = z(N, D, ws.S_pad)              # padding columns stay zero (multiplied by p = 0)
            ws.h = e(N, S, Ff)
            ws.o2 = None                           # second cross-attention output: only the two-launch form (dual_cross off) needs it
            ws.tok = e(N, S, self.w.out_w.shape[0])
            if self.fp8:
                ws.q8 = e(N * S, max(D, Ff), dt=torch.uint8)      # e4m3 copy of the current GEMM input
                ws.q8s = e(N * S, dt=torch.float32)
            ws.temb_f32 = e(N, cfg.freq_dim, dt=torch.float32)
            ws.t1 = e(N, D, dt=torch.float32)
            ws.temb, ws.temb_silu = e(N, D), e(N, D)
            ws.tproj = e(N, 6 * D)
            ws.mod = e(cfg.num_layers, N, 6, D, dt=torch.float32)
            ws.mod_out = e(1, N, 2, D, dt=torch.float32)
            ws.txt_h, ws.txt = e(N, n_txt, D), e(N, n_txt, D)
            ws.kt = e(N, n_txt, D)
            ws.txt_pad = (n_txt + 63) // 64 * 64
            ws.vtt = z(N, D, ws.txt_pad)
            if n_img:
                I = cfg.image_dim
                ws.img_n, ws.img_h = e(N, n_img, I), e(N, n_img, I)
                ws.img_p, ws.img = e(N, n_img, D), e(N, n_img, D)
                ws.ki = e(N, n_img, D)
                ws.img_pad = (n_img + 63) // 64 * 64
                ws.vti = z(N, D, ws.img_pad)
            self._ws[key] = ws
        return ws

    # ---- forward -----------------------------------------------------------------------------------------------------
    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
                 attention_kwargs=None, return_dict=True):
        cfg, w, G, T = self.config, self.w, _lib.gemm, self._timed
        if hidden_states.device.type != "cuda":
            raise _lib.AlgHipError("WanTransformer3DModel needs device tensors; there is no CPU fallback")
        N, C, F_, H, W = hidden_states.shape
        pt, ph, pw = cfg.patch_size
        if C != cfg.in_channels or H % ph or W % pw:
            raise ValueError(f"hidden_states must be [N, {cfg.in_channels}, F, H, W] with H, W divisible by the patch size")
        if self.has_image and encoder_hidden_states_image is None:
            raise ValueError("this checkpoint has an image embedder: pass encoder_hidden_states_image")
        D, Ff, heads = cfg.dim, cfg.ffn_dim, cfg.num_attention_heads
        S = F_ * (H // ph) * (W // pw)
        n_txt = encoder_hidden_states.shape[1]
        if encoder_hidden_states_image is not None and cfg.pos_embed_seq_len is not None:
            # FLF2V: the [first, last] image embeddings arrive as separate batch rows ([2 N, 257, I], wan:805-812 + wan:904-908) and
            # are one sample's 2 x 257 tokens: WanImageEmbedding views them as [N, 514, I] before anything else
            b2, s2, i2 = encoder_hidden_states_image.shape
            if (b2 * s2) % cfg.pos_embed_seq_len or b2 * s2 // cfg.pos_embed_seq_len != N:
                raise ValueError("FLF2V: encoder_hidden_states_image %s does not view as [%d, %d, %d]" % (
                    tuple(encoder_hidden_states_image.shape), N, cfg.pos_embed_seq_len, i2))
            encoder_hidden_states_image = encoder_hidden_states_image.reshape(N, cfg.pos_embed_seq_len, i2)
        n_img = encoder_hidden_states_image.shape[1] if encoder_hidden_states_image is not None else 0
        ws = self._workspace(N, S, n_txt, n_img)
        S_pad, scale = ws.S_pad, 1.0 / math.sqrt(cfg.attention_head_dim)
        cos, sin = self.rope_tables(F_, H, W)
        hs = hidden_states.to(BF).contiguous()
        ehs_in = encoder_hidden_states.to(BF).contiguous()

        # ---- embedders ----
        T("patchify", _lib.patchify3d, hs, ws.patches, N, C, F_, H, W, ph, pw, self.k_patch)
        T("gemm_patch", G, ws.patches, w.patch_w, ws.x, N * S, D, self.k_patch, self.k_patch, self.k_patch, D, bias=w.patch_b)
        t = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(N)
        t = t.contiguous()
        _lib.timestep_embedding_f32(t, ws.temb_f32, N, cfg.freq_dim)
        _lib.linear_f32(ws.temb_f32, w.t1_w, w.t1_b, ws.t1, None, None, N, D, cfg.freq_dim, act=1)
        _lib.linear_f32(ws.t1, w.t2_w, w.t2_b, None, ws.temb, ws.temb_silu, N, D, D, act=0)
        G(ws.temb_silu, w.tp_w, ws.tproj, N, 6 * D, D, D, D, 6 * D, bias=w.tp_b)
        _lib.wan_modulation(w.tables, ws.tproj, ws.mod, cfg.num_layers, N, 6, D, True)
        _lib.wan_modulation(w.table_out, ws.temb, ws.mod_out, 1, N, 2, D, False)
        G(ehs_in, w.x1_w, ws.txt_h, N * n_txt, D, cfg.text_dim, cfg.text_dim, cfg.text_dim, D, bias=w.x1_b,
          act=_lib.ACT_GELU_TANH)
        G(ws.txt_h, w.x2_w, ws.txt, N * n_txt, D, D, D, D, D, bias=w.x2_b)
        if n_img:
            I = cfg.image_dim
            im = encoder_hidden_states_image.to(BF).contiguous()
            if w.img_pos is not None:   # x + pos_embed in the embedder's dtype (one bf16 rounding), per sample (the table has no batch)
                im = torch.stack([_lib.lincomb([(1.0, im[n]), (1.0, w.img_pos)], BF) for n in range(N)])
            _lib.layernorm_mod_f32(im, ws.img_n, w.in1_w, w.in1_b, None, None, 0, N, n_img, I, 1e-5)
            G(ws.img_n, w.if1_w, ws.img_h, N * n_img, I, I, I, I, I, bias=w.if1_b)
            _lib.gelu_erf_(ws.img_h)
            G(ws.img_h, w.if2_w, ws.img_p, N * n_img, D, I, I, I, D, bias=w.if2_b)
            _lib.layernorm_mod_f32(ws.img_p, ws.img, w.in2_w, w.in2_b, None, None, 0, N, n_img, D, 1e-5)

        mod_bs = 6 * D

        def lin(name, A, Wt, C, M_, N_, K_, lda, ldc, requant=True, **kw):
            """C = epilogue(A @ W^T): bf16, or e4m3 operands when the block weights are quantised (A is re-quantised per
            token unless the previous call already left its e4m3 copy in the workspace)."""
            if not self.fp8:
                return T(name, G, A, (packed.get(id(Wt)) or Wt) if packed else Wt, C, M_, N_, K_, lda, K_, ldc, **kw)
            if requant:
                T("quant", _lib.quantize_fp8_rows, A, ws.q8, ws.q8s, N * S, K_, x_rstride=lda)
            sa = kw.pop("strideA", 0)
            return T(name, G, ws.q8, Wt[0], C, M_, N_, K_, K_, K_, ldc, a_scale=ws.q8s, b_scale=Wt[1],
                     strideA=(S * K_ if sa else 0), strideAScale=(S if sa else 0), **kw)

        # fp8 blocks: the norms write the e4m3 tokens + row scales straight into the GEMM operand workspace (the bytes
        # alg_quantize_fp8_rows would make of the bf16 norm output); ws.y is not touched
        fuse_q = self.fp8 and D % 512 == 0 and self.fuse_quant

        def ln_mod(wgt, bia, sc, sh, bs, **kw):
            if fuse_q:
                return T("ln_mod", _lib.layernorm_mod_f32_fp8, ws.x, ws.q8, ws.q8s, wgt, bia, sc, sh, bs, N, S, D, cfg.eps, **kw)
            return T("ln_mod", _lib.layernorm_mod_f32, ws.x, ws.y, wgt, bia, sc, sh, bs, N, S, D, cfg.eps, **kw)

        use_packed = not self.fp8 and self.packed_weights and os.environ.get("ALG_GEMM_PIPE", "10") == "10"
        packed = None
        for li, L in enumerate(self.blocks):
            if use_packed:   # lin() looks the row-major weight up by identity
                packed = {id(getattr(L, name)): pw for name, pw in L.packed.items()}
            m0 = li * N * 6 * D  # element offset of this block's [N, 6, D] modulation: shift, scale, gate, c_shift, c_scale, c_gate
            # ---- self-attention ----
            ln_mod(None, None, ws.mod, ws.mod, mod_bs, scale_off=m0 + D, shift_off=m0)
            vt_kw = dict(bias=L.bv, batch=N, strideB=S * D, strideC=D * S_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            if self.fp8:
                lin("gemm_qk", ws.y, L.wqk, ws.qk, N * S, 2 * D, D, D, 2 * D, requant=not fuse_q, bias=L.bqk)
                # V^T: the weight is the A operand, the (already quantised) tokens are B
                T("gemm_vt", G, L.wv[0], ws.q8, ws.vt, D, S, D, D, D, S_pad, a_scale=L.wv[1], b_scale=ws.q8s, strideBScale=S, **vt_kw)
            elif self.pair_qkv:   # both projections read y: one persistent launch (alg_gemm_bf16_pair), bit-identical
                T("gemm_qkv", _lib.gemm_pair, ((ws.y, L.wqk, ws.qk, N * S, 2 * D, D, D, D, 2 * D), dict(bias=L.bqk)),
                  ((L.wv, ws.y, ws.vt, D, S, D, D, D, S_pad), vt_kw))
            else:
                lin("gemm_qk", ws.y, L.wqk, ws.qk, N * S, 2 * D, D, D, 2 * D, bias=L.bqk)
                T("gemm_vt", G, L.wv, ws.y, ws.vt, D, S, D, D, D, S_pad, **vt_kw)
            T("rms_rope", _lib.rmsnorm_rope_, ws.qk, L.nq, cos, sin, 2 * D, N, S, D, cfg.eps)
            T("rms_rope", _lib.rmsnorm_rope_, ws.qk, L.nk, cos, sin, 2 * D, N, S, D, cfg.eps, x_off=D)
            T("attn_self", _lib.flash_attn_d128, ws.qk, ws.qk, ws.vt, ws.att, N, heads, S, S, S * 2 * D, 2 * D,
              S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, scale, k_off=D)
            lin("gemm_out", ws.att, L.wo, ws.x, S, D, D, D, D, bias=L.bo, R=ws.x, ldr=D, gate=ws.mod,
                gate_off=m0 + 2 * D, strideGate=mod_bs, batch=N, strideA=S * D, strideC=S * D, strideR=S * D,
                seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)
            # ---- cross-attention: image tokens and text tokens attend separately, outputs are added ----
            if cfg.cross_attn_norm:
                ln_mod(L.n2w, L.n2b, None, None, 0)
                lin("gemm_cq", ws.y, L.cq_w, ws.qc, N * S, D, D, D, D, requant=not fuse_q, bias=L.cq_b)
            else:
                lin("gemm_cq", ws.x, L.cq_w, ws.qc, N * S, D, D, D, D, bias=L.cq_b)
            T("rms_rope", _lib.rmsnorm_rope_, ws.qc, L.cnq, None, None, D, N, S, D, cfg.eps)
            G(ws.txt, L.ck_w, ws.kt, N * n_txt, D, D, D, D, D, bias=L.ck_b)
            _lib.rmsnorm_rope_(ws.kt, L.cnk, None, None, D, N, n_txt, D, cfg.eps)
            G(L.cv_w, ws.txt, ws.vtt, D, n_txt, D, D, D, ws.txt_pad, bias=L.cv_b, batch=N, strideB=n_txt * D,
              strideC=D * ws.txt_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            if n_img:
                G(ws.img, L.ak_w, ws.ki, N * n_img, D, D, D, D, D, bias=L.ak_b)
                _lib.rmsnorm_rope_(ws.ki, L.cnak, None, None, D, N, n_img, D, cfg.eps)
                G(L.av_w, ws.img, ws.vti, D, n_img, D, D, D, ws.img_pad, bias=L.av_b, batch=N, strideB=n_img * D,
                  strideC=D * ws.img_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            if n_img and self.dual_cross:   # sdpa(q, k, v) + sdpa(q, k_img, v_img): one launch, Q read once, no add pass
                T("attn_cross", _lib.flash_attn_d128_dual, ws.qc, ws.kt, ws.vtt, n_txt, n_txt * D, D, D * ws.txt_pad, ws.txt_pad,
                  ws.ki, ws.vti, n_img, n_img * D, D, D * ws.img_pad, ws.img_pad, ws.att, N, heads, S, S * D, D, S * D, D, scale)
            else:
                T("attn_cross", _lib.flash_attn_d128, ws.qc, ws.kt, ws.vtt, ws.att, N, heads, S, n_txt, S * D, D, n_txt * D, D,
                  D * ws.txt_pad, ws.txt_pad, S * D, D, scale)
                if n_img:
                    if ws.o2 is None:
                        ws.o2 = torch.empty_like(ws.att)
                    T("attn_cross", _lib.flash_attn_d128, ws.qc, ws.ki, ws.vti, ws.o2, N, heads, S, n_img, S * D, D, n_img * D,
                      D, D * ws.img_pad, ws.img_pad, S * D, D, scale)
                    T("add", _lib.lincomb, [(1.0, ws.att), (1.0, ws.o2)], BF, out=ws.att)
            lin("gemm_cout", ws.att, L.co_w, ws.x, N * S, D, D, D, D, bias=L.co_b, R=ws.x, ldr=D)
            # ---- feed-forward ----
            ln_mod(None, None, ws.mod, ws.mod, mod_bs, scale_off=m0 + 4 * D, shift_off=m0 + 3 * D)
            lin("gemm_ff1", ws.y, L.f1_w, ws.h, N * S, Ff, D, D, Ff, requant=not fuse_q, bias=L.f1_b,
                act=_lib.ACT_GELU_TANH)
            lin("gemm_ff2", ws.h, L.f2_w, ws.x, S, D, Ff, Ff, D, bias=L.f2_b, R=ws.x, ldr=D, gate=ws.mod,
                gate_off=m0 + 5 * D, strideGate=mod_bs, batch=N, strideA=S * Ff, strideC=S * D, strideR=S * D,
                seg_split=1 << 30, flags=_lib.GEMM_GATE_F32)

        # ---- output head ----
        _lib.layernorm_mod_f32(ws.x, ws.y, None, None, ws.mod_out, ws.mod_out, 2 * D, N, S, D, cfg.eps, scale_off=D,
                               shift_off=0)
        n_out = w.out_w.shape[0]
        G(ws.y, w.out_w, ws.tok, N * S, n_out, D, D, D, n_out, bias=w.out_b)
        out = torch.empty(N, cfg.out_channels, F_, H, W, dtype=BF, device=self.device)
        _lib.unpatchify3d(ws.tok, n_out, out, N, cfg.out_channels, F_, H, W, ph, pw)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
