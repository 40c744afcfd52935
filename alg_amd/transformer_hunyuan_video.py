"""HunyuanVideo image-to-video DiT forward, MI355X-native.

Drop-in for the component the reference's HunyuanVideo sampler calls at pipeline_hunyuan_video_image2video_lowpass.py:
1243-1252 (``transformer(hidden_states=, timestep=, encoder_hidden_states=, encoder_attention_mask=, pooled_projections=,
guidance=, attention_kwargs=, return_dict=False)[0]``) and reads ``.dtype`` and ``.config.{image_condition_type,
in_channels, guidance_embeds, patch_size}`` from (hy:1016, 1049-1051, 1116, 781).  The arithmetic follows diffusers'
HunyuanVideoTransformer3DModel (diffusers @ be2fb77, not in the reference tree -- parity unpinned, oracle/hy_oracle.py);
state-dict names are diffusers'.

Layout: one joint residual buffer ``[N, S + L, D]`` with the S latent tokens first and the L (padded) prompt tokens
after them, so the 20 dual-stream blocks address the two streams as row ranges and the 40 single-stream blocks run in
place on the whole buffer.  Padded prompt tokens are masked as attention KEYS in the published model, i.e. every query
attends to the prefix ``S + valid[b]`` of the joint sequence: ``alg_flash_attn_d128`` is launched per sample with that
key length (their own rows are carried along and never read by a latent).  "token_replace" conditioning maps onto the
two-segment modulation / gate support the CogVideoX kernels already have: rows < first-frame tokens take the
timestep-0 vectors (``alg_layernorm_modulate_seg``, ``alg_gemm_bf16`` gate segments).
Kernels: alg_patchify3d, alg_timestep_embedding, alg_gemm_bf16 (all linears, fused SiLU / GELU-tanh / gated-residual
epilogues, V written transposed), alg_masked_mean, alg_silu, alg_lincomb, alg_layernorm_modulate(_seg), alg_headnorm_rope,
alg_flash_attn_d128, alg_unpatchify3d.  No CPU fallback.
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
class HunyuanVideoTransformerConfig:
    in_channels: int = 16
    out_channels: int = 16
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    num_layers: int = 20
    num_single_layers: int = 40
    num_refiner_layers: int = 2
    mlp_ratio: float = 4.0
    patch_size: int = 2
    patch_size_t: int = 1
    qk_norm: str = "rms_norm"
    guidance_embeds: bool = False
    text_embed_dim: int = 4096
    pooled_projection_dim: int = 768
    rope_theta: float = 256.0
    rope_axes_dim: tuple = (16, 56, 56)
    image_condition_type: str = "token_replace"

    @property
    def dim(self):
        return self.num_attention_heads * self.attention_head_dim

    def to_dict(self):
        return asdict(self)


def parameter_shapes(cfg):
    """diffusers state-dict name -> shape (all bf16 after from_pretrained(torch_dtype=bf16))."""
    D, M = cfg.dim, int(cfg.dim * cfg.mlp_ratio)
    p, pt = cfg.patch_size, cfg.patch_size_t
    s = {}

    def lin(name, n_out, n_in):
        s[name + ".weight"] = (n_out, n_in)
        s[name + ".bias"] = (n_out,)

    s["x_embedder.proj.weight"] = (D, cfg.in_channels, pt, p, p)
    s["x_embedder.proj.bias"] = (D,)
    ce = "context_embedder."
    lin(ce + "time_text_embed.timestep_embedder.linear_1", D, 256)
    lin(ce + "time_text_embed.timestep_embedder.linear_2", D, D)
    lin(ce + "time_text_embed.text_embedder.linear_1", D, cfg.text_embed_dim)
    lin(ce + "time_text_embed.text_embedder.linear_2", D, D)
    lin(ce + "proj_in", D, cfg.text_embed_dim)
    for l in range(cfg.num_refiner_layers):
        b = ce + f"token_refiner.refiner_blocks.{l}."
        for n in ("norm1", "norm2"):
            s[b + n + ".weight"] = (D,)
            s[b + n + ".bias"] = (D,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(b + "attn." + n, D, D)
        lin(b + "ff.net.0.proj", M, D)
        lin(b + "ff.net.2", D, M)
        lin(b + "norm_out.linear", 2 * D, D)
    te = "time_text_embed."
    lin(te + "timestep_embedder.linear_1", D, 256)
    lin(te + "timestep_embedder.linear_2", D, D)
    lin(te + "text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin(te + "text_embedder.linear_2", D, D)
    if cfg.guidance_embeds:
        lin(te + "guidance_embedder.linear_1", D, 256)
        lin(te + "guidance_embedder.linear_2", D, D)
    for l in range(cfg.num_layers):
        b = f"transformer_blocks.{l}."
        lin(b + "norm1.linear", 6 * D, D)
        lin(b + "norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out"):
            lin(b + "attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[b + "attn." + n + ".weight"] = (cfg.attention_head_dim,)
        lin(b + "ff.net.0.proj", M, D)
        lin(b + "ff.net.2", D, M)
        lin(b + "ff_context.net.0.proj", M, D)
        lin(b + "ff_context.net.2", D, M)
    for l in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{l}."
        lin(b + "norm.linear", 3 * D, D)
        for n in ("to_q", "to_k", "to_v"):
            lin(b + "attn." + n, D, D)
        for n in ("norm_q", "norm_k"):
            s[b + "attn." + n + ".weight"] = (cfg.attention_head_dim,)
        lin(b + "proj_mlp", M, D)
        lin(b + "proj_out", D, D + M)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", pt * p * p * cfg.out_channels, D)
    return s


def synthetic_state_dict(cfg, seed=1234, device="cuda"):
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    sd = {}
    for name, shape in parameter_shapes(cfg).items():
        r = lambda: torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        if name.endswith("weight") and len(shape) == 1:
            t = 1.0 + 0.1 * r()
        elif name.endswith("bias"):
            t = 0.02 * r()
        else:
            t = r() / math.prod(shape[1:]) ** 0.5
        sd[name] = t.to(BF)
    return sd


class HunyuanVideoTransformer3DModel:
    dtype = BF

    def __init__(self, config: HunyuanVideoTransformerConfig, weights: dict, device="cuda"):
        if config.qk_norm != "rms_norm" or config.attention_head_dim != 128 or config.patch_size_t != 1:
            raise NotImplementedError("the HunyuanVideo DiT path is built for rms_norm, head_dim 128, patch_size_t 1")
        if sum(config.rope_axes_dim) != config.attention_head_dim:
            raise ValueError("rope_axes_dim must add up to the head dimension")
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AlgHipError("HunyuanVideoTransformer3DModel runs on the GPU only (HIP kernels); no CPU fallback")
        _lib.load_library()
        missing = [k for k in parameter_shapes(config) if k not in weights]
        if missing:
            raise KeyError("state dict is missing %d tensors, e.g. %s" % (len(missing), missing[:3]))
        dev, D = self.device, config.dim
        bf = lambda n: weights[n].to(device=dev, dtype=BF).contiguous()
        cat = lambda *ns: torch.cat([bf(n) for n in ns], dim=0).contiguous()
        lin = lambda pre: (bf(pre + ".weight"), bf(pre + ".bias"))
        w = SimpleNamespace()
        kin = config.in_channels * config.patch_size ** 2
        self.k_patch = (kin + 63) // 64 * 64
        wp = torch.zeros(D, self.k_patch, dtype=BF, device=dev)
        wp[:, :kin] = bf("x_embedder.proj.weight").reshape(D, kin)
        w.patch_w, w.patch_b = wp, bf("x_embedder.proj.bias")
        ce, te = "context_embedder.", "time_text_embed."
        w.r_t1, w.r_t2 = lin(ce + "time_text_embed.timestep_embedder.linear_1"), lin(ce + "time_text_embed.timestep_embedder.linear_2")
        w.r_x1, w.r_x2 = lin(ce + "time_text_embed.text_embedder.linear_1"), lin(ce + "time_text_embed.text_embedder.linear_2")
        w.r_in = lin(ce + "proj_in")
        self.refiner = []
        for l in range(config.num_refiner_layers):
            b = ce + f"token_refiner.refiner_blocks.{l}."
            R = SimpleNamespace()
            R.n1w, R.n1b, R.n2w, R.n2b = bf(b + "norm1.weight"), bf(b + "norm1.bias"), bf(b + "norm2.weight"), bf(b + "norm2.bias")
            R.wqk = cat(b + "attn.to_q.weight", b + "attn.to_k.weight")
            R.bqk = cat(b + "attn.to_q.bias", b + "attn.to_k.bias")
            R.v, R.o = lin(b + "attn.to_v"), lin(b + "attn.to_out.0")
            R.f1, R.f2, R.ada = lin(b + "ff.net.0.proj"), lin(b + "ff.net.2"), lin(b + "norm_out.linear")
            self.refiner.append(R)
        w.t1, w.t2 = lin(te + "timestep_embedder.linear_1"), lin(te + "timestep_embedder.linear_2")
        w.p1, w.p2 = lin(te + "text_embedder.linear_1"), lin(te + "text_embedder.linear_2")
        if config.guidance_embeds:
            w.g1, w.g2 = lin(te + "guidance_embedder.linear_1"), lin(te + "guidance_embedder.linear_2")
        self.dual = []
        for l in range(config.num_layers):
            b = f"transformer_blocks.{l}."
            L = SimpleNamespace()
            L.ada, L.ada_c = lin(b + "norm1.linear"), lin(b + "norm1_context.linear")
            L.wqk = cat(b + "attn.to_q.weight", b + "attn.to_k.weight")
            L.bqk = cat(b + "attn.to_q.bias", b + "attn.to_k.bias")
            L.wqk_c = cat(b + "attn.add_q_proj.weight", b + "attn.add_k_proj.weight")
            L.bqk_c = cat(b + "attn.add_q_proj.bias", b + "attn.add_k_proj.bias")
            L.v, L.v_c = lin(b + "attn.to_v"), lin(b + "attn.add_v_proj")
            L.o, L.o_c = lin(b + "attn.to_out.0"), lin(b + "attn.to_add_out")
            L.nq, L.nk = bf(b + "attn.norm_q.weight"), bf(b + "attn.norm_k.weight")
            L.nq_c, L.nk_c = bf(b + "attn.norm_added_q.weight"), bf(b + "attn.norm_added_k.weight")
            L.f1, L.f2 = lin(b + "ff.net.0.proj"), lin(b + "ff.net.2")
            L.f1_c, L.f2_c = lin(b + "ff_context.net.0.proj"), lin(b + "ff_context.net.2")
            L.packed = {id(t): _lib.PackedB(t) for t in (L.wqk, L.o[0], L.f1[0], L.f2[0])}     # the latent stream's linears (see packed_weights)
            self.dual.append(L)
        self.single = []
        for l in range(config.num_single_layers):
            b = f"single_transformer_blocks.{l}."
            L = SimpleNamespace()
            L.ada = lin(b + "norm.linear")
            L.wqk = cat(b + "attn.to_q.weight", b + "attn.to_k.weight")
            L.bqk = cat(b + "attn.to_q.bias", b + "attn.to_k.bias")
            L.v = lin(b + "attn.to_v")
            L.nq, L.nk = bf(b + "attn.norm_q.weight"), bf(b + "attn.norm_k.weight")
            L.mlp, L.out = lin(b + "proj_mlp"), lin(b + "proj_out")
            L.packed = {id(t): _lib.PackedB(t) for t in (L.wqk, L.mlp[0], L.out[0])}
            self.single.append(L)
        w.ada_out, w.out = lin("norm_out.linear"), lin("proj_out")
        self.w = w
        self._ws = {}
        self._rope_cache = {}
        self.profile = None  # set to a dict to collect (start, stop) HIP event pairs per kernel family of the large launches
        # True (default): the weight-times-token linears of the latent / joint stream (Q|K, out, ff1, ff2; single blocks: Q|K, proj_mlp,
        # proj_out) keep a copy packed in MFMA-fragment order (alg_pack_b_p11) and run GEMM schedule 11 (bit-identical to schedule 10);
        # False, or ALG_GEMM_PIPE set to another schedule than 10: the row-major weights.  The prompt stream's few rows stay as they are.
        self.packed_weights = True

    @classmethod
    def from_synthetic(cls, config=None, seed=1234, device="cuda"):
        config = config or HunyuanVideoTransformerConfig()
        return cls(config, synthetic_state_dict(config, seed=seed, device=device), device=device)

    @classmethod
    def from_pretrained(cls, path, subfolder="transformer", torch_dtype=BF, device="cuda", **_):
        root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        cfg_path = os.path.join(root, "config.json")
        if not os.path.exists(cfg_path):
            raise FileNotFoundError("%s not found: weights must be on local disk (no network access in this build; use "
                                    "HunyuanVideoTransformer3DModel.from_synthetic)" % cfg_path)
        with open(cfg_path) as f:
            raw = json.load(f)
        fields = HunyuanVideoTransformerConfig.__dataclass_fields__
        cfg = HunyuanVideoTransformerConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items()
                                               if k in fields})
        from .weights import read_shards
        sd = read_shards(root)
        return cls(cfg, sd, device=device)

    def to(self, *a, **k):
        return self

    def _timed(self, name, fn, *a, **k):
        """the launch, bracketed by a HIP event pair under `name` when `profile` is a dict (bench / tests)"""
        if self.profile is None:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.profile.setdefault(name, []).append((e0, e1))
        return r

    def rope_tables(self, F_, H, W):
        key = (F_, H, W)
        hit = self._rope_cache.get(key)
        if hit is None:
            cfg = self.config
            sizes = (F_ // cfg.patch_size_t, H // cfg.patch_size, W // cfg.patch_size)
            grids = torch.meshgrid(*[torch.arange(0, n, dtype=torch.float32) for n in sizes], indexing="ij")
            cos, sin = [], []
            for dim, grid in zip(cfg.rope_axes_dim, grids):
                freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
                ang = torch.outer(grid.reshape(-1), freqs)
                cos.append(ang.cos().repeat_interleave(2, dim=1))
                sin.append(ang.sin().repeat_interleave(2, dim=1))
            hit = (torch.cat(cos, dim=1).float().to(self.device).contiguous(),
                   torch.cat(sin, dim=1).float().to(self.device).contiguous())
            self._rope_cache[key] = hit
        return hit

    def _workspace(self, N, S, L):
        key = (N, S, L)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            cfg, dev = self.config, self.device
            D, M = cfg.dim, int(cfg.dim * cfg.mlp_ratio)
            J = S + L
            e = lambda *s, dt=BF: torch.empty(*s, dtype=dt, device=dev)
            z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
            ws = SimpleNamespace(J=J, J_pad=(J + 63) // 64 * 64, L_pad=(L + 63) // 64 * 64)
            ws.patches = e(N, S, self.k_patch)
            ws.x, ws.y = e(N, J, D), e(N, J, D)
            ws.qk = e(N, J, 2 * D)
            ws.vt = z(N, D, ws.J_pad)
            ws.am = e(N, J, D + M)                       # [attention out | mlp hidden] per token (single-block proj_out input)
            ws.tok = e(N, S, self.w.out[0].shape[0])
            ws.tsin = e(2 * N, 256)
            ws.h1, ws.tA, ws.tB = e(2 * N, D), e(2 * N, D), e(2 * N, D)
            ws.emb2, ws.semb2 = e(N, 2, D), e(N, 2, D)    # [token-replace embedding, timestep embedding] per sample
            ws.semb1 = e(N, D)
            ws.mod = e(N, 2, 6 * D)
            ws.modc = e(N, 6 * D)
            ws.mod_out = e(N, 2 * D)
            # token refiner
            ws.pool, ws.rt = e(N, cfg.text_embed_dim), e(N, D)
            ws.e, ws.en = e(N, L, D), e(N, L, D)
            ws.eqk, ws.evt, ws.ea = e(N, L, 2 * D), z(N, D, ws.L_pad), e(N, L, D)
            ws.eh = e(N, L, M)
            ws.rgate = e(N, 2 * D)
            self._ws[key] = ws
        return ws

    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                 guidance=None, attention_kwargs=None, return_dict=True):
        cfg, w, G, T = self.config, self.w, _lib.gemm, self._timed
        if hidden_states.device.type != "cuda":
            raise _lib.AlgHipError("HunyuanVideoTransformer3DModel needs device tensors; there is no CPU fallback")
        N, C, F_, H, W = hidden_states.shape
        p = cfg.patch_size
        if C != cfg.in_channels or H % p or W % p:
            raise ValueError(f"hidden_states must be [N, {cfg.in_channels}, F, H, W] with H, W divisible by {p}")
        if cfg.guidance_embeds and guidance is None:
            raise ValueError("this checkpoint has a guidance embedder: pass `guidance`")
        D, M, heads = cfg.dim, int(cfg.dim * cfg.mlp_ratio), cfg.num_attention_heads
        S, first = F_ * (H // p) * (W // p), (H // p) * (W // p)
        if S % 4:
            raise NotImplementedError("the joint [latents; text] layout needs a multiple of 4 latent tokens")
        L = encoder_hidden_states.shape[1]
        ws = self._workspace(N, S, L)
        J, dev = ws.J, self.device
        tr = cfg.image_condition_type == "token_replace"
        cos, sin = self.rope_tables(F_, H, W)
        scale = 1.0 / math.sqrt(cfg.attention_head_dim)
        valid_t = encoder_attention_mask.to(device=dev).float().sum(dim=1).to(torch.int32).contiguous()
        valid = valid_t.tolist()                                   # one small D2H per forward (key lengths are host scalars)
        if min(valid) < 1:
            raise ValueError("every prompt needs at least one unmasked token")
        hs = hidden_states.to(BF).contiguous()
        txt = encoder_hidden_states.to(device=dev, dtype=BF).contiguous()
        pooled = pooled_projections.to(device=dev, dtype=BF).contiguous()
        t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        t = (t.expand(N) if t.numel() == 1 else t).contiguous()

        def temb_of(tvals, l1, l2, out, rows):
            """Timesteps(256) -> Linear -> SiLU -> Linear for `rows` scalars -> out [rows, D]."""
            _lib.timestep_embedding(tvals, ws.tsin, rows, 256, True)
            G(ws.tsin, l1[0], ws.h1, rows, D, 256, 256, 256, D, bias=l1[1], act=_lib.ACT_SILU)
            G(ws.h1, l2[0], out, rows, D, D, D, D, D, bias=l2[1])

        # ---- conditioning: temb (and the timestep-0 "token replace" embedding) ----
        temb_of(torch.cat([torch.zeros_like(t), t]).contiguous(), w.t1, w.t2, ws.tA, 2 * N)      # rows [0..N) = t 0, [N..2N) = t
        G(pooled, w.p1[0], ws.h1, N, D, cfg.pooled_projection_dim, cfg.pooled_projection_dim, cfg.pooled_projection_dim, D,
          bias=w.p1[1], act=_lib.ACT_SILU)
        G(ws.h1, w.p2[0], ws.tB, N, D, D, D, D, D, bias=w.p2[1])                                   # pooled projection
        pooled_e = ws.tB[:N]
        emb_t = _lib.lincomb([(1.0, ws.tA[N:2 * N].contiguous()), (1.0, pooled_e.contiguous())], BF)
        if cfg.guidance_embeds:
            g_in = guidance.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            gbuf = torch.empty(N, D, dtype=BF, device=dev)
            _lib.timestep_embedding(g_in, ws.tsin, N, 256, True)
            G(ws.tsin, w.g1[0], ws.h1, N, D, 256, 256, 256, D, bias=w.g1[1], act=_lib.ACT_SILU)
            G(ws.h1, w.g2[0], gbuf, N, D, D, D, D, D, bias=w.g2[1])
            emb_t = _lib.lincomb([(1.0, emb_t), (1.0, gbuf)], BF)
        emb_tr = _lib.lincomb([(1.0, ws.tA[:N].contiguous()), (1.0, pooled_e.contiguous())], BF) if tr else emb_t
        ws.emb2[:, 0].copy_(emb_tr)
        ws.emb2[:, 1].copy_(emb_t)
        _lib.silu(ws.emb2, ws.semb2)
        _lib.silu(emb_t.contiguous(), ws.semb1)

        # ---- patch embed into the latent rows of the joint buffer ----
        _lib.patchify3d(hs, ws.patches, N, C, F_, H, W, p, p, self.k_patch)
        G(ws.patches, w.patch_w, ws.x, S, D, self.k_patch, self.k_patch, self.k_patch, D, bias=w.patch_b, batch=N,
          strideA=S * self.k_patch, strideC=J * D)

        # ---- token refiner on the prompt tokens -> text rows of the joint buffer ----
        Td = cfg.text_embed_dim
        _lib.masked_mean(txt, valid_t, ws.pool, N, L, Td)
        temb_of(t, w.r_t1, w.r_t2, ws.tA, N)
        G(ws.pool, w.r_x1[0], ws.h1, N, D, Td, Td, Td, D, bias=w.r_x1[1], act=_lib.ACT_SILU)
        G(ws.h1, w.r_x2[0], ws.tB, N, D, D, D, D, D, bias=w.r_x2[1])
        r_temb = _lib.lincomb([(1.0, ws.tA[:N].contiguous()), (1.0, ws.tB[:N].contiguous())], BF)
        _lib.silu(r_temb, ws.rt)
        G(txt, w.r_in[0], ws.e, N * L, D, Td, Td, Td, D, bias=w.r_in[1])
        n_ref = len(self.refiner)
        for li, R in enumerate(self.refiner):
            last = li == n_ref - 1
            G(ws.rt, R.ada[0], ws.rgate, N, 2 * D, D, D, D, 2 * D, bias=R.ada[1])           # gate_msa | gate_mlp
            _lib.layernorm_modulate(ws.e, ws.en, R.n1w, R.n1b, None, None, 0, N, L, D, 0, 1e-6)
            G(ws.en, R.wqk, ws.eqk, N * L, 2 * D, D, D, D, 2 * D, bias=R.bqk)
            G(R.v[0], ws.en, ws.evt, D, L, D, D, D, ws.L_pad, bias=R.v[1], batch=N, strideB=L * D, strideC=D * ws.L_pad,
              flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            for b in range(N):
                _lib.flash_attn_d128(ws.eqk, ws.eqk, ws.evt, ws.ea, 1, heads, L, valid[b], L * 2 * D, 2 * D, L * 2 * D, 2 * D,
                                     D * ws.L_pad, ws.L_pad, L * D, D, scale, q_off=b * L * 2 * D, k_off=b * L * 2 * D + D,
                                     vt_off=b * D * ws.L_pad, o_off=b * L * D)
            G(ws.ea, R.o[0], ws.e, L, D, D, D, D, D, bias=R.o[1], R=ws.e, ldr=D, gate=ws.rgate, strideGate=2 * D,
              gate_seg_stride=0, batch=N, strideA=L * D, strideC=L * D, strideR=L * D)
            _lib.layernorm_modulate(ws.e, ws.en, R.n2w, R.n2b, None, None, 0, N, L, D, 0, 1e-6)
            G(ws.en, R.f1[0], ws.eh, N * L, M, D, D, D, M, bias=R.f1[1], act=_lib.ACT_SILU)
            # the last refiner block writes the refined prompt straight into the text rows of the joint buffer
            G(ws.eh, R.f2[0], ws.x if last else ws.e, L, D, M, M, M, D, bias=R.f2[1], R=ws.e, ldr=D, gate=ws.rgate,
              gate_off=D, strideGate=2 * D, gate_seg_stride=0, batch=N, strideA=L * M, strideC=(J if last else L) * D,
              strideR=L * D, c_off=(S * D if last else 0))
        if n_ref == 0:
            ws.x[:, S:].copy_(ws.e)

        mod_bs, seg = (12 * D, 6 * D) if tr else (6 * D, 0)
        split = first if tr else 0

        def attention():
            prof = self.profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for b in range(N):
                _lib.flash_attn_d128(ws.qk, ws.qk, ws.vt, ws.am, 1, heads, J, S + valid[b], J * 2 * D, 2 * D, J * 2 * D, 2 * D,
                                     D * ws.J_pad, ws.J_pad, J * (D + M), D + M, scale, q_off=b * J * 2 * D,
                                     k_off=b * J * 2 * D + D, vt_off=b * D * ws.J_pad, o_off=b * J * (D + M))
            if prof is not None:
                e1.record()
                prof.setdefault("attn_self", []).append((e0, e1))

        def ada(lin_, rows_in, out, n_out, two):
            """AdaLN linear on silu(emb): `two` -> both embeddings of every sample ([N][2][n_out]), else temb only."""
            G(rows_in, lin_[0], out, (2 * N if two else N), n_out, D, D, D, n_out, bias=lin_[1])

        # ---- dual-stream blocks: latent rows [0, S) and text rows [S, J) of the joint buffer ----
        AM = D + M
        use_packed = self.packed_weights and os.environ.get("ALG_GEMM_PIPE", "10") == "10"
        PK = lambda Lw_, t: (Lw_.packed.get(id(t)) or t) if use_packed else t     # the packed copy of a block's weight, if it has one
        for Lw in self.dual:
            ada(Lw.ada, ws.semb2 if tr else ws.semb1, ws.mod, 6 * D, tr)     # [N][2][6D] (token replace) or [N][6D]
            mv = ws.mod
            ada(Lw.ada_c, ws.semb1, ws.modc, 6 * D, False)
            # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp at +0, +D, ... of every 6D vector
            T("ln_mod", _lib.layernorm_modulate_seg, ws.x, ws.y, None, None, mv, mv, mod_bs, seg, N, S, D, split, 1e-6,
              x_bstride=J * D, y_bstride=J * D, scale_off=D, shift_off=0)
            _lib.layernorm_modulate_seg(ws.x, ws.y, None, None, ws.modc, ws.modc, 6 * D, 0, N, L, D, 0, 1e-6,
                                        x_bstride=J * D, y_bstride=J * D, x_off=S * D, y_off=S * D, scale_off=D, shift_off=0)
            T("gemm_qk", G, ws.y, PK(Lw, Lw.wqk), ws.qk, S, 2 * D, D, D, D, 2 * D, bias=Lw.bqk, batch=N, strideA=J * D, strideC=J * 2 * D)
            G(ws.y, Lw.wqk_c, ws.qk, L, 2 * D, D, D, D, 2 * D, bias=Lw.bqk_c, batch=N, strideA=J * D, strideC=J * 2 * D,
              a_off=S * D, c_off=S * 2 * D)
            T("gemm_vt", G, Lw.v[0], ws.y, ws.vt, D, S, D, D, D, ws.J_pad, bias=Lw.v[1], batch=N, strideB=J * D,
              strideC=D * ws.J_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            G(Lw.v_c[0], ws.y, ws.vt, D, L, D, D, D, ws.J_pad, bias=Lw.v_c[1], batch=N, strideB=J * D,
              strideC=D * ws.J_pad, b_off=S * D, perm_col0=S, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            # latent rows: norm_q / norm_k + rope; prompt rows: norm_added_q / norm_added_k, no rope
            T("headnorm_rope", _lib.headnorm_rope_, ws.qk, Lw.nq, cos, sin, 2 * D, J * 2 * D, N, S, heads, S, 1e-6)
            T("headnorm_rope", _lib.headnorm_rope_, ws.qk, Lw.nk, cos, sin, 2 * D, J * 2 * D, N, S, heads, S, 1e-6, x_off=D)
            _lib.headnorm_rope_(ws.qk, Lw.nq_c, None, None, 2 * D, J * 2 * D, N, L, heads, 0, 1e-6, x_off=S * 2 * D)
            _lib.headnorm_rope_(ws.qk, Lw.nk_c, None, None, 2 * D, J * 2 * D, N, L, heads, 0, 1e-6, x_off=S * 2 * D + D)
            attention()
            T("gemm_out", G, ws.am, PK(Lw, Lw.o[0]), ws.x, S, D, D, AM, D, D, bias=Lw.o[1], R=ws.x, ldr=D, gate=mv, gate_off=2 * D,
              strideGate=mod_bs, gate_seg_stride=seg, seg_split=split, batch=N, strideA=J * AM, strideC=J * D, strideR=J * D)
            G(ws.am, Lw.o_c[0], ws.x, L, D, D, AM, D, D, bias=Lw.o_c[1], R=ws.x, ldr=D, gate=ws.modc, gate_off=2 * D,
              strideGate=6 * D, gate_seg_stride=0, batch=N, strideA=J * AM, strideC=J * D, strideR=J * D, a_off=S * AM,
              c_off=S * D, r_off=S * D)
            T("ln_mod", _lib.layernorm_modulate_seg, ws.x, ws.y, None, None, mv, mv, mod_bs, seg, N, S, D, split, 1e-6,
              x_bstride=J * D, y_bstride=J * D, scale_off=4 * D, shift_off=3 * D)
            _lib.layernorm_modulate_seg(ws.x, ws.y, None, None, ws.modc, ws.modc, 6 * D, 0, N, L, D, 0, 1e-6,
                                        x_bstride=J * D, y_bstride=J * D, x_off=S * D, y_off=S * D, scale_off=4 * D,
                                        shift_off=3 * D)
            T("gemm_ff1", G, ws.y, PK(Lw, Lw.f1[0]), ws.am, S, M, D, D, D, AM, bias=Lw.f1[1], act=_lib.ACT_GELU_TANH, batch=N,
              strideA=J * D, strideC=J * AM, c_off=D)
            G(ws.y, Lw.f1_c[0], ws.am, L, M, D, D, D, AM, bias=Lw.f1_c[1], act=_lib.ACT_GELU_TANH, batch=N, strideA=J * D,
              strideC=J * AM, a_off=S * D, c_off=S * AM + D)
            T("gemm_ff2", G, ws.am, PK(Lw, Lw.f2[0]), ws.x, S, D, M, AM, M, D, bias=Lw.f2[1], R=ws.x, ldr=D, gate=mv, gate_off=5 * D,
              strideGate=mod_bs, gate_seg_stride=seg, seg_split=split, batch=N, strideA=J * AM, strideC=J * D, strideR=J * D,
              a_off=D)
            G(ws.am, Lw.f2_c[0], ws.x, L, D, M, AM, M, D, bias=Lw.f2_c[1], R=ws.x, ldr=D, gate=ws.modc, gate_off=5 * D,
              strideGate=6 * D, gate_seg_stride=0, batch=N, strideA=J * AM, strideC=J * D, strideR=J * D, a_off=S * AM + D,
              c_off=S * D, r_off=S * D)

        # ---- single-stream blocks, in place on the joint buffer ----
        smod_bs, sseg = (6 * D, 3 * D) if tr else (3 * D, 0)
        for Lw in self.single:
            if tr:
                G(ws.semb2, Lw.ada[0], ws.mod, 2 * N, 3 * D, D, D, D, 3 * D, bias=Lw.ada[1])       # [N][2][3D]: shift, scale, gate
            else:
                G(ws.semb1, Lw.ada[0], ws.mod, N, 3 * D, D, D, D, 3 * D, bias=Lw.ada[1])
            T("ln_mod", _lib.layernorm_modulate_seg, ws.x, ws.y, None, None, ws.mod, ws.mod, smod_bs, sseg, N, J, D, split, 1e-6,
              scale_off=D, shift_off=0)
            T("gemm_ff1", G, ws.y, PK(Lw, Lw.mlp[0]), ws.am, N * J, M, D, D, D, AM, bias=Lw.mlp[1], act=_lib.ACT_GELU_TANH, c_off=D)
            T("gemm_qk", G, ws.y, PK(Lw, Lw.wqk), ws.qk, N * J, 2 * D, D, D, D, 2 * D, bias=Lw.bqk)
            T("gemm_vt", G, Lw.v[0], ws.y, ws.vt, D, J, D, D, D, ws.J_pad, bias=Lw.v[1], batch=N, strideB=J * D,
              strideC=D * ws.J_pad, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
            T("headnorm_rope", _lib.headnorm_rope_, ws.qk, Lw.nq, cos, sin, 2 * D, J * 2 * D, N, J, heads, S, 1e-6)
            T("headnorm_rope", _lib.headnorm_rope_, ws.qk, Lw.nk, cos, sin, 2 * D, J * 2 * D, N, J, heads, S, 1e-6, x_off=D)
            attention()
            T("gemm_out_mlp", G, ws.am, PK(Lw, Lw.out[0]), ws.x, J, D, AM, AM, AM, D, bias=Lw.out[1], R=ws.x, ldr=D, gate=ws.mod, gate_off=2 * D,
              strideGate=smod_bs, gate_seg_stride=sseg, seg_split=split, batch=N, strideA=J * AM, strideC=J * D,
              strideR=J * D)

        # ---- output head: AdaLayerNormContinuous (scale | shift), projection, unpatchify ----
        G(ws.semb1, w.ada_out[0], ws.mod_out, N, 2 * D, D, D, D, 2 * D, bias=w.ada_out[1])
        _lib.layernorm_modulate_seg(ws.x, ws.y, None, None, ws.mod_out, ws.mod_out, 2 * D, 0, N, S, D, 0, 1e-6,
                                    x_bstride=J * D, y_bstride=J * D, scale_off=0, shift_off=D)
        n_out = w.out[0].shape[0]
        G(ws.y, w.out[0], ws.tok, S, n_out, D, D, D, n_out, bias=w.out[1], batch=N, strideA=J * D, strideC=S * n_out)
        out = torch.empty(N, cfg.out_channels, F_, H, W, dtype=BF, device=dev)
        _lib.unpatchify3d(ws.tok, n_out, out, N, cfg.out_channels, F_, H, W, p, p, channel_major=True)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
