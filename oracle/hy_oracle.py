"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch, fp32 by default) of the HunyuanVideo DiT forward that
pipeline_hunyuan_video_image2video_lowpass.py:1243-1252 calls.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import it.

The module lives in the `diffusers` dependency (requirements.txt pins git @ be2fb77dc164083bf8f033874066c96bc0a75a11; not
in /root/reference, not installable here): HunyuanVideoTransformer3DModel with its token refiner, 20 dual-stream and 40
single-stream blocks, per-head RMSNorm on q/k, 3-axis RoPE (16, 56, 56) on the latent tokens, AdaLayerNormZero
modulation, optional guidance embedding and the "token_replace" image conditioning (first-frame tokens are modulated
and gated with the timestep-0 embedding).  Restated from the published architecture; PARITY UNPINNED.  State-dict names
follow diffusers.

Padded text tokens (encoder_attention_mask == 0) are masked as KEYS everywhere in the published model, so they never
influence a latent or a valid text token; their own rows are carried along but not part of the contract (tests compare
the latent output only).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class HyConfig:
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


def param_shapes(cfg: HyConfig):
    D, M = cfg.dim, int(cfg.dim * cfg.mlp_ratio)
    p, pt = cfg.patch_size, cfg.patch_size_t
    s = {}

    def lin(name, n_out, n_in):
        s[name + ".weight"] = (n_out, n_in)
        s[name + ".bias"] = (n_out,)

    s["x_embedder.proj.weight"] = (D, cfg.in_channels, pt, p, p)
    s["x_embedder.proj.bias"] = (D,)
    # token refiner
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
    # condition embedding
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


def init_weights(cfg: HyConfig, seed=0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.prod(shape[1:]) ** 0.5
        sd[name] = t.to(dtype)
    return sd


def rope_tables(cfg: HyConfig, F_, H, W):
    """HunyuanVideoRotaryPosEmbed: (cos, sin) [S, 128] float32, each axis' frequencies repeat-interleaved by 2."""
    p, pt = cfg.patch_size, cfg.patch_size_t
    sizes = (F_ // pt, H // p, W // p)
    grids = torch.meshgrid(*[torch.arange(0, n, dtype=torch.float32) for n in sizes], indexing="ij")
    cos, sin = [], []
    for dim, grid in zip(cfg.rope_axes_dim, grids):
        freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(grid.reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


def _timesteps(t):
    half = 128
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _rms_head(x, w, dt, eps=1e-6):
    var = x.float().pow(2).mean(-1, keepdim=True)
    return (x.float() * torch.rsqrt(var + eps)).to(dt) * w.to(dt)


def _rope(x, cos, sin):
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos.to(x.device) + rot.float() * sin.to(x.device)).to(x.dtype)


def _sdpa(q, k, v, kv_len, max_scores=1 << 28):
    """q, k, v [B, H, S, d]; keys >= kv_len[b] are masked.  Rows of a softmax are independent: long sequences are evaluated in
    (batch, head, query-block) pieces of at most `max_scores` scores -- the same arithmetic per row, bounded memory (the C4 token
    count has 1.4e10 scores per head)."""
    B, H, S, d = q.shape
    Skv = k.shape[2]
    idx = torch.arange(Skv, device=q.device)[None, None, None, :]
    if B * H * S * Skv <= max_scores:
        s = q.float() @ k.float().transpose(-1, -2) / math.sqrt(d)
        s = s.masked_fill(idx >= kv_len.view(B, 1, 1, 1), float("-inf"))
        return (torch.softmax(s, dim=-1) @ v.float()).to(q.dtype)
    out = torch.empty(B, H, S, d, dtype=q.dtype, device=q.device)
    rows = max(1, max_scores // Skv)
    for b in range(B):
        dead = idx[0, 0] >= kv_len[b]                                     # [1, Skv]
        for h in range(H):
            kf, vf = k[b, h].float(), v[b, h].float()
            for r0 in range(0, S, rows):
                s = q[b, h, r0:r0 + rows].float() @ kf.t() / math.sqrt(d)
                out[b, h, r0:r0 + rows] = (torch.softmax(s.masked_fill(dead, float("-inf")), dim=-1) @ vf).to(q.dtype)
    return out


def patch_embed(x, weight, bias, patch):
    """Conv3d(kernel = stride = patch)(x).flatten(2).transpose(1, 2) written as a linear layer over the unfolded patches (rows
    (c, dt, dy, dx): the Conv3d weight's own memory order) -- the formulation the product's patchify + GEMM uses.  Pinned against
    F.conv3d itself in tests/test_oracle_patch_embed_cpu.py."""
    B, C, F_, H, Wd = x.shape
    pt, ph, pw = patch
    hp = x.reshape(B, C, F_ // pt, pt, H // ph, ph, Wd // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return F.linear(hp.reshape(B, -1, C * pt * ph * pw), weight.reshape(weight.shape[0], -1), bias)


def hy_forward(cfg: HyConfig, sd, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask,
               pooled_projections, guidance=None, dtype=torch.float32):
    """hidden_states [B, C, F, H, W]; timestep [B]; text [B, L, text_dim]; mask [B, L] (prefix of ones);
    pooled [B, pooled_dim]; guidance [B] or None.  Returns [B, out_channels, F, H, W]."""
    W_ = lambda n: sd[n].to(dtype)
    lin = lambda x, n: F.linear(x, W_(n + ".weight"), W_(n + ".bias"))
    ln = lambda x, w=None, b=None: F.layer_norm(x, (x.shape[-1],), w, b, 1e-6)
    B, C, F_, H, Wd = hidden_states.shape
    p, pt = cfg.patch_size, cfg.patch_size_t
    D, heads, hd = cfg.dim, cfg.num_attention_heads, cfg.attention_head_dim
    S = (F_ // pt) * (H // p) * (Wd // p)
    first = (H // p) * (Wd // p)
    token_replace = cfg.image_condition_type == "token_replace"
    cos, sin = rope_tables(cfg, F_, H, Wd)
    heads_view = lambda t: t.unflatten(2, (heads, hd)).transpose(1, 2)

    # ---- condition embedding ----
    te = "time_text_embed."
    temb_of = lambda t, pre: lin(F.silu(lin(_timesteps(t).to(dtype), pre + ".linear_1")), pre + ".linear_2")
    pooled = lin(F.silu(lin(pooled_projections.to(dtype), te + "text_embedder.linear_1")), te + "text_embedder.linear_2")
    temb = temb_of(timestep, te + "timestep_embedder") + pooled
    tr_emb = temb_of(torch.zeros_like(timestep), te + "timestep_embedder") + pooled if token_replace else None
    if cfg.guidance_embeds:
        temb = temb + temb_of(guidance, te + "guidance_embedder")

    # ---- patch embed ----
    # x_embedder = Conv3d(kernel = stride = patch): one dot product per output voxel over its own patch, i.e. a linear layer over
    # the unfolded patches (rows (c, dt, dy, dx): the weight's own memory order)
    x = patch_embed(hidden_states.to(dtype), W_("x_embedder.proj.weight"), W_("x_embedder.proj.bias"), (pt, p, p))

    # ---- token refiner ----
    ce = "context_embedder."
    txt = encoder_hidden_states.to(dtype)
    valid = encoder_attention_mask.to(torch.int64).sum(dim=1)
    mf = encoder_attention_mask.float().unsqueeze(-1)
    pooled_txt = ((txt.float() * mf).sum(dim=1) / mf.sum(dim=1)).to(dtype)
    r_temb = temb_of(timestep, ce + "time_text_embed.timestep_embedder") + lin(
        F.silu(lin(pooled_txt, ce + "time_text_embed.text_embedder.linear_1")), ce + "time_text_embed.text_embedder.linear_2")
    e = lin(txt, ce + "proj_in")
    for l in range(cfg.num_refiner_layers):
        b = ce + f"token_refiner.refiner_blocks.{l}."
        n1 = ln(e, W_(b + "norm1.weight"), W_(b + "norm1.bias"))
        q, k, v = (heads_view(lin(n1, b + "attn." + n)) for n in ("to_q", "to_k", "to_v"))
        a = _sdpa(q, k, v, valid).transpose(1, 2).flatten(2, 3)      # padded queries: rows outside the contract
        a = lin(a, b + "attn.to_out.0")
        gate_msa, gate_mlp = lin(F.silu(r_temb), b + "norm_out.linear").chunk(2, dim=1)
        e = e + a * gate_msa.unsqueeze(1)
        f = lin(F.silu(lin(ln(e, W_(b + "norm2.weight"), W_(b + "norm2.bias")), b + "ff.net.0.proj")), b + "ff.net.2")
        e = e + f * gate_mlp.unsqueeze(1)
    L = e.shape[1]
    kv_len = S + valid

    def mod_ln(x_, shift, scale, shift_tr=None, scale_tr=None):
        n = ln(x_)
        if shift_tr is None:
            return n * (1 + scale[:, None]) + shift[:, None]
        return torch.cat([n[:, :first] * (1 + scale_tr[:, None]) + shift_tr[:, None],
                          n[:, first:] * (1 + scale[:, None]) + shift[:, None]], dim=1)

    def gated(x_, y, gate, gate_tr=None):
        if gate_tr is None:
            return x_ + y * gate.unsqueeze(1)
        return torch.cat([x_[:, :first] + y[:, :first] * gate_tr.unsqueeze(1),
                          x_[:, first:] + y[:, first:] * gate.unsqueeze(1)], dim=1)

    def joint_attention(q, k, v):
        """q, k, v [B, S + L, D] (latent tokens first); per-head RMSNorm is applied by the caller."""
        qh, kh, vh = heads_view(q), heads_view(k), heads_view(v)
        qh = torch.cat([_rope(qh[:, :, :S], cos, sin), qh[:, :, S:]], dim=2)
        kh = torch.cat([_rope(kh[:, :, :S], cos, sin), kh[:, :, S:]], dim=2)
        return _sdpa(qh, kh, vh, kv_len).transpose(1, 2).flatten(2, 3)

    hn = lambda t, w: _rms_head(t.unflatten(2, (heads, hd)), sd[w], dtype).flatten(2, 3)

    # ---- dual-stream blocks ----
    for l in range(cfg.num_layers):
        b = f"transformer_blocks.{l}."
        m = lin(F.silu(temb), b + "norm1.linear").chunk(6, dim=1)       # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        mt = lin(F.silu(tr_emb), b + "norm1.linear").chunk(6, dim=1) if token_replace else [None] * 6
        c = lin(F.silu(temb), b + "norm1_context.linear").chunk(6, dim=1)
        nx = mod_ln(x, m[0], m[1], mt[0], mt[1])
        ne = mod_ln(e, c[0], c[1])
        q = torch.cat([hn(lin(nx, b + "attn.to_q"), b + "attn.norm_q.weight"),
                       hn(lin(ne, b + "attn.add_q_proj"), b + "attn.norm_added_q.weight")], dim=1)
        k = torch.cat([hn(lin(nx, b + "attn.to_k"), b + "attn.norm_k.weight"),
                       hn(lin(ne, b + "attn.add_k_proj"), b + "attn.norm_added_k.weight")], dim=1)
        v = torch.cat([lin(nx, b + "attn.to_v"), lin(ne, b + "attn.add_v_proj")], dim=1)
        a = joint_attention(q, k, v)
        x = gated(x, lin(a[:, :S], b + "attn.to_out.0"), m[2], mt[2])
        e = gated(e, lin(a[:, S:], b + "attn.to_add_out"), c[2])
        nx = mod_ln(x, m[3], m[4], mt[3], mt[4])
        ne = mod_ln(e, c[3], c[4])
        x = gated(x, lin(F.gelu(lin(nx, b + "ff.net.0.proj"), approximate="tanh"), b + "ff.net.2"), m[5], mt[5])
        e = gated(e, lin(F.gelu(lin(ne, b + "ff_context.net.0.proj"), approximate="tanh"), b + "ff_context.net.2"), c[5])

    # ---- single-stream blocks on the joint sequence [latents; text] ----
    for l in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{l}."
        z = torch.cat([x, e], dim=1)
        m = lin(F.silu(temb), b + "norm.linear").chunk(3, dim=1)        # shift, scale, gate
        mt = lin(F.silu(tr_emb), b + "norm.linear").chunk(3, dim=1) if token_replace else [None] * 3
        nz = mod_ln(z, m[0], m[1], mt[0], mt[1])
        mlp = F.gelu(lin(nz, b + "proj_mlp"), approximate="tanh")
        q = hn(lin(nz, b + "attn.to_q"), b + "attn.norm_q.weight")
        k = hn(lin(nz, b + "attn.to_k"), b + "attn.norm_k.weight")
        a = joint_attention(q, k, lin(nz, b + "attn.to_v"))
        out = lin(torch.cat([a, mlp], dim=2), b + "proj_out")
        if token_replace:
            z = torch.cat([z[:, :first] + out[:, :first] * mt[2].unsqueeze(1),
                           z[:, first:] + out[:, first:] * m[2].unsqueeze(1)], dim=1)
        else:
            z = z + out * m[2].unsqueeze(1)
        x, e = z[:, :S], z[:, S:]

    # ---- output head: AdaLayerNormContinuous (scale first, then shift) ----
    scale, shift = lin(F.silu(temb), "norm_out.linear").chunk(2, dim=1)
    x = ln(x) * (1 + scale)[:, None] + shift[:, None]
    x = lin(x, "proj_out")
    x = x.reshape(B, F_ // pt, H // p, Wd // p, -1, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
