"""Plain-PyTorch CPU restatement of CogVideoXTransformer3DModel.forward (the DiT the
reference calls at pipeline_cogvideox_image2video_lowpass.py:1082-1090) and of the RoPE table
helper it feeds (cog:542-584, crop helper cog:76-91).

TEST INFRASTRUCTURE -- never imported by ``alg_amd``.

PARITY UNPINNED: the arithmetic lives in diffusers @ git be2fb77 (requirements.txt:13), which is
neither under /root/reference nor installed; the reference holds no test or golden vector for it.
This file restates the published architecture and is the
self-consistency checker for the HIP kernels (HIP bf16 vs this fp32/fp64 restatement).

Weights are a flat ``dict[str, Tensor]`` whose keys are the diffusers state-dict names
(``transformer_blocks.0.attn1.to_q.weight`` ...), so a real checkpoint maps 1:1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    """Fields of CogVideoXTransformer3DModel.config the pipeline reads (cog:549-556, 899-901, 964,
    973, 993, 998) plus the architecture sizes.  Defaults = THUDM/CogVideoX-5b-I2V."""

    num_attention_heads: int = 48
    attention_head_dim: int = 64
    in_channels: int = 32
    out_channels: int = 16
    num_layers: int = 42
    time_embed_dim: int = 512
    text_embed_dim: int = 4096
    max_text_seq_length: int = 226
    sample_width: int = 90
    sample_height: int = 60
    sample_frames: int = 49
    patch_size: int = 2
    patch_size_t: object = None
    temporal_compression_ratio: int = 4
    ff_inner_mult: int = 4
    norm_eps: float = 1e-5
    qk_norm_eps: float = 1e-6
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    use_rotary_positional_embeddings: bool = True
    use_learned_positional_embeddings: bool = True
    ofs_embed_dim: object = None
    spatial_interpolation_scale: float = 1.875
    temporal_interpolation_scale: float = 1.0

    @property
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    @property
    def ff_dim(self):
        return self.ff_inner_mult * self.inner_dim


def param_shapes(cfg: DiTConfig):
    """name -> shape for every parameter/buffer, in diffusers state-dict naming."""
    D, H = cfg.inner_dim, cfg.attention_head_dim
    p = cfg.patch_size
    lat_f = (cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1
    n_patch = (cfg.sample_height // p) * (cfg.sample_width // p) * lat_f
    p_t = cfg.patch_size_t
    s = {
        "patch_embed.proj.weight": (D, cfg.in_channels, p, p) if p_t is None else (D, cfg.in_channels * p * p * p_t),
        "patch_embed.proj.bias": (D,),
        "patch_embed.text_proj.weight": (D, cfg.text_embed_dim),
        "patch_embed.text_proj.bias": (D,),
        "time_embedding.linear_1.weight": (cfg.time_embed_dim, D),
        "time_embedding.linear_1.bias": (cfg.time_embed_dim,),
        "time_embedding.linear_2.weight": (cfg.time_embed_dim, cfg.time_embed_dim),
        "time_embedding.linear_2.bias": (cfg.time_embed_dim,),
        "norm_final.weight": (D,),
        "norm_final.bias": (D,),
        "norm_out.linear.weight": (2 * D, cfg.time_embed_dim),
        "norm_out.linear.bias": (2 * D,),
        "norm_out.norm.weight": (D,),
        "norm_out.norm.bias": (D,),
        "proj_out.weight": (p * p * (p_t or 1) * cfg.out_channels, D),
        "proj_out.bias": (p * p * (p_t or 1) * cfg.out_channels,),
    }
    if cfg.ofs_embed_dim is not None:
        O = cfg.ofs_embed_dim
        s.update({"ofs_embedding.linear_1.weight": (O, O), "ofs_embedding.linear_1.bias": (O,),
                  "ofs_embedding.linear_2.weight": (O, O), "ofs_embedding.linear_2.bias": (O,)})
    if cfg.use_learned_positional_embeddings:
        s["patch_embed.pos_embedding"] = (1, cfg.max_text_seq_length + n_patch, D)
    for i in range(cfg.num_layers):
        b = "transformer_blocks.%d." % i
        for nm in ("norm1", "norm2"):
            s[b + nm + ".linear.weight"] = (6 * D, cfg.time_embed_dim)
            s[b + nm + ".linear.bias"] = (6 * D,)
            s[b + nm + ".norm.weight"] = (D,)
            s[b + nm + ".norm.bias"] = (D,)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            s[b + "attn1." + nm + ".weight"] = (D, D)
            s[b + "attn1." + nm + ".bias"] = (D,)
        for nm in ("norm_q", "norm_k"):
            s[b + "attn1." + nm + ".weight"] = (H,)
            s[b + "attn1." + nm + ".bias"] = (H,)
        s[b + "ff.net.0.proj.weight"] = (cfg.ff_dim, D)
        s[b + "ff.net.0.proj.bias"] = (cfg.ff_dim,)
        s[b + "ff.net.2.weight"] = (D, cfg.ff_dim)
        s[b + "ff.net.2.bias"] = (D,)
    return s


def count_params(cfg: DiTConfig):
    return sum(math.prod(v) for v in param_shapes(cfg).values())


def init_weights(cfg: DiTConfig, seed=1234, std=0.02, dtype=torch.float32, randomize_affine=False):
    """Synthetic seeded weights: N(0, std^2) matrices, zero biases, unit norm gains.
    ``randomize_affine`` perturbs biases/gains/pos-emb so tests exercise every term."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".weight") and len(shape) >= 2:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif name.endswith("pos_embedding"):
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif name.endswith(".weight"):  # norm gains
            t = torch.ones(shape)
            if randomize_affine:
                t = t + 0.1 * torch.randn(shape, generator=g)
        else:  # biases
            t = torch.zeros(shape)
            if randomize_affine:
                t = 0.05 * torch.randn(shape, generator=g)
        w[name] = t.to(dtype)
    return w


# ---------------------------------------------------------------------------------------------
# RoPE tables (cog:542-584 -> diffusers get_3d_rotary_pos_embed, 'linspace' grid)
# ---------------------------------------------------------------------------------------------


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """cog:76-91."""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height = th
        resize_width = int(round(th / h * w))
    else:
        resize_width = tw
        resize_height = int(round(tw / w * h))
    crop_top = int(round((th - resize_height) / 2.0))
    crop_left = int(round((tw - resize_width) / 2.0))
    return (crop_top, crop_left), (crop_top + resize_height, crop_left + resize_width)


def _rope_1d(dim, pos, theta=10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    f = torch.outer(pos, freqs)
    return f.cos().repeat_interleave(2, dim=1).float(), f.sin().repeat_interleave(2, dim=1).float()


def rope_tables(cfg: DiTConfig, height, width, latent_frames, vae_scale_factor_spatial=8):
    """(cos, sin), each [tokens_t*gh*gw, head_dim] fp32: CogVideoX 1.0 (cog:558-569: cropped linspace grid) or, with
    `patch_size_t`, CogVideoX 1.5 (cog:570-582: `grid_type="slice"`, integer positions, (F + p_t - 1) // p_t time steps)."""
    p = cfg.patch_size
    gh = height // (vae_scale_factor_spatial * p)
    gw = width // (vae_scale_factor_spatial * p)
    base_w = cfg.sample_width // p
    base_h = cfg.sample_height // p
    if cfg.patch_size_t is None:
        start, stop = get_resize_crop_region_for_grid((gh, gw), base_w, base_h)
        grid_h = torch.linspace(start[0], stop[0] * (gh - 1) / gh, gh, dtype=torch.float32)
        grid_w = torch.linspace(start[1], stop[1] * (gw - 1) / gw, gw, dtype=torch.float32)
    else:
        # get_3d_rotary_pos_embed(grid_type="slice", max_size=(base_h, base_w)): tables over arange(max), sliced to the grid
        latent_frames = (latent_frames + cfg.patch_size_t - 1) // cfg.patch_size_t
        grid_h = torch.arange(base_h, dtype=torch.float32)[:gh]
        grid_w = torch.arange(base_w, dtype=torch.float32)[:gw]
        if grid_h.numel() != gh or grid_w.numel() != gw:
            raise ValueError("the grid exceeds the configured sample size (slice rotary embedding)")
    grid_t = torch.arange(latent_frames, dtype=torch.float32)
    d = cfg.attention_head_dim
    dim_t, dim_h, dim_w = d // 4, d // 8 * 3, d // 8 * 3
    tc, ts = _rope_1d(dim_t, grid_t)
    hc, hs = _rope_1d(dim_h, grid_h)
    wc, ws = _rope_1d(dim_w, grid_w)

    def comb(t, h, w):
        t = t[:, None, None, :].expand(-1, gh, gw, -1)
        h = h[None, :, None, :].expand(latent_frames, -1, gw, -1)
        w = w[None, None, :, :].expand(latent_frames, gh, -1, -1)
        return torch.cat([t, h, w], dim=-1).reshape(latent_frames * gh * gw, -1).contiguous()

    return comb(tc, hc, wc), comb(ts, hs, ws)


def apply_rotary(x, cos, sin):
    """diffusers apply_rotary_emb(use_real=True, unbind_dim=-1); x [B, H, S, D]."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


# ---------------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------------


def timestep_sinusoid(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def sincos_pos_embed_3d(cfg: DiTConfig, gh, gw, lat_f):
    """Non-learned joint positional embedding (diffusers get_3d_sincos_pos_embed); only used when
    use_learned_positional_embeddings is False (CogVideoX-2B style).  Returned [1, T+P, D]."""
    D = cfg.inner_dim
    ds, dt = 3 * D // 4, D // 4

    def sc1(dim, pos):
        omega = torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0)
        omega = 1.0 / 10000 ** omega
        out = torch.outer(pos.reshape(-1).double(), omega)
        return torch.cat([out.sin(), out.cos()], dim=1)

    gh_ = torch.arange(gh, dtype=torch.float32) / cfg.spatial_interpolation_scale
    gw_ = torch.arange(gw, dtype=torch.float32) / cfg.spatial_interpolation_scale
    gt_ = torch.arange(lat_f, dtype=torch.float32) / cfg.temporal_interpolation_scale
    grid = torch.stack(torch.meshgrid(gw_, gh_, indexing="xy"), dim=0).reshape(2, 1, gh, gw)
    emb_h = sc1(ds // 2, grid[0])
    emb_w = sc1(ds // 2, grid[1])
    sp = torch.cat([emb_h, emb_w], dim=1)  # [gh*gw, ds]
    tp = sc1(dt, gt_)  # [lat_f, dt]
    sp = sp[None].expand(lat_f, -1, -1)
    tp = tp[:, None].expand(-1, gh * gw, -1)
    pe = torch.cat([tp, sp], dim=-1).reshape(lat_f * gh * gw, D).float()
    joint = torch.zeros(1, cfg.max_text_seq_length + pe.shape[0], D)
    joint[:, cfg.max_text_seq_length:] = pe
    return joint


def dit_forward(cfg: DiTConfig, w, hidden_states, encoder_hidden_states, timestep, image_rotary_emb=None,
                collect=None, ofs=None):
    """hidden_states [B, F, C_in, H, W]; encoder_hidden_states [B, T, text_dim]; timestep [B];
    image_rotary_emb (cos, sin) or None.  Returns [B, F, C_out, H, W] in the compute dtype of ``w``.
    ``collect`` (dict) receives named intermediates for kernel-level tests."""
    dt = w["proj_out.weight"].dtype
    x_in = hidden_states.to(dt)
    B, Fr, C, Hh, Ww = x_in.shape
    D, nh, hd, p = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim, cfg.patch_size
    T = encoder_hidden_states.shape[1]

    # 1. time embedding (Timesteps -> TimestepEmbedding: linear, SiLU, linear)
    t_emb = timestep_sinusoid(timestep, D, cfg.flip_sin_to_cos, cfg.freq_shift).to(dt)
    emb = F.linear(t_emb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
    if cfg.ofs_embed_dim is not None:
        # CogVideoX 1.5: ofs_emb = ofs_embedding(ofs_proj(ofs)); emb = emb + ofs_emb
        o = timestep_sinusoid(ofs.reshape(-1).float().expand(B) if ofs.numel() == 1 else ofs.float(), cfg.ofs_embed_dim,
                              cfg.flip_sin_to_cos, cfg.freq_shift).to(dt)
        o = F.linear(o, w["ofs_embedding.linear_1.weight"], w["ofs_embedding.linear_1.bias"])
        o = F.linear(F.silu(o), w["ofs_embedding.linear_2.weight"], w["ofs_embedding.linear_2.bias"])
        emb = emb + o

    # 2. patch embedding (CogVideoXPatchEmbed): Conv2d k=p s=p per frame (patch_size_t None) or a Linear over
    #    (c, p_t, p, p) blocks of p_t frames (CogVideoX 1.5); text Linear
    txt = F.linear(encoder_hidden_states.to(dt), w["patch_embed.text_proj.weight"], w["patch_embed.text_proj.bias"])
    p_t = cfg.patch_size_t
    if p_t is None:
        img = F.conv2d(x_in.reshape(-1, C, Hh, Ww), w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=p)
        img = img.view(B, Fr, D, -1).transpose(2, 3).flatten(1, 2)  # [B, F*gh*gw, D]
    else:
        img = x_in.permute(0, 1, 3, 4, 2)                                          # [B, F, H, W, C]
        img = img.reshape(B, Fr // p_t, p_t, Hh // p, p, Ww // p, p, C)
        img = img.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)     # [B, tokens, C * p_t * p * p]
        img = F.linear(img, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"])
    x = torch.cat([txt, img], dim=1)
    if cfg.use_learned_positional_embeddings:
        if cfg.sample_width != Ww or cfg.sample_height != Hh:
            raise ValueError("learned positional embeddings need the configured sample height/width")
        pre_frames = (Fr - 1) * cfg.temporal_compression_ratio + 1
        if pre_frames == cfg.sample_frames:
            pos = w["patch_embed.pos_embedding"]
        else:
            pos = sincos_pos_embed_3d(cfg, Hh // p, Ww // p, Fr)
        x = x + pos.to(dt)
    elif not cfg.use_rotary_positional_embeddings:
        x = x + sincos_pos_embed_3d(cfg, Hh // p, Ww // p, Fr).to(dt)
    if collect is not None:
        collect["embed"] = x.clone()
        collect["temb"] = emb.clone()

    S = x.shape[1]
    semb = F.silu(emb)
    for i in range(cfg.num_layers):
        b = "transformer_blocks.%d." % i

        def ln_zero(nm, x):
            mod = F.linear(semb, w[b + nm + ".linear.weight"], w[b + nm + ".linear.bias"])
            shift, scale, gate, eshift, escale, egate = mod.chunk(6, dim=1)
            n = F.layer_norm(x, (D,), w[b + nm + ".norm.weight"], w[b + nm + ".norm.bias"], cfg.norm_eps)
            nt = n[:, :T] * (1 + escale)[:, None] + eshift[:, None]
            nv = n[:, T:] * (1 + scale)[:, None] + shift[:, None]
            g = torch.cat([egate[:, None].expand(-1, T, -1), gate[:, None].expand(-1, S - T, -1)], dim=1)
            return torch.cat([nt, nv], dim=1), g

        # attention
        n1, g1 = ln_zero("norm1", x)
        q = F.linear(n1, w[b + "attn1.to_q.weight"], w[b + "attn1.to_q.bias"]).view(B, S, nh, hd).transpose(1, 2)
        k = F.linear(n1, w[b + "attn1.to_k.weight"], w[b + "attn1.to_k.bias"]).view(B, S, nh, hd).transpose(1, 2)
        v = F.linear(n1, w[b + "attn1.to_v.weight"], w[b + "attn1.to_v.bias"]).view(B, S, nh, hd).transpose(1, 2)
        q = F.layer_norm(q, (hd,), w[b + "attn1.norm_q.weight"], w[b + "attn1.norm_q.bias"], cfg.qk_norm_eps)
        k = F.layer_norm(k, (hd,), w[b + "attn1.norm_k.weight"], w[b + "attn1.norm_k.bias"], cfg.qk_norm_eps)
        if image_rotary_emb is not None:
            cos, sin = image_rotary_emb
            q = torch.cat([q[:, :, :T], apply_rotary(q[:, :, T:], cos, sin)], dim=2)
            k = torch.cat([k[:, :, :T], apply_rotary(k[:, :, T:], cos, sin)], dim=2)
        if collect is not None and i == 0:
            collect["n1_0"], collect["q_0"], collect["k_0"], collect["v_0"] = n1.clone(), q.clone(), k.clone(), v.clone()
        a = F.scaled_dot_product_attention(q, k, v)
        a = a.transpose(1, 2).reshape(B, S, D)
        if collect is not None and i == 0:
            collect["attn_0"] = a.clone()
        a = F.linear(a, w[b + "attn1.to_out.0.weight"], w[b + "attn1.to_out.0.bias"])
        x = x + g1 * a
        # feed-forward (GELU tanh)
        n2, g2 = ln_zero("norm2", x)
        h = F.gelu(F.linear(n2, w[b + "ff.net.0.proj.weight"], w[b + "ff.net.0.proj.bias"]), approximate="tanh")
        h = F.linear(h, w[b + "ff.net.2.weight"], w[b + "ff.net.2.bias"])
        x = x + g2 * h
        if collect is not None:
            collect["block_%d" % i] = x.clone()

    # final norm on the concatenated sequence, then slice the video tokens (5B branch)
    x = F.layer_norm(x, (D,), w["norm_final.weight"], w["norm_final.bias"], cfg.norm_eps)[:, T:]
    # AdaLayerNorm(chunk_dim=1): shift, scale = linear(silu(temb)).chunk(2)
    mod = F.linear(semb, w["norm_out.linear.weight"], w["norm_out.linear.bias"])
    shift, scale = mod.chunk(2, dim=1)
    x = F.layer_norm(x, (D,), w["norm_out.norm.weight"], w["norm_out.norm.bias"], cfg.norm_eps)
    x = x * (1 + scale)[:, None] + shift[:, None]
    x = F.linear(x, w["proj_out.weight"], w["proj_out.bias"])
    # unpatchify
    if p_t is None:
        out = x.reshape(B, Fr, Hh // p, Ww // p, -1, p, p)
        out = out.permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    else:
        out = x.reshape(B, (Fr + p_t - 1) // p_t, Hh // p, Ww // p, -1, p_t, p, p)
        out = out.permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)
    return out


def flops_per_forward(cfg: DiTConfig, tokens):
    """Algorithmic FLOPs of one sample-forward: L*(24*S*d^2 + 4*S^2*d) + small terms."""
    d, L = cfg.inner_dim, cfg.num_layers
    ffm = cfg.ff_inner_mult
    per_layer = 2 * tokens * d * (4 * d + 2 * ffm * d) + 4 * tokens * tokens * d
    vid = tokens - cfg.max_text_seq_length
    small = 2 * vid * (cfg.in_channels * cfg.patch_size ** 2) * d + 2 * cfg.max_text_seq_length * cfg.text_embed_dim * d \
        + 2 * vid * d * cfg.patch_size ** 2 * cfg.out_channels
    return L * per_layer + small
