"""CogVideoXTransformer3DModel forward on hand-written gfx950 kernels.

Mirrors the component protocol the reference loop requires of its injected transformer:

    transformer(hidden_states=, encoder_hidden_states=, timestep=, ofs=, image_rotary_emb=,
                attention_kwargs=, return_dict=False)[0]        (pipeline_cogvideox_image2video_lowpass.py:1082-1090)
    transformer.config.{sample_height, sample_width, sample_frames, patch_size, patch_size_t, in_channels,
                        use_rotary_positional_embeddings, ofs_embed_dim, attention_head_dim}, transformer.dtype

The arithmetic follows diffusers' CogVideoXTransformer3DModel (diffusers @ be2fb77, not in the reference
tree -- parity unpinned); the weight dict uses the diffusers state-dict names so a real
checkpoint maps 1:1.  Host code here only owns buffers and launch order: every FLOP and every byte of the
forward moves through libalg_hip.so (GEMM + epilogues, flash attention, LayerNorm/AdaLN, QK-norm + RoPE,
patch gather / unpatchify, timestep sinusoid).

HBM layout (all bf16, allocated once per batch size N and reused every step):
    x      [N, S, D]        residual stream, S = text_len + patches, text tokens first
    y      [N, S, D]        LayerNorm+modulate output (GEMM A operand)
    qk     [N, S, 2, H, 64] fused Q|K projection, normalised + rotated in place
    vt     [N, H*64, S_pad] V^T written directly by a transposed GEMM (kv order permuted for the MFMA B operand)
    att    [N, S, D]        attention output (A operand of the out-projection)
    h      [N, S, 4D]       GELU(FF1)
    mod    [N, L*12D + 4D]  every AdaLN shift/scale/gate vector of the forward, one GEMM
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, asdict
from typing import Optional

import torch

from . import _lib


@dataclass
class CogVideoXTransformerConfig:
    """Defaults = THUDM/CogVideoX-5b-I2V."""

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
    patch_size_t: Optional[int] = None
    temporal_compression_ratio: int = 4
    ff_inner_mult: int = 4
    norm_eps: float = 1e-5
    qk_norm_eps: float = 1e-6
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    use_rotary_positional_embeddings: bool = True
    use_learned_positional_embeddings: bool = True
    ofs_embed_dim: Optional[int] = None
    spatial_interpolation_scale: float = 1.875
    temporal_interpolation_scale: float = 1.0

    @property
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    def to_dict(self):
        return asdict(self)


def _bf(t, device):
    return t.to(device=device, dtype=torch.bfloat16).contiguous()


class CogVideoXTransformer3DModel:
    dtype = torch.bfloat16

    def __init__(self, config: CogVideoXTransformerConfig, weights: dict, device="cuda"):
        cfg = self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AlgHipError("CogVideoXTransformer3DModel runs on the GPU only (HIP kernels, no CPU fallback)")
        if cfg.attention_head_dim != 64:
            raise _lib.AlgHipError("the attention kernel is specialised for head_dim 64")
        if cfg.ofs_embed_dim is not None and cfg.ofs_embed_dim != cfg.time_embed_dim:
            raise _lib.AlgHipError("ofs_embed_dim must equal time_embed_dim (emb = emb + ofs_emb)")
        D = cfg.inner_dim
        if D % 512:
            raise _lib.AlgHipError("inner_dim must be a multiple of 512 (LayerNorm kernel tiling)")
        _lib.load_library()
        # True (default): the softmax scale * log2(e) rides in Q's last rounding and the attention takes log2-unit scores
        # (alg_qk_norm_rope_scaled + ALG_ATTN_Q_PRESCALED); False keeps the per-score multiply (A/B and parity tests)
        self.attn_prescale = True
        # True (default): the Q|K projection and the transposed V projection of a block go out as one alg_gemm_bf16_pair
        # launch (bit-identical to the two launches; False keeps them apart: A/B runs, tests)
        self.pair_qkv = True
        # True: the per-head QK LayerNorm + rotary embedding runs inside the Q|K projection's store loop (alg_gemm_bf16_pair_qk;
        # bit-identical to the stand-alone kernel behind the pair launch, one read + one write of qk less).  Off by default:
        # measured on the same box (profiles/r4_fuse_qk_norm_ab.txt) the ~20 VALU operations per element cost more in a
        # one-wave-per-SIMD store loop (+9.5 ms per step) than the stand-alone kernel at full occupancy (8.4 ms per step)
        self.fuse_qk_norm = False
        # True (default): the attention-output and feed-forward weights are also kept packed in MFMA-fragment order
        # (alg_pack_b_p11, once, here) and those three GEMMs run schedule 11 (1 x 4 waves, the weight straight from L2 into registers;
        # bit-identical to schedule 10).  False, or ALG_GEMM_PIPE set to another schedule than 10: the row-major weights (A/B runs).
        # The Q|K projection stays on the pair launch with V^T (whose B operand is the activation): on schedule 11 with V^T as its own
        # launch the step was 0.4 % slower (profiles/r6_bench_steps10_ab_packed_qk.json).
        self.packed_weights = True
        self._sincos = {}
        dev = self.device
        w = weights
        p = cfg.patch_size
        self.p_t = cfg.patch_size_t or 1   # CogVideoX 1.5: Linear patch embed over (c, t, py, px), frames folded in p_t
        self.k_patch = cfg.in_channels * p * p * self.p_t
        if self.k_patch % 64 or cfg.text_embed_dim % 64 or cfg.time_embed_dim % 64:
            raise _lib.AlgHipError("GEMM K dimensions must be multiples of 64")
        self.w_patch = _bf(w["patch_embed.proj.weight"].reshape(D, self.k_patch), dev)
        self.b_patch = _bf(w["patch_embed.proj.bias"], dev)
        self.w_text = _bf(w["patch_embed.text_proj.weight"], dev)
        self.b_text = _bf(w["patch_embed.text_proj.bias"], dev)
        self.pos_emb = _bf(w["patch_embed.pos_embedding"][0], dev) if cfg.use_learned_positional_embeddings else None
        self.w_t1 = _bf(w["time_embedding.linear_1.weight"], dev)
        self.b_t1 = _bf(w["time_embedding.linear_1.bias"], dev)
        self.w_t2 = _bf(w["time_embedding.linear_2.weight"], dev)
        self.b_t2 = _bf(w["time_embedding.linear_2.bias"], dev)
        if cfg.ofs_embed_dim is not None:  # CogVideoX 1.5: emb = time_embedding(t) + ofs_embedding(Timesteps(ofs))
            self.w_o1, self.b_o1 = _bf(w["ofs_embedding.linear_1.weight"], dev), _bf(w["ofs_embedding.linear_1.bias"], dev)
            self.w_o2, self.b_o2 = _bf(w["ofs_embedding.linear_2.weight"], dev), _bf(w["ofs_embedding.linear_2.bias"], dev)

        # AdaLN linears of all layers + norm_out, concatenated into one [TOT, time_embed_dim] weight.  Rows are
        # re-ordered from diffusers' chunk order (shift, scale, gate, enc_shift, enc_scale, enc_gate) to
        # (shift_txt, shift_vid, scale_txt, scale_vid, gate_txt, gate_vid) so each quantity is a [2][D] pair
        # indexed by segment (0 = text rows, 1 = video rows), which is what the kernels consume.
        def reorder6(t):
            c = t.chunk(6, dim=0)
            return torch.cat([c[3], c[0], c[4], c[1], c[5], c[2]], dim=0)

        mw, mb = [], []
        self.layers = []
        for i in range(cfg.num_layers):
            b = "transformer_blocks.%d." % i
            L = {}
            for nm in ("norm1", "norm2"):
                mw.append(reorder6(w[b + nm + ".linear.weight"]))
                mb.append(reorder6(w[b + nm + ".linear.bias"]))
                L[nm + "_w"] = _bf(w[b + nm + ".norm.weight"], dev)
                L[nm + "_b"] = _bf(w[b + nm + ".norm.bias"], dev)
            L["wqk"] = _bf(torch.cat([w[b + "attn1.to_q.weight"], w[b + "attn1.to_k.weight"]], dim=0), dev)
            L["bqk"] = _bf(torch.cat([w[b + "attn1.to_q.bias"], w[b + "attn1.to_k.bias"]], dim=0), dev)
            L["wv"] = _bf(w[b + "attn1.to_v.weight"], dev)
            L["bv"] = _bf(w[b + "attn1.to_v.bias"], dev)
            for nm in ("norm_q", "norm_k"):
                L[nm + "_w"] = _bf(w[b + "attn1." + nm + ".weight"], dev)
                L[nm + "_b"] = _bf(w[b + "attn1." + nm + ".bias"], dev)
            L["wo"] = _bf(w[b + "attn1.to_out.0.weight"], dev)
            L["bo"] = _bf(w[b + "attn1.to_out.0.bias"], dev)
            L["wf1"] = _bf(w[b + "ff.net.0.proj.weight"], dev)
            L["bf1"] = _bf(w[b + "ff.net.0.proj.bias"], dev)
            L["wf2"] = _bf(w[b + "ff.net.2.weight"], dev)
            L["bf2"] = _bf(w[b + "ff.net.2.bias"], dev)
            for nm in ("wo", "wf1", "wf2"):
                L["p" + nm] = _lib.PackedB(L[nm])
            self.layers.append(L)
        # norm_out: chunk order (shift, scale) -> (shift, shift, scale, scale) so both segments are valid
        ow, ob = w["norm_out.linear.weight"], w["norm_out.linear.bias"]
        sh_w, sc_w = ow.chunk(2, dim=0)
        sh_b, sc_b = ob.chunk(2, dim=0)
        mw.append(torch.cat([sh_w, sh_w, sc_w, sc_w], dim=0))
        mb.append(torch.cat([sh_b, sh_b, sc_b, sc_b], dim=0))
        self.w_mod = _bf(torch.cat(mw, dim=0), dev)
        self.b_mod = _bf(torch.cat(mb, dim=0), dev)
        self.mod_cols = self.w_mod.shape[0]  # L*12D + 4D
        self.norm_final_w = _bf(w["norm_final.weight"], dev)
        self.norm_final_b = _bf(w["norm_final.bias"], dev)
        self.norm_out_w = _bf(w["norm_out.norm.weight"], dev)
        self.norm_out_b = _bf(w["norm_out.norm.bias"], dev)
        self.w_out = _bf(w["proj_out.weight"], dev)
        self.b_out = _bf(w["proj_out.bias"], dev)
        self._ws = {}
        self._rope_cache = {}
        self.profile = None  # set to a dict to collect (start, stop) HIP event pairs per kernel family

    def _positional(self, Hh, Ww, Fr, S):
        """The joint [text; video] positional embedding added behind the patch embedding, [S, D] bf16, or None.
        CogVideoXPatchEmbed (the transformer cog:1082 calls): the learned table when the frame count is the configured one,
        otherwise -- and for checkpoints with neither learned nor rotary embeddings (CogVideoX-2B style) -- the 3-D sincos
        embedding computed for the actual grid (`get_3d_sincos_pos_embed`: a quarter of the width for time, three quarters for
        (h, w), positions divided by the interpolation scales), zero rows for the text tokens."""
        cfg = self.config
        learned = cfg.use_learned_positional_embeddings
        if not learned and cfg.use_rotary_positional_embeddings:
            return None
        pre_frames = (Fr - 1) * cfg.temporal_compression_ratio + 1
        if learned and pre_frames == cfg.sample_frames and self.pos_emb is not None and self.pos_emb.shape[0] == S:
            return self.pos_emb
        p = cfg.patch_size
        gh, gw, gt = Hh // p, Ww // p, Fr // self.p_t
        key = (gh, gw, gt)
        hit = self._sincos.get(key)
        if hit is None:
            D = cfg.inner_dim
            ds, dt = 3 * D // 4, D // 4

            def sc1(dim, pos):
                omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
                out = torch.outer(pos.reshape(-1).double(), omega)
                return torch.cat([out.sin(), out.cos()], dim=1)

            hh = torch.arange(gh, dtype=torch.float32) / cfg.spatial_interpolation_scale
            ww = torch.arange(gw, dtype=torch.float32) / cfg.spatial_interpolation_scale
            tt = torch.arange(gt, dtype=torch.float32) / cfg.temporal_interpolation_scale
            grid = torch.stack(torch.meshgrid(ww, hh, indexing="xy"), dim=0).reshape(2, 1, gh, gw)
            sp = torch.cat([sc1(ds // 2, grid[0]), sc1(ds // 2, grid[1])], dim=1)            # [gh * gw, ds]
            tp = sc1(dt, tt)                                                                   # [gt, dt]
            pe = torch.cat([tp[:, None].expand(-1, gh * gw, -1), sp[None].expand(gt, -1, -1)], dim=-1).reshape(gt * gh * gw, D)
            joint = torch.zeros(cfg.max_text_seq_length + pe.shape[0], D, dtype=torch.float32)
            joint[cfg.max_text_seq_length:] = pe.float()
            hit = joint.to(device=self.device, dtype=torch.bfloat16).contiguous()
            self._sincos[key] = hit
        if hit.shape[0] != S:
            raise ValueError("positional embedding covers %d tokens, the sequence has %d" % (hit.shape[0], S))
        return hit

    def _timed(self, name, fn, *args, **kwargs):
        """Launch ``fn``; when profiling is on, bracket it with HIP events on the launch stream."""
        if self.profile is None:
            return fn(*args, **kwargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kwargs)
        e1.record()
        self.profile.setdefault(name, []).append((e0, e1))
        return out

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config=None, seed=1234, std=0.02, device="cuda", randomize_affine=False):
        """Seeded synthetic weights at the configured shapes, generated on the device tensor by tensor (the
        real checkpoint cannot be downloaded here).  Matrices N(0, std^2), biases 0, norm gains 1."""
        from .weights import synthetic_state_dict

        config = config or CogVideoXTransformerConfig()
        return cls(config, synthetic_state_dict(config, seed=seed, std=std, device=device,
                                                 randomize_affine=randomize_affine), device=device)

    @classmethod
    def from_pretrained(cls, path, subfolder="transformer", torch_dtype=torch.bfloat16, device="cuda", **_):
        """Load a diffusers-format checkpoint directory (config.json + *.safetensors) from local disk."""
        from .weights import load_diffusers_transformer

        config, sd = load_diffusers_transformer(path, subfolder)
        return cls(config, sd, device=device)

    def to(self, *args, **kwargs):
        return self

    # ------------------------------------------------------------------------------------------------
    def _workspace(self, N, S, P):
        key = (N, S, P)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        cfg, dev = self.config, self.device
        D = cfg.inner_dim
        bf = dict(device=dev, dtype=torch.bfloat16)
        S_pad = (S + 127) // 128 * 128  # V^T rows: S rounded up to the widest KV stage of the attention variants
        ws = dict(
            S_pad=S_pad,
            x=torch.empty(N, S, D, **bf),
            y=torch.empty(N, S, D, **bf),
            qk=torch.empty(N, S, 2 * D, **bf),
            vt=torch.zeros(N, D, S_pad, **bf),  # pad columns stay zero (stores are guarded)
            att=torch.empty(N, S, D, **bf),
            h=torch.empty(N, S, cfg.ff_inner_mult * D, **bf),
            patches=torch.empty(N, P, self.k_patch, **bf),
            tsin=torch.empty(N, D, **bf),
            t1=torch.empty(N, cfg.time_embed_dim, **bf),
            semb=torch.empty(N, cfg.time_embed_dim, **bf),
            mod=torch.empty(N, self.mod_cols, **bf),
            po=torch.empty(N * P, self.w_out.shape[0], **bf),
        )
        if len(self._ws) >= 4:  # 2-pass and 3-pass shapes both stay resident; anything older is dropped
            self._ws.pop(next(iter(self._ws)))
        self._ws[key] = ws
        return ws

    def _rope(self, image_rotary_emb):
        if image_rotary_emb is None:
            return None, None
        cos, sin = image_rotary_emb
        key = (cos.data_ptr(), sin.data_ptr(), tuple(cos.shape))
        hit = self._rope_cache.get(key)
        if hit is None:
            hit = (cos.to(device=self.device, dtype=torch.float32).contiguous(),
                   sin.to(device=self.device, dtype=torch.float32).contiguous())
            self._rope_cache = {key: hit}
        return hit

    # ------------------------------------------------------------------------------------------------
    def forward_assembled(self, latents, conds, encoder_hidden_states, timestep, image_rotary_emb=None, ofs=None):
        """The loop's form of the forward, with the CFG batch assembly (cog:1060-1070) folded into the patch
        gather: sample n sees channels [latents | conds[n]].

        latents [1 or N, F, C, H, W] bf16; conds: list of N tensors [1, F, C, H, W] (or [F, C, H, W]) bf16;
        encoder_hidden_states [N, T, text_dim]; timestep [N].  Returns noise prediction [N, F, C_out, H, W] bf16.
        """
        cfg = self.config
        N = len(conds)
        lat = latents.to(device=self.device, dtype=torch.bfloat16).contiguous()
        Bl, Fr, C, Hh, Ww = lat.shape
        if Bl not in (1, N):
            raise ValueError("latents batch must be 1 or len(conds)")
        if 2 * C != cfg.in_channels:
            raise ValueError("latents have %d channels, transformer expects in_channels=%d" % (C, cfg.in_channels))
        conds = [c.to(device=self.device, dtype=torch.bfloat16).contiguous() for c in conds]
        for c in conds:
            if c.numel() != Fr * C * Hh * Ww:
                raise ValueError("conditioning latents must have shape [1, F, C, H, W] matching the latents")
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        if ehs.shape[0] != N:
            raise ValueError("encoder_hidden_states batch %d != %d samples" % (ehs.shape[0], N))
        T = ehs.shape[1]
        p, p_t = cfg.patch_size, self.p_t
        if Fr % p_t:
            raise ValueError("latent frames (%d) must be a multiple of patch_size_t=%d (cog:963-968 pads them)" % (Fr, p_t))
        P = (Fr // p_t) * (Hh // p) * (Ww // p)
        S = T + P
        D, Hn = cfg.inner_dim, cfg.num_attention_heads
        if cfg.use_learned_positional_embeddings:
            if cfg.sample_width != Ww or cfg.sample_height != Hh:
                raise ValueError(
                    "It is currently not possible to generate videos at a different resolution that the defaults. "
                    "This should only be the case with 'THUDM/CogVideoX-5b-I2V'.")
        pos_emb = self._positional(Hh, Ww, Fr, S)
        ws = self._workspace(N, S, P)
        x, y, qk, vt, att, h, mod = ws["x"], ws["y"], ws["qk"], ws["vt"], ws["att"], ws["h"], ws["mod"]
        S_pad = ws["S_pad"]
        cos, sin = self._rope(image_rotary_emb)
        G = _lib.gemm
        TM = self._timed

        # 1. timestep embedding: sinusoid -> linear_1 + SiLU -> linear_2 (+ SiLU: only silu(temb) is consumed)
        ts = timestep.to(device=self.device, dtype=torch.float32).contiguous()
        if ts.numel() != N:
            ts = ts.reshape(-1)[:1].expand(N).contiguous()
        _lib.timestep_embedding(ts, ws["tsin"], N, D, cfg.flip_sin_to_cos)
        E = cfg.time_embed_dim
        G(ws["tsin"], self.w_t1, ws["t1"], N, E, D, D, D, E, bias=self.b_t1, act=_lib.ACT_SILU)
        if cfg.ofs_embed_dim is None:
            G(ws["t1"], self.w_t2, ws["semb"], N, E, E, E, E, E, bias=self.b_t2, act=_lib.ACT_SILU)
        else:
            if ofs is None:
                raise ValueError("this transformer has an ofs embedding: pass `ofs` (cog:998)")
            O = cfg.ofs_embed_dim
            emb, osin, o1, oemb = (torch.empty(N, E, device=self.device, dtype=torch.bfloat16) for _ in range(4))
            G(ws["t1"], self.w_t2, emb, N, E, E, E, E, E, bias=self.b_t2)
            ofs_v = ofs.to(device=self.device, dtype=torch.float32).reshape(-1)
            ofs_v = (ofs_v if ofs_v.numel() == N else ofs_v[:1].expand(N)).contiguous()
            _lib.timestep_embedding(ofs_v, osin, N, O, cfg.flip_sin_to_cos)
            G(osin, self.w_o1, o1, N, O, O, O, O, O, bias=self.b_o1, act=_lib.ACT_SILU)
            G(o1, self.w_o2, oemb, N, O, O, O, O, O, bias=self.b_o2)
            _lib.lincomb([(1.0, emb), (1.0, oemb)], torch.bfloat16, out=emb)
            _lib.silu(emb, ws["semb"])
        # every AdaLN vector of the forward in one GEMM
        G(ws["semb"], self.w_mod, mod, N, self.mod_cols, E, E, E, self.mod_cols, bias=self.b_mod)

        # 2. patch embedding (+ positional embedding as the residual operand), text tokens first
        G(ehs, self.w_text, x, T, D, cfg.text_embed_dim, cfg.text_embed_dim, cfg.text_embed_dim, D,
          bias=self.b_text, R=pos_emb, ldr=D, batch=N, strideA=T * cfg.text_embed_dim, strideC=S * D)
        _lib.patchify(lat, 0 if Bl == 1 else Fr * C * Hh * Ww, conds, ws["patches"], N, Fr, C, Hh, Ww, p, p_t)
        G(ws["patches"], self.w_patch, x, P, D, self.k_patch, self.k_patch, self.k_patch, D, bias=self.b_patch,
          R=pos_emb, ldr=D, r_off=T * D, batch=N, strideA=P * self.k_patch, strideC=S * D, c_off=T * D)

        # 3. transformer blocks
        scale = 1.0 / math.sqrt(cfg.attention_head_dim)
        prescale = self.attn_prescale
        q_scale = scale * 1.4426950408889634 if prescale else 1.0
        F4 = cfg.ff_inner_mult * D
        packed = self.packed_weights and os.environ.get("ALG_GEMM_PIPE", "10") == "10"
        wo_k, wf1_k, wf2_k = ("pwo", "pwf1", "pwf2") if packed else ("wo", "wf1", "wf2")
        for li, L in enumerate(self.layers):
            m1 = li * 12 * D          # norm1: shift @+0, scale @+2D, gate @+4D (each [2][D])
            m2 = m1 + 6 * D           # norm2
            TM("ln_mod", _lib.layernorm_modulate, x, y, L["norm1_w"], L["norm1_b"], mod, mod, self.mod_cols, N, S, D,
               T, cfg.norm_eps, scale_off=m1 + 2 * D, shift_off=m1)
            qk_call = ((y, L["wqk"], qk, S, 2 * D, D, D, D, 2 * D), dict(bias=L["bqk"], batch=N, strideA=S * D, strideC=S * 2 * D))
            vt_call = ((L["wv"], y, vt, D, S, D, D, D, S_pad), dict(bias=L["bv"], batch=N, strideB=S * D, strideC=D * S_pad,
                                                                   flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS))
            # the softmax scale * log2(e) rides in Q's last rounding (attn_prescale = False: scaled per score in the attention)
            if self.pair_qkv and self.fuse_qk_norm:   # ... and QK LayerNorm + rope in the Q|K store loop: qk is written once
                TM("gemm_qkv", _lib.gemm_pair_qk, qk_call, vt_call, L["norm_q_w"], L["norm_q_b"], L["norm_k_w"], L["norm_k_b"], cos, sin,
                               Hn, T, cfg.qk_norm_eps, q_scale=q_scale)
            else:
                if self.pair_qkv:   # both projections read y: ONE persistent launch, the two partial last rounds become one
                    TM("gemm_qkv", _lib.gemm_pair, qk_call, vt_call)
                else:
                    TM("gemm_qk", G, *qk_call[0], **qk_call[1])
                    TM("gemm_vt", G, *vt_call[0], **vt_call[1])
                TM("qk_norm_rope", _lib.qk_norm_rope_, qk, L["norm_q_w"], L["norm_q_b"], L["norm_k_w"], L["norm_k_b"], cos, sin, N, S, Hn, T,
                                   cfg.qk_norm_eps, q_scale=q_scale)
            TM("attn", _lib.flash_attn_d64, qk, qk, vt, att, N, Hn, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, scale,
                                k_off=D, q_prescaled=prescale)
            TM("gemm_out", G, att, L[wo_k], x, S, D, D, D, D, D, bias=L["bo"], R=x, ldr=D, gate=mod, gate_off=m1 + 4 * D,
              strideGate=self.mod_cols, seg_split=T, batch=N, strideA=S * D, strideC=S * D, strideR=S * D)
            TM("ln_mod", _lib.layernorm_modulate, x, y, L["norm2_w"], L["norm2_b"], mod, mod, self.mod_cols, N, S, D,
               T, cfg.norm_eps, scale_off=m2 + 2 * D, shift_off=m2)
            TM("gemm_ff1", G, y, L[wf1_k], h, S, F4, D, D, D, F4, bias=L["bf1"], act=_lib.ACT_GELU_TANH, batch=N, strideA=S * D,
              strideC=S * F4)
            TM("gemm_ff2", G, h, L[wf2_k], x, S, D, F4, F4, F4, D, bias=L["bf2"], R=x, ldr=D, gate=mod, gate_off=m2 + 4 * D,
              strideGate=self.mod_cols, seg_split=T, batch=N, strideA=S * F4, strideC=S * D, strideR=S * D)

        # 4. norm_final over the joint sequence, AdaLayerNorm on the video tokens, proj_out, unpatchify
        _lib.layernorm_modulate(x, y, self.norm_final_w, self.norm_final_b, None, None, 0, N, S, D, T, cfg.norm_eps)
        mo = cfg.num_layers * 12 * D   # norm_out: shift pair @+0, scale pair @+2D
        _lib.layernorm_modulate(y, att, self.norm_out_w, self.norm_out_b, mod, mod, self.mod_cols, N, P, D, 0,
                                cfg.norm_eps, x_bstride=S * D, y_bstride=P * D, x_off=T * D, scale_off=mo + 2 * D,
                                shift_off=mo)
        n_out = self.w_out.shape[0]
        G(att, self.w_out, ws["po"], N * P, n_out, D, D, D, n_out, bias=self.b_out)
        out = torch.empty(N, Fr, cfg.out_channels, Hh, Ww, device=self.device, dtype=torch.bfloat16)
        _lib.unpatchify(ws["po"], out, N, Fr, cfg.out_channels, Hh, Ww, p, p_t)
        return out

    def __call__(self, hidden_states, encoder_hidden_states, timestep, timestep_cond=None, ofs=None,
                 image_rotary_emb=None, attention_kwargs=None, return_dict=True):
        """diffusers-style entry: hidden_states [N, F, 2C, H, W] (latents and condition already concatenated on
        the channel axis, cog:1068-1070)."""
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond is not used by the CogVideoX I2V checkpoints")
        C = hidden_states.shape[2] // 2
        lat = hidden_states[:, :, :C].contiguous()
        conds = [hidden_states[n:n + 1, :, C:].contiguous() for n in range(hidden_states.shape[0])]
        out = self.forward_assembled(lat, conds, encoder_hidden_states, timestep, image_rotary_emb, ofs=ofs)
        if not return_dict:
            return (out,)
        return TransformerOutput(sample=out)


@dataclass
class TransformerOutput:
    sample: torch.Tensor
