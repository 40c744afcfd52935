"""HunyuanVideo's prompt encoder on MI355X -- transformers' `LlavaForConditionalGeneration` (llava-llama-3-8b: a CLIP
ViT-L/14-336 vision tower, a two-layer projector, a Llama-3-8B decoder) as the reference pipeline calls it
(`pipeline_hunyuan_video_image2video_lowpass.py:282-420`):

    self.text_encoder(input_ids=, attention_mask=, position_ids=, pixel_values=, output_hidden_states=True)
        .hidden_states[-(num_hidden_layers_to_skip + 1)]          and  .config.image_token_index / .config.pad_token_id / .dtype

Same call signature, transformers state-dict names (4.48 `language_model.model.*` / `vision_tower.vision_model.*` and the
flattened 5.x `model.language_model.*` / `model.vision_tower.*` alike).  Launch order over the C ABI: `alg_embed_rows`; the
vision tower (`CLIPVisionModel`, 577 tokens: flash attention path) -> projector GEMMs + `alg_gelu_erf`; per decoder layer
`alg_t5_layernorm` (RMSNorm, weight applied after the cast back), one fused q|k GEMM + the V^T GEMM with swapped operands,
`alg_rope_half` (rotate-half RoPE, theta 500000, at `position_ids`), `alg_flash_attn_d128_ex` (causal, grouped-query 32 / 8),
o-projection and down-projection with the residual in the GEMM epilogue, gate GEMM with the SiLU epilogue, `alg_mul_bf16`.
Right-padded prompts only (what the reference's tokenizer call produces): under a causal mask valid tokens never see padding
keys; the rows of padding tokens are outside the contract (the DiT masks them as keys).  Once per video; no torch fallback.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from .image_encoder_clip import CLIPVisionEncoderConfig, CLIPVisionModel

BF = torch.bfloat16


def _clip_l_336():
    return CLIPVisionEncoderConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                                   image_size=336, patch_size=14, hidden_act="quick_gelu")


@dataclass
class LlavaConfig:
    """Defaults = hunyuanvideo-community/HunyuanVideo-I2V `text_encoder/config.json` (llava-llama-3-8b)."""
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128320
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    image_token_index: int = 128257
    pad_token_id: int = 128258
    vision_feature_layer: int = -2
    vision_feature_select_strategy: str = "default"
    projector_hidden_act: str = "gelu"
    vision_config: CLIPVisionEncoderConfig = field(default_factory=_clip_l_336)


@dataclass
class LlavaOutput:
    hidden_states: List[torch.Tensor]
    last_hidden_state: Optional[torch.Tensor] = None


def _normalise_names(sd):
    """transformers 4.48 / 5.x checkpoint names -> language_model.* / vision_tower.* / multi_modal_projector.*"""
    out = {}
    for k, v in sd.items():
        if k.startswith("model."):
            k = k[len("model."):]
        k = k.replace("language_model.model.", "language_model.").replace("vision_tower.vision_model.", "vision_tower.")
        out[k] = v
    return out


class LlavaForConditionalGeneration:
    dtype = BF

    def __init__(self, config: Optional[LlavaConfig] = None, device="cuda", dtype=BF):
        self.config = config or LlavaConfig()
        c = self.config
        if dtype != BF:
            raise ValueError("the HIP encoder computes in bfloat16")
        if c.hidden_size // c.num_attention_heads != 128 or c.num_attention_heads % c.num_key_value_heads or \
                c.hidden_size % 64 or c.intermediate_size % 64 or c.vision_feature_select_strategy != "default" or \
                c.projector_hidden_act != "gelu":
            raise ValueError("unsupported Llava configuration (head_dim 128, 'default' feature selection, GELU projector)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AlgHipError("LlavaForConditionalGeneration runs on the GPU only (HIP kernels, no CPU fallback)")
        self.vision_tower = CLIPVisionModel(c.vision_config, device=device)
        self.w = {}
        self._rope = {}

    def param_shapes(self):
        c = self.config
        D, M = c.hidden_size, c.intermediate_size
        kv = c.num_key_value_heads * 128
        out = {"language_model.embed_tokens.weight": (c.vocab_size, D), "language_model.norm.weight": (D,),
               "multi_modal_projector.linear_1.weight": (D, c.vision_config.hidden_size),
               "multi_modal_projector.linear_1.bias": (D,), "multi_modal_projector.linear_2.weight": (D, D),
               "multi_modal_projector.linear_2.bias": (D,)}
        for i in range(c.num_hidden_layers):
            p = "language_model.layers.%d." % i
            out[p + "self_attn.q_proj.weight"], out[p + "self_attn.o_proj.weight"] = (D, D), (D, D)
            out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = (kv, D), (kv, D)
            out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = (M, D), (M, D)
            out[p + "mlp.down_proj.weight"] = (D, M)
            out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = (D,), (D,)
        for k, v in self.vision_tower.param_shapes().items():
            out["vision_tower." + k] = v
        return out

    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda"):
        self = cls(config, device=device)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.param_shapes().items():
            if "norm" in name and name.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif name.endswith(".bias") or "class_embedding" in name or "position_embedding" in name:
                t = 0.1 * torch.randn(shape, generator=g)
            elif "embed_tokens" in name:
                t = torch.randn(shape, generator=g)
            else:
                t = torch.randn(shape, generator=g) * torch.Size(shape[1:]).numel() ** -0.5
            sd[name] = t.to(BF)
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, subfolder="text_encoder", torch_dtype=BF, device="cuda", **_):
        """transformers-format directory on local disk (`text_encoder/config.json` with `text_config` / `vision_config` +
        safetensors shards)."""
        from .weights import config_from_dict, load_component
        raw, sd, _ = load_component(path, subfolder)
        text = dict(raw.get("text_config") or {})
        fields = LlavaConfig.__dataclass_fields__
        kw = {k: v for k, v in text.items() if k in fields and v is not None}
        kw.update({k: v for k, v in raw.items() if k in fields and v is not None and not isinstance(v, dict)})
        if isinstance(raw.get("vision_config"), dict):
            kw["vision_config"] = config_from_dict(CLIPVisionEncoderConfig, raw["vision_config"])
        return cls(LlavaConfig(**kw), device=device).load_state_dict(sd)

    def load_state_dict(self, sd, strict=True):
        sd = _normalise_names(sd)
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd and "post_layernorm" not in k]
        if missing and strict:
            raise KeyError("missing Llava weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        self.vision_tower.load_state_dict({k[len("vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower.")})
        dev = self.device
        put = lambda t: t.to(dev, BF).contiguous()
        W = {"embed": put(sd["language_model.embed_tokens.weight"]), "norm": put(sd["language_model.norm.weight"]),
             "p1": put(sd["multi_modal_projector.linear_1.weight"]), "p1_b": put(sd["multi_modal_projector.linear_1.bias"]),
             "p2": put(sd["multi_modal_projector.linear_2.weight"]), "p2_b": put(sd["multi_modal_projector.linear_2.bias"])}
        for i in range(self.config.num_hidden_layers):
            p = "language_model.layers.%d." % i
            W[p + "qk"] = put(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"]], 0))
            for n, src in (("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"), ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"),
                           ("down", "mlp.down_proj")):
                W[p + n] = put(sd[p + src + ".weight"])
            W[p + "ln0"], W[p + "ln1"] = put(sd[p + "input_layernorm.weight"]), put(sd[p + "post_attention_layernorm.weight"])
        self.w = W
        return self

    # ---- host logic ------------------------------------------------------------------------------------------------------
    def _rope_tables(self, n_pos):
        """LlamaRotaryEmbedding (default rope): fp32 angles, cos / sin rounded to bf16 like the activations' dtype."""
        if n_pos not in self._rope:
            inv = 1.0 / (self.config.rope_theta ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128))
            freqs = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
            emb = torch.cat([freqs, freqs], dim=-1)
            self._rope = {n_pos: (emb.cos().to(BF).float().to(self.device).contiguous(),
                                  emb.sin().to(BF).float().to(self.device).contiguous())}
        return self._rope[n_pos]

    # ---- forward -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, position_ids=None, pixel_values=None, output_hidden_states=True,
                 return_dict=True, **_):
        if not (torch.is_tensor(input_ids) and input_ids.is_cuda and input_ids.dim() == 2):
            raise _lib.AlgHipError("LlavaForConditionalGeneration: input_ids must be a [B, L] device tensor (HIP-only path)")
        c, W, dev = self.config, self.w, self.device
        B, L = input_ids.shape
        D, M, H, Hk = c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads
        T, KV = B * L, Hk * 128
        ids = input_ids.to(torch.int64).contiguous()
        if attention_mask is None:
            attention_mask = torch.ones_like(ids)
        am = attention_mask.to(dev)
        valid = (am > 0).sum(dim=1)
        if not bool(((torch.arange(L, device=dev)[None, :] < valid[:, None]) == (am > 0)).all()):
            raise NotImplementedError("left- or gap-padded prompts are not built (the reference's tokenizer call pads on the right)")
        if position_ids is None:
            position_ids = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
        pos = position_ids.to(dev, torch.int32).contiguous().view(-1)
        cos, sin = self._rope_tables(max(L, int(pos.max().item()) + 1))

        x = torch.empty(T, D, device=dev, dtype=BF)
        _lib.embed_rows(ids, W["embed"], x)
        if pixel_values is not None:
            vs = self.vision_tower(pixel_values=pixel_values.to(dev), output_hidden_states=True).hidden_states
            feats = vs[c.vision_feature_layer][:, 1:].contiguous()                     # "default": drop the class token
            n_img, Dv = feats.shape[0] * feats.shape[1], feats.shape[2]
            h1 = torch.empty(n_img, D, device=dev, dtype=BF)
            _lib.gemm(feats, W["p1"], h1, n_img, D, Dv, Dv, Dv, D, bias=W["p1_b"])
            _lib.gelu_erf_(h1)
            h2 = torch.empty(n_img, D, device=dev, dtype=BF)
            _lib.gemm(h1, W["p2"], h2, n_img, D, D, D, D, D, bias=W["p2_b"])
            slots = (ids.view(-1) == c.image_token_index).nonzero().view(-1)
            if slots.numel() != n_img:
                raise ValueError("Image features and image tokens do not match: tokens: %d, features %d" % (slots.numel(), n_img))
            x[slots] = h2                                                              # masked_scatter (bytes only)
        L_pad = (L + 63) // 64 * 64
        n = torch.empty(T, D, device=dev, dtype=BF)
        qk = torch.empty(T, D + KV, device=dev, dtype=BF)
        vt = torch.zeros(B, KV, L_pad, device=dev, dtype=BF)
        att = torch.empty(T, D, device=dev, dtype=BF)
        g = torch.empty(T, M, device=dev, dtype=BF)
        u = torch.empty(T, M, device=dev, dtype=BF)
        states = [x.view(B, L, D).clone()]
        for i in range(c.num_hidden_layers):
            p = "language_model.layers.%d." % i
            _lib.t5_layernorm(x, W[p + "ln0"], n, T, D, c.rms_norm_eps)
            _lib.gemm(n, W[p + "qk"], qk, T, D + KV, D, D, D, D + KV)
            _lib.gemm(W[p + "v"], n, vt, KV, L, D, D, D, L_pad, batch=B, strideB=L * D, strideC=KV * L_pad,
                      flags=_lib.GEMM_PERMUTE_COLS)
            _lib.rope_half_(qk, cos, sin, pos, T, H, D + KV)
            _lib.rope_half_(qk, cos, sin, pos, T, Hk, D + KV, x_off=D)
            _lib.flash_attn_d128(qk, qk, vt, att, B, H, L, L, L * (D + KV), D + KV, L * (D + KV), D + KV, KV * L_pad, L_pad,
                                 L * D, D, 128 ** -0.5, k_off=D, kv_group=H // Hk, causal=True)
            _lib.gemm(att, W[p + "o"], x, T, D, D, D, D, D, R=x, ldr=D)
            _lib.t5_layernorm(x, W[p + "ln1"], n, T, D, c.rms_norm_eps)
            _lib.gemm(n, W[p + "gate"], g, T, M, D, D, D, M, act=_lib.ACT_SILU)
            _lib.gemm(n, W[p + "up"], u, T, M, D, D, D, M)
            _lib.mul_bf16(g, u, g)
            _lib.gemm(g, W[p + "down"], x, T, D, M, M, M, D, R=x, ldr=D)
            states.append(x.view(B, L, D).clone())
        last = torch.empty(T, D, device=dev, dtype=BF)
        _lib.t5_layernorm(x, W["norm"], last, T, D, c.rms_norm_eps)
        states[-1] = last.view(B, L, D)
        out = LlavaOutput(hidden_states=states, last_hidden_state=states[-1])
        return out if return_dict else (out.last_hidden_state, states)

    def to(self, *_, **__):
        return self

    def eval(self):
        return self
