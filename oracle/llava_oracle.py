"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the prompt encoder of the reference's HunyuanVideo pipeline
(`pipeline_hunyuan_video_image2video_lowpass.py:282-420`: `self.text_encoder(**expanded_inputs, pixel_values=...,
output_hidden_states=True).hidden_states[-(num_hidden_layers_to_skip + 1)]`, transformers' `LlavaForConditionalGeneration`:
a CLIP ViT-L/14-336 vision tower, a two-layer projector, a Llama-3-8B decoder).

Third-party code (transformers, pinned 4.48.1 by the reference; 5.15.0 in this image).  **Pinned** against that package:
`tests/golden/llava_vectors.npz` holds `LlavaForConditionalGeneration` outputs (fp32, CPU) generated HERE by
`tests/golden/make_llava_golden.py` on the seeded weights below; `tests/test_llava_cpu.py` checks this restatement against
them.

Restated: token embedding; vision tower hidden state `vision_feature_layer` (-2) without the class token ("default"
selection) -> Linear, GELU(erf), Linear -> scattered over the `image_token_index` positions; Llama decoder layers
(RMSNorm eps 1e-5 with the weight applied after the cast back; q / k / v / o projections without bias; grouped-query
attention, rotate-half RoPE with theta 500000 at `position_ids`, causal mask + key padding mask, scale d^-0.5; SwiGLU MLP);
`hidden_states[i]` = input of layer i, the last one after the final RMSNorm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F

from . import clip_oracle


class LlavaConfig:
    def __init__(self, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=8, vocab_size=128320, rms_norm_eps=1e-5, rope_theta=500000.0, image_token_index=128257,
                 pad_token_id=128258, vision_feature_layer=-2, vision=None):
        self.hidden_size, self.intermediate_size, self.num_hidden_layers = hidden_size, intermediate_size, num_hidden_layers
        self.num_attention_heads, self.num_key_value_heads = num_attention_heads, num_key_value_heads
        self.vocab_size, self.rms_norm_eps, self.rope_theta = vocab_size, rms_norm_eps, rope_theta
        self.image_token_index, self.pad_token_id, self.vision_feature_layer = image_token_index, pad_token_id, vision_feature_layer
        self.vision = vision or clip_oracle.CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                                             num_attention_heads=16, image_size=336, patch_size=14,
                                                             hidden_act="quick_gelu")
        self.head_dim = hidden_size // num_attention_heads


def param_shapes(cfg):
    D, M, dh = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    kv = cfg.num_key_value_heads * dh
    out = {"language_model.embed_tokens.weight": (cfg.vocab_size, D), "language_model.norm.weight": (D,),
           "multi_modal_projector.linear_1.weight": (D, cfg.vision.hidden_size), "multi_modal_projector.linear_1.bias": (D,),
           "multi_modal_projector.linear_2.weight": (D, D), "multi_modal_projector.linear_2.bias": (D,)}
    for i in range(cfg.num_hidden_layers):
        p = "language_model.layers.%d." % i
        out[p + "self_attn.q_proj.weight"], out[p + "self_attn.o_proj.weight"] = (D, D), (D, D)
        out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = (kv, D), (kv, D)
        out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"], out[p + "mlp.down_proj.weight"] = (M, D), (M, D), (D, M)
        out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = (D,), (D,)
    for k, v in clip_oracle.param_shapes(cfg.vision).items():
        out["vision_tower." + k] = v
    return out


def synthetic_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("norm.weight") or ("norm" in name and name.endswith(".weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias") or "class_embedding" in name or "position_embedding" in name:
            t = 0.1 * torch.randn(shape, generator=g)
        elif "embed_tokens" in name:
            t = torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * fan_in ** -0.5
        sd[name] = t.bfloat16().float()
    return sd


def rms_norm(x, w, eps):
    dt = x.dtype
    x32 = x.float()
    x32 = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return w * x32.to(dt)


def rope_tables(cfg, position_ids, dtype):
    """LlamaRotaryEmbedding (default rope): fp32 angles, cos / sin cast to the activations' dtype."""
    dh = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dh, 2, dtype=torch.int64).float() / dh))
    freqs = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def vision_states(cfg, sd, pixel_values):
    vsd = {k[len("vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower.")}
    v = cfg.vision
    if v.hidden_act == "quick_gelu":   # clip_oracle restates the exact-GELU tower; CLIP-L uses x * sigmoid(1.702 x)
        old = F.gelu
        try:
            clip_oracle.F.gelu = lambda t: t * torch.sigmoid(1.702 * t)
            return clip_oracle.encode(v, vsd, pixel_values)
        finally:
            clip_oracle.F.gelu = old
    return clip_oracle.encode(v, vsd, pixel_values)


def forward(cfg, sd, input_ids, attention_mask, position_ids, pixel_values):
    """-> list of hidden states (num_hidden_layers + 1), each [B, L, D], in the dtype of the weights."""
    dt = sd["language_model.norm.weight"].dtype
    B, L = input_ids.shape
    D, H, Hk, dh = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    x = sd["language_model.embed_tokens.weight"][input_ids]
    if pixel_values is not None:
        feats = vision_states(cfg, sd, pixel_values.to(dt))[cfg.vision_feature_layer][:, 1:]
        feats = F.linear(feats, sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"])
        feats = F.linear(F.gelu(feats), sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"])
        mask = (input_ids == cfg.image_token_index)
        x = x.masked_scatter(mask[..., None].expand_as(x), feats.to(dt))
    cos, sin = rope_tables(cfg, position_ids, dt)
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
    allow = causal[None, None] & (attention_mask[:, None, None, :] > 0)
    bias = torch.zeros(B, 1, L, L, dtype=torch.float32).masked_fill(~allow, float("-inf"))
    bias = torch.where(torch.isinf(bias).all(-1, keepdim=True), torch.zeros_like(bias), bias)   # fully masked rows stay finite
    states = [x]
    for i in range(cfg.num_hidden_layers):
        p = "language_model.layers.%d." % i
        n = rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(n, sd[p + "self_attn.q_proj.weight"]).view(B, L, H, dh).transpose(1, 2)
        k = F.linear(n, sd[p + "self_attn.k_proj.weight"]).view(B, L, Hk, dh).transpose(1, 2)
        v = F.linear(n, sd[p + "self_attn.v_proj.weight"]).view(B, L, Hk, dh).transpose(1, 2)
        q = q * cos[:, None] + rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + rotate_half(k) * sin[:, None]
        k, v = k.repeat_interleave(H // Hk, dim=1), v.repeat_interleave(H // Hk, dim=1)
        s = (q.float() @ k.float().transpose(-1, -2)) * dh ** -0.5 + bias
        a = (torch.softmax(s, dim=-1) @ v.float()).to(dt)
        x = x + F.linear(a.transpose(1, 2).reshape(B, L, D), sd[p + "self_attn.o_proj.weight"])
        n = rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        x = x + F.linear(F.silu(F.linear(n, sd[p + "mlp.gate_proj.weight"])) * F.linear(n, sd[p + "mlp.up_proj.weight"]),
                         sd[p + "mlp.down_proj.weight"])
        states.append(x)
    states[-1] = rms_norm(states[-1], sd["language_model.norm.weight"], cfg.rms_norm_eps)
    return states


GOLDEN = dict(text=dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                        vocab_size=300, image_token_index=299, pad_token_id=0),
              vision=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28,
                          patch_size=14, hidden_act="quick_gelu"), seed=31)


def golden_inputs():
    cfg = LlavaConfig(vision=clip_oracle.CLIPVisionConfig(**GOLDEN["vision"]), **GOLDEN["text"])
    sd = synthetic_state_dict(cfg, GOLDEN["seed"])
    g = torch.Generator().manual_seed(GOLDEN["seed"] + 100)
    n_img = (28 // 14) ** 2                                    # 4 image tokens per picture
    ids = torch.randint(1, 298, (2, 24), generator=g)
    ids[:, 3:3 + n_img] = cfg.image_token_index                # image placeholder run, as hy:100-140 lays it out
    ids[0, 19:] = cfg.pad_token_id                             # right padding
    mask = (ids != cfg.pad_token_id).long()
    pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 1)
    px = torch.randn(2, 3, 28, 28, generator=g).bfloat16().float()
    return cfg, sd, ids, mask, pos, px
