"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CLIP vision tower the Wan pipeline calls for its image embeddings
(wan:228-234 `encode_image`: `self.image_encoder(**image, output_hidden_states=True).hidden_states[-2]`, transformers'
`CLIPVisionModel`; Wan 2.1 ships the ViT-H/14 tower: 1280 wide, 32 layers, 16 heads of 80, MLP 5120, exact GELU,
224 x 224 images -> 257 tokens).

Third-party code (transformers, pinned 4.48.1 by the reference; 5.15.0 in this image).  **Pinned** against that package:
`tests/golden/clip_vectors.npz` holds `CLIPVisionModel` outputs (fp32, CPU) generated HERE by
`tests/golden/make_clip_golden.py` on the seeded weights of `synthetic_state_dict`; `tests/test_clip_cpu.py` checks this
restatement against them.

Restated: patch embedding (Conv2d k = s = patch, no bias) + class token + learned positions, pre-LayerNorm, pre-norm
transformer blocks (LayerNorm eps 1e-5 with bias; q/k/v/out projections with bias; softmax((q k^T) * d^-0.5) v; MLP
fc1 - GELU(erf) - fc2), `hidden_states[i]` = input of block i (so [-2] = output of the last-but-one block).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F


class CLIPVisionConfig:
    def __init__(self, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16,
                 image_size=224, patch_size=14, layer_norm_eps=1e-5, hidden_act="gelu"):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.image_size, self.patch_size = image_size, patch_size
        self.layer_norm_eps, self.hidden_act = layer_norm_eps, hidden_act


def param_shapes(cfg):
    D, M = cfg.hidden_size, cfg.intermediate_size
    n = (cfg.image_size // cfg.patch_size) ** 2 + 1
    out = {"embeddings.class_embedding": (D,), "embeddings.patch_embedding.weight": (D, 3, cfg.patch_size, cfg.patch_size),
           "embeddings.position_embedding.weight": (n, D), "pre_layrnorm.weight": (D,), "pre_layrnorm.bias": (D,)}
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + "self_attn.%s.weight" % nm], out[p + "self_attn.%s.bias" % nm] = (D, D), (D,)
        for nm in ("layer_norm1", "layer_norm2"):
            out[p + nm + ".weight"], out[p + nm + ".bias"] = (D,), (D,)
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (M, D), (M,)
        out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (D, M), (D,)
    out["post_layernorm.weight"], out["post_layernorm.bias"] = (D,), (D,)
    return out


def synthetic_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias") or "class_embedding" in name or "position_embedding" in name:
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * fan_in ** -0.5
        sd[name] = t.bfloat16().float()
    return sd


def encode(cfg, sd, pixel_values):
    """pixel_values [B, 3, S, S] -> list of hidden states (len num_hidden_layers + 1), each [B, tokens, D], fp32."""
    B = pixel_values.shape[0]
    D, H = cfg.hidden_size, cfg.num_attention_heads
    dh = D // H
    x = F.conv2d(pixel_values, sd["embeddings.patch_embedding.weight"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    x = torch.cat([sd["embeddings.class_embedding"].expand(B, 1, D), x], dim=1) + sd["embeddings.position_embedding.weight"]
    x = F.layer_norm(x, (D,), sd["pre_layrnorm.weight"], sd["pre_layrnorm.bias"], cfg.layer_norm_eps)
    states = [x]
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layers.%d." % i
        n = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.layer_norm_eps)
        proj = lambda nm: F.linear(n, sd[p + "self_attn.%s.weight" % nm], sd[p + "self_attn.%s.bias" % nm]).view(
            B, -1, H, dh).transpose(1, 2)
        a = torch.softmax(proj("q_proj") @ proj("k_proj").transpose(-1, -2) * dh ** -0.5, dim=-1) @ proj("v_proj")
        x = x + F.linear(a.transpose(1, 2).reshape(B, -1, D), sd[p + "self_attn.out_proj.weight"],
                         sd[p + "self_attn.out_proj.bias"])
        n = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.layer_norm_eps)
        x = x + F.linear(F.gelu(F.linear(n, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                         sd[p + "mlp.fc2.bias"])
        states.append(x)
    return states


GOLDEN = dict(cfg=dict(hidden_size=320, intermediate_size=640, num_hidden_layers=3, num_attention_heads=4, image_size=56,
                       patch_size=14), seed=21, batch=2)


def golden_inputs():
    cfg = CLIPVisionConfig(**GOLDEN["cfg"])
    g = torch.Generator().manual_seed(GOLDEN["seed"] + 100)
    px = torch.randn(GOLDEN["batch"], 3, cfg.image_size, cfg.image_size, generator=g).bfloat16().float()
    return cfg, synthetic_state_dict(cfg, GOLDEN["seed"]), px
