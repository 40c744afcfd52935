"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CLIP text tower HunyuanVideo uses for its pooled prompt embedding
(hy:421-452 `_get_clip_prompt_embeds`: `self.text_encoder_2(text_input_ids, output_hidden_states=False).pooler_output`,
transformers' `CLIPTextModel`; CLIP-L/14 text: 768 wide, 12 layers, 12 heads of 64, MLP 3072, quick_gelu, 77 positions).

Third-party code (transformers, pinned 4.48.1 by the reference; 5.15.0 in this image).  **Pinned** against that package:
`tests/golden/clip_text_vectors.npz` holds `CLIPTextModel` outputs (fp32, CPU) generated HERE by
`tests/golden/make_clip_text_golden.py` on the seeded weights of `synthetic_state_dict`; `tests/test_clip_cpu.py` checks
this restatement against them.

Restated: token + position embeddings; pre-norm blocks with CAUSAL self-attention (softmax((q k^T) d^-0.5 + causal mask)),
MLP fc1 - quick_gelu (x * sigmoid(1.702 x)) - fc2; final LayerNorm; pooled output = the final-norm state at the
end-of-sequence token (argmax of the ids when eos_token_id == 2, the legacy rule, else the first eos_token_id).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F


class CLIPTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=2):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings, self.layer_norm_eps, self.eos_token_id = max_position_embeddings, layer_norm_eps, eos_token_id


def param_shapes(cfg):
    D, M = cfg.hidden_size, cfg.intermediate_size
    out = {"embeddings.token_embedding.weight": (cfg.vocab_size, D),
           "embeddings.position_embedding.weight": (cfg.max_position_embeddings, D)}
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + "self_attn.%s.weight" % nm], out[p + "self_attn.%s.bias" % nm] = (D, D), (D,)
        for nm in ("layer_norm1", "layer_norm2"):
            out[p + nm + ".weight"], out[p + nm + ".bias"] = (D,), (D,)
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (M, D), (M,)
        out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (D, M), (D,)
    out["final_layer_norm.weight"], out["final_layer_norm.bias"] = (D,), (D,)
    return out


def synthetic_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias") or "position_embedding" in name:
            t = 0.1 * torch.randn(shape, generator=g)
        elif "token_embedding" in name:
            t = torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) * shape[1] ** -0.5
        sd[name] = t.bfloat16().float()
    return sd


def eos_positions(cfg, input_ids):
    if cfg.eos_token_id == 2:
        return input_ids.argmax(dim=-1)
    return (input_ids == cfg.eos_token_id).int().argmax(dim=-1)


def encode(cfg, sd, input_ids):
    """input_ids [B, L] -> (last_hidden_state [B, L, D] after the final LayerNorm, pooler_output [B, D]), fp32."""
    B, L = input_ids.shape
    D, H = cfg.hidden_size, cfg.num_attention_heads
    dh = D // H
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:L]
    causal = torch.full((L, L), float("-inf")).triu(1)
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layers.%d." % i
        n = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.layer_norm_eps)
        proj = lambda nm: F.linear(n, sd[p + "self_attn.%s.weight" % nm], sd[p + "self_attn.%s.bias" % nm]).view(
            B, L, H, dh).transpose(1, 2)
        a = torch.softmax(proj("q_proj") @ proj("k_proj").transpose(-1, -2) * dh ** -0.5 + causal, dim=-1) @ proj("v_proj")
        x = x + F.linear(a.transpose(1, 2).reshape(B, L, D), sd[p + "self_attn.out_proj.weight"],
                         sd[p + "self_attn.out_proj.bias"])
        n = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.layer_norm_eps)
        h = F.linear(n, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        x = x + F.linear(h * torch.sigmoid(1.702 * h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.layer_norm(x, (D,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], cfg.layer_norm_eps)
    return x, x[torch.arange(B), eos_positions(cfg, input_ids)]


GOLDEN = dict(cfg=dict(vocab_size=99, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                       max_position_embeddings=16, eos_token_id=98), seed=31)


def golden_inputs():
    cfg = CLIPTextConfig(**GOLDEN["cfg"])
    g = torch.Generator().manual_seed(GOLDEN["seed"] + 100)
    ids = torch.randint(1, 90, (2, 16), generator=g)
    ids[:, 0] = 97
    ids[0, 9], ids[1, 5] = 98, 98
    ids[0, 10:], ids[1, 6:] = 0, 0
    return cfg, synthetic_state_dict(cfg, GOLDEN["seed"]), ids
