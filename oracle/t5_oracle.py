"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the text encoders the reference calls for its prompt embeddings:
transformers' `T5EncoderModel` (cog:228-268 `_get_t5_prompt_embeds`: `self.text_encoder(text_input_ids.to(device))[0]`,
no attention mask) and `UMT5EncoderModel` (wan:185-234: `self.text_encoder(ids, mask).last_hidden_state`).

The model code is a third-party dependency (transformers, pinned 4.48.1 by the reference's requirements.txt; 5.15.0 is
what this image holds).  **Pinned** against that package itself: `tests/golden/t5_vectors.npz` holds outputs of
`transformers.T5EncoderModel` / `UMT5EncoderModel` (fp32, CPU) generated HERE by `tests/golden/make_t5_golden.py` on the
seeded weights of `synthetic_state_dict` below; `tests/test_t5_cpu.py` checks this restatement against them.

Restated: token embedding; per block  x += o(attn(T5LayerNorm(x)))  with un-scaled dot-product attention plus the
bucketed relative-position bias (bidirectional, 32 buckets, max distance 128; T5 shares block 0's table across blocks,
UMT5 has one per block) and an additive key mask;  x += wo(gelu_new(wi_0 n) * wi_1 n)  on  n = T5LayerNorm(x);  final
T5LayerNorm.  T5LayerNorm = x * rsqrt(mean(x^2) + eps) * weight (no mean subtraction, no bias).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch


class T5Config:
    def __init__(self, vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                 per_layer_bias=False):
        self.vocab_size, self.d_model, self.d_kv, self.d_ff = vocab_size, d_model, d_kv, d_ff
        self.num_layers, self.num_heads = num_layers, num_heads
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.relative_attention_max_distance = relative_attention_max_distance
        self.layer_norm_epsilon = layer_norm_epsilon
        self.per_layer_bias = per_layer_bias          # UMT5


def param_shapes(cfg):
    inner = cfg.num_heads * cfg.d_kv
    out = {"shared.weight": (cfg.vocab_size, cfg.d_model)}
    for i in range(cfg.num_layers):
        p = "encoder.block.%d." % i
        for n in "qkv":
            out[p + "layer.0.SelfAttention.%s.weight" % n] = (inner, cfg.d_model)
        out[p + "layer.0.SelfAttention.o.weight"] = (cfg.d_model, inner)
        if i == 0 or cfg.per_layer_bias:
            out[p + "layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg.relative_attention_num_buckets,
                                                                              cfg.num_heads)
        out[p + "layer.0.layer_norm.weight"] = (cfg.d_model,)
        out[p + "layer.1.DenseReluDense.wi_0.weight"] = (cfg.d_ff, cfg.d_model)
        out[p + "layer.1.DenseReluDense.wi_1.weight"] = (cfg.d_ff, cfg.d_model)
        out[p + "layer.1.DenseReluDense.wo.weight"] = (cfg.d_model, cfg.d_ff)
        out[p + "layer.1.layer_norm.weight"] = (cfg.d_model,)
    out["encoder.final_layer_norm.weight"] = (cfg.d_model,)
    return out


def synthetic_state_dict(cfg, seed=0):
    """Seeded weights, bf16-representable (stored fp32): O(1) activations and O(1) attention logits."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name == "shared.weight":
            t = torch.randn(shape, generator=g)
        elif "relative_attention_bias" in name:
            t = torch.randn(shape, generator=g)
        elif ".q." in name or ".k." in name:
            t = torch.randn(shape, generator=g) * (shape[1] ** -0.5) * 0.6     # logits ~ N(0, 0.36 * d_kv / ...)
        else:
            t = torch.randn(shape, generator=g) * (shape[1] ** -0.5)
        sd[name] = t.bfloat16().float()
    return sd


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """transformers T5Attention._relative_position_bucket, bidirectional."""
    num_buckets //= 2
    ret = (relative_position > 0).to(torch.long) * num_buckets
    n = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(is_small, n, large)


def _norm(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def encode(cfg, sd, input_ids, attention_mask=None):
    """input_ids [B, L] (long), attention_mask [B, L] (1 = keep) or None -> last_hidden_state [B, L, d_model] fp32."""
    B, L = input_ids.shape
    H, dk = cfg.num_heads, cfg.d_kv
    x = sd["shared.weight"][input_ids]
    ctx = torch.arange(L)
    bucket = relative_position_bucket(ctx[None, :] - ctx[:, None], cfg.relative_attention_num_buckets,
                                      cfg.relative_attention_max_distance)                      # [query, key]
    madd = None
    if attention_mask is not None:
        madd = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg.num_layers):
        p = "encoder.block.%d." % i
        bp = p if cfg.per_layer_bias else "encoder.block.0."
        bias = sd[bp + "layer.0.SelfAttention.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]   # [1,H,L,L]
        n = _norm(x, sd[p + "layer.0.layer_norm.weight"], cfg.layer_norm_epsilon)
        proj = lambda nm: (n @ sd[p + "layer.0.SelfAttention.%s.weight" % nm].T).view(B, L, H, dk).transpose(1, 2)
        s = proj("q") @ proj("k").transpose(-1, -2) + bias
        if madd is not None:
            s = s + madd
        a = torch.softmax(s, dim=-1) @ proj("v")
        x = x + a.transpose(1, 2).reshape(B, L, H * dk) @ sd[p + "layer.0.SelfAttention.o.weight"].T
        n = _norm(x, sd[p + "layer.1.layer_norm.weight"], cfg.layer_norm_epsilon)
        ff = gelu_new(n @ sd[p + "layer.1.DenseReluDense.wi_0.weight"].T) * (n @ sd[p + "layer.1.DenseReluDense.wi_1.weight"].T)
        x = x + ff @ sd[p + "layer.1.DenseReluDense.wo.weight"].T
    return _norm(x, sd["encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)


GOLDEN_CASES = {
    # name: (config kwargs, seed, batch, length, masked tail of sample 1)
    "t5": (dict(vocab_size=100, d_model=512, d_kv=64, d_ff=1024, num_layers=2, num_heads=8), 11, 2, 24, 0),
    "t5_long": (dict(vocab_size=64, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2), 12, 1, 226, 0),
    "umt5": (dict(vocab_size=100, d_model=512, d_kv=64, d_ff=1024, num_layers=2, num_heads=8, per_layer_bias=True), 13, 2,
             40, 15),
}


def golden_inputs(name):
    kw, seed, B, L, tail = GOLDEN_CASES[name]
    g = torch.Generator().manual_seed(seed + 100)
    ids = torch.randint(0, kw["vocab_size"], (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    if tail:
        mask[B - 1, L - tail:] = 0
    return T5Config(**kw), synthetic_state_dict(T5Config(**kw), seed), ids, (mask if tail else None)
