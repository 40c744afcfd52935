"""CLIP text tower on MI355X -- HunyuanVideo's pooled prompt embedding:
`pipeline_hunyuan_video_image2video_lowpass.py:421-452` calls
`self.text_encoder_2(text_input_ids, output_hidden_states=False).pooler_output` on transformers' `CLIPTextModel` (CLIP-L/14
text: 768 wide, 12 layers, 12 heads of 64, quick_gelu, 77 positions).  Same call signature, transformers state-dict names
(with or without the 4.x `text_model.` prefix).

Launch order over the C ABI: `alg_embed_rows` + `alg_lincomb` (token + position), `alg_layernorm_mod_f32`, fused QKV
`alg_gemm_bf16` with bias, `alg_attn_bias` (eager graph, causal), output projection / fc2 with bias + residual in the GEMM
epilogue, `alg_quick_gelu`.  77 tokens once per video; the HIP extension is mandatory, there is no torch fallback.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


@dataclass
class CLIPTextEncoderConfig:
    """Defaults = the CLIP-L/14 text encoder HunyuanVideo ships as `text_encoder_2`."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    eos_token_id: int = 2


@dataclass
class CLIPTextOutput:
    last_hidden_state: torch.Tensor
    pooler_output: torch.Tensor


class CLIPTextModel:
    def __init__(self, config: Optional[CLIPTextEncoderConfig] = None, device="cuda", dtype=torch.bfloat16):
        self.config = config or CLIPTextEncoderConfig()
        c = self.config
        if dtype != torch.bfloat16:
            raise ValueError("the HIP encoder computes in bfloat16")
        if c.hidden_size // c.num_attention_heads != 64 or c.hidden_size % 64 or c.intermediate_size % 64 or \
                c.hidden_act != "quick_gelu":
            raise ValueError("unsupported CLIP text configuration (head_dim 64, quick_gelu, widths multiples of 64)")
        self.device, self.dtype = torch.device(device), dtype
        self.w = {}

    def param_shapes(self):
        c = self.config
        D, M = c.hidden_size, c.intermediate_size
        out = {"embeddings.token_embedding.weight": (c.vocab_size, D),
               "embeddings.position_embedding.weight": (c.max_position_embeddings, D)}
        for i in range(c.num_hidden_layers):
            p = "encoder.layers.%d." % i
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                out[p + "self_attn.%s.weight" % nm], out[p + "self_attn.%s.bias" % nm] = (D, D), (D,)
            for nm in ("layer_norm1", "layer_norm2"):
                out[p + nm + ".weight"], out[p + nm + ".bias"] = (D,), (D,)
            out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (M, D), (M,)
            out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (D, M), (D,)
        out["final_layer_norm.weight"], out["final_layer_norm.bias"] = (D,), (D,)
        return out

    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda"):
        self = cls(config, device=device)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.param_shapes().items():
            if "norm" in name and name.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif name.endswith(".bias") or "position_embedding" in name:
                t = 0.1 * torch.randn(shape, generator=g)
            elif "token_embedding" in name:
                t = torch.randn(shape, generator=g)
            else:
                t = torch.randn(shape, generator=g) * shape[1] ** -0.5
            sd[name] = t.bfloat16()
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, subfolder="text_encoder_2", torch_dtype=torch.bfloat16, device="cuda", **_):
        """transformers-format directory on local disk; a joint CLIP config's `text_config` is unwrapped."""
        from .weights import component_from_pretrained
        return component_from_pretrained(cls, CLIPTextEncoderConfig, path, subfolder, device=device, nested="text_config")

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if missing and strict:
            raise KeyError("missing text-encoder weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        c, dev, bf = self.config, self.device, torch.bfloat16
        put = lambda t: t.to(dev, bf).contiguous()
        f32 = lambda t: t.to(bf).to(dev, torch.float32).contiguous()
        W = {"tok": put(sd["embeddings.token_embedding.weight"]), "pos": put(sd["embeddings.position_embedding.weight"]),
             "final_ln": (f32(sd["final_layer_norm.weight"]), f32(sd["final_layer_norm.bias"]))}
        for i in range(c.num_hidden_layers):
            p, a = "encoder.layers.%d." % i, "encoder.layers.%d.self_attn." % i
            W[p + "qkv"] = put(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0))
            W[p + "qkv_b"] = put(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0))
            W[p + "o"], W[p + "o_b"] = put(sd[a + "out_proj.weight"]), put(sd[a + "out_proj.bias"])
            for n in ("layer_norm1", "layer_norm2"):
                W[p + n] = (f32(sd[p + n + ".weight"]), f32(sd[p + n + ".bias"]))
            for n in ("fc1", "fc2"):
                W[p + n], W[p + n + "_b"] = put(sd[p + "mlp.%s.weight" % n]), put(sd[p + "mlp.%s.bias" % n])
        self.w = W
        return self

    def eos_positions(self, input_ids):
        """transformers' pooling rule: argmax of the ids for the legacy eos_token_id == 2, else the first eos token."""
        if self.config.eos_token_id == 2:
            return input_ids.argmax(dim=-1)
        return (input_ids == self.config.eos_token_id).int().argmax(dim=-1)

    @torch.no_grad()
    def __call__(self, input_ids=None, output_hidden_states=False, return_dict=True, **_):
        c, W = self.config, self.w
        if not (torch.is_tensor(input_ids) and input_ids.is_cuda and input_ids.dim() == 2):
            raise _lib.AlgHipError("CLIPTextModel: input_ids must be a [B, L] device tensor (HIP-only path)")
        B, L = input_ids.shape
        if L > c.max_position_embeddings:
            raise ValueError("sequence length %d exceeds max_position_embeddings %d" % (L, c.max_position_embeddings))
        dev, bf = self.device, torch.bfloat16
        D, M, H = c.hidden_size, c.intermediate_size, c.num_attention_heads
        T = B * L
        ids = input_ids.to(torch.int64).contiguous()
        x = torch.empty(B, L, D, device=dev, dtype=bf)
        _lib.embed_rows(ids, W["tok"], x)
        pos = W["pos"][:L].contiguous()
        for b in range(B):
            _lib.lincomb([(1.0, x[b]), (1.0, pos)], bf, out=x[b])
        x = x.view(T, D)
        n = torch.empty(T, D, device=dev, dtype=bf)
        qkv = torch.empty(T, 3 * D, device=dev, dtype=bf)
        att = torch.empty(T, D, device=dev, dtype=bf)
        mid = torch.empty(T, M, device=dev, dtype=bf)
        for i in range(c.num_hidden_layers):
            p = "encoder.layers.%d." % i
            _lib.layernorm_mod_f32(x, n, W[p + "layer_norm1"][0], W[p + "layer_norm1"][1], None, None, 0, 1, T, D,
                                   c.layer_norm_eps)
            _lib.gemm(n, W[p + "qkv"], qkv, T, 3 * D, D, D, D, 3 * D, bias=W[p + "qkv_b"])
            _lib.attn_bias(qkv, att, None, None, None, B, H, L, scale=0.125, head_dim=64, causal=True)
            _lib.gemm(att, W[p + "o"], x, T, D, D, D, D, D, bias=W[p + "o_b"], R=x, ldr=D)
            _lib.layernorm_mod_f32(x, n, W[p + "layer_norm2"][0], W[p + "layer_norm2"][1], None, None, 0, 1, T, D,
                                   c.layer_norm_eps)
            _lib.gemm(n, W[p + "fc1"], mid, T, M, D, D, D, M, bias=W[p + "fc1_b"])
            _lib.quick_gelu_(mid)
            _lib.gemm(mid, W[p + "fc2"], x, T, D, M, M, M, D, bias=W[p + "fc2_b"], R=x, ldr=D)
        out = torch.empty(T, D, device=dev, dtype=bf)
        _lib.layernorm_mod_f32(x, out, W["final_ln"][0], W["final_ln"][1], None, None, 0, 1, T, D, c.layer_norm_eps)
        out = out.view(B, L, D)
        pooled = out[torch.arange(B, device=dev), self.eos_positions(ids)]
        res = CLIPTextOutput(last_hidden_state=out, pooler_output=pooled)
        return res if return_dict else (out, pooled)

    def to(self, *_, **__):
        return self

    def eval(self):
        return self
