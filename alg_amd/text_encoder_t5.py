"""T5 / UMT5 text encoders on MI355X -- the component behind the reference's prompt embeddings:
`pipeline_cogvideox_image2video_lowpass.py:228-268` calls `self.text_encoder(text_input_ids.to(device))[0]` on
transformers' `T5EncoderModel` (T5 v1.1 XXL, no attention mask), `pipeline_wan_image2video_lowpass.py:185-234` calls
`self.text_encoder(ids, mask).last_hidden_state` on `UMT5EncoderModel` (same block, one relative-position table per
block instead of a shared one).  Same call signature, `.dtype`, `.config`, transformers state-dict names; tokenisation
stays outside (sentencepiece vocabularies are checkpoint files).

Launch order per block over the C ABI: `alg_t5_layernorm`, one fused QKV `alg_gemm_bf16`, `alg_attn_bias` (eager
bf16 graph: un-scaled scores + bucketed relative bias + key mask, fp32 softmax), output projection with the residual in
the GEMM epilogue, `alg_t5_layernorm`, `wi_0` with the tanh-GELU epilogue, `wi_1`, `alg_mul_bf16`, `wo` + residual.  A
few hundred tokens once per video: nothing here is performance-critical, it exists so that the prompt path needs no
second framework.  The HIP extension is mandatory: there is no torch fallback.
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


@dataclass
class T5EncoderConfig:
    """Defaults = the T5 v1.1 XXL encoder CogVideoX ships (`text_encoder/config.json`)."""
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    feed_forward_proj: str = "gated-gelu"


@dataclass
class BaseModelOutput:
    last_hidden_state: torch.Tensor

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class T5EncoderModel:
    per_layer_bias = False

    def __init__(self, config: Optional[T5EncoderConfig] = None, device="cuda", dtype=torch.bfloat16):
        self.config = config or T5EncoderConfig()
        c = self.config
        if dtype != torch.bfloat16:
            raise ValueError("the HIP encoder computes in bfloat16")
        if c.d_kv != 64 or c.feed_forward_proj != "gated-gelu" or c.d_model % 64 or c.d_ff % 64:
            raise ValueError("unsupported T5 configuration (d_kv 64, gated-gelu, d_model / d_ff multiples of 64)")
        self.device, self.dtype = torch.device(device), dtype
        self.w = {}
        self._luts = {}

    # ---- weights -------------------------------------------------------------------------------------------------
    def param_shapes(self):
        c = self.config
        inner = c.num_heads * c.d_kv
        out = {"shared.weight": (c.vocab_size, c.d_model)}
        for i in range(c.num_layers):
            p = "encoder.block.%d." % i
            for n in "qkv":
                out[p + "layer.0.SelfAttention.%s.weight" % n] = (inner, c.d_model)
            out[p + "layer.0.SelfAttention.o.weight"] = (c.d_model, inner)
            if i == 0 or self.per_layer_bias:
                out[p + "layer.0.SelfAttention.relative_attention_bias.weight"] = (c.relative_attention_num_buckets,
                                                                                  c.num_heads)
            out[p + "layer.0.layer_norm.weight"] = (c.d_model,)
            out[p + "layer.1.DenseReluDense.wi_0.weight"] = (c.d_ff, c.d_model)
            out[p + "layer.1.DenseReluDense.wi_1.weight"] = (c.d_ff, c.d_model)
            out[p + "layer.1.DenseReluDense.wo.weight"] = (c.d_model, c.d_ff)
            out[p + "layer.1.layer_norm.weight"] = (c.d_model,)
        out["encoder.final_layer_norm.weight"] = (c.d_model,)
        return out

    @classmethod
    def from_synthetic(cls, config=None, seed=0, device="cuda"):
        self = cls(config, device=device)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.param_shapes().items():
            if name.endswith("layer_norm.weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif name == "shared.weight" or "relative_attention_bias" in name:
                t = torch.randn(shape, generator=g)
            else:
                t = torch.randn(shape, generator=g) * (shape[1] ** -0.5) * (0.6 if (".q." in name or ".k." in name) else 1.0)
            sd[name] = t.bfloat16()
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, subfolder="text_encoder", torch_dtype=torch.bfloat16, device="cuda", **_):
        """transformers-format directory on local disk (`text_encoder/config.json` + safetensors shards)."""
        from .weights import component_from_pretrained
        return component_from_pretrained(cls, T5EncoderConfig, path, subfolder, device=device)

    def load_state_dict(self, sd, strict=True):
        shapes = self.param_shapes()
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd = dict(sd, **{"shared.weight": sd["encoder.embed_tokens.weight"]})
        missing = [k for k in shapes if k not in sd]
        if missing and strict:
            raise KeyError("missing text-encoder weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        dev, bf = self.device, torch.bfloat16
        put = lambda t: t.to(dev, bf).contiguous()
        W = {"shared": put(sd["shared.weight"]), "final_ln": put(sd["encoder.final_layer_norm.weight"])}
        for i in range(self.config.num_layers):
            p, a = "encoder.block.%d." % i, "encoder.block.%d.layer.0.SelfAttention." % i
            W[p + "qkv"] = put(torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], 0))
            W[p + "o"] = put(sd[a + "o.weight"])
            if i == 0 or self.per_layer_bias:
                W[p + "bias"] = put(sd[a + "relative_attention_bias.weight"])
            W[p + "ln0"] = put(sd[p + "layer.0.layer_norm.weight"])
            W[p + "ln1"] = put(sd[p + "layer.1.layer_norm.weight"])
            for n in ("wi_0", "wi_1", "wo"):
                W[p + n] = put(sd[p + "layer.1.DenseReluDense.%s.weight" % n])
        self.w = W
        return self

    # ---- host logic ------------------------------------------------------------------------------------------------
    def bucket_lut(self, L):
        """int32 [2L - 1]: bucket of relative position (key - query) = index - (L - 1); transformers'
        `T5Attention._relative_position_bucket` (bidirectional) in the same float32 arithmetic."""
        c = self.config
        rp = torch.arange(-(L - 1), L)
        nb = c.relative_attention_num_buckets // 2
        ret = (rp > 0).to(torch.long) * nb
        n = torch.abs(rp)
        max_exact = nb // 2
        large = max_exact + (torch.log(n.float() / max_exact) / math.log(c.relative_attention_max_distance / max_exact)
                             * (nb - max_exact)).to(torch.long)
        large = torch.min(large, torch.full_like(large, nb - 1))
        return (ret + torch.where(n < max_exact, n, large)).to(torch.int32)

    # ---- forward ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, input_ids, attention_mask=None, return_dict=True, **_):
        if not (torch.is_tensor(input_ids) and input_ids.is_cuda and input_ids.dim() == 2):
            raise _lib.AlgHipError("T5EncoderModel: input_ids must be a [B, L] device tensor (HIP-only path)")
        c, W = self.config, self.w
        B, L = input_ids.shape
        if L > 512:
            raise ValueError("sequence length %d > 512 is not built (the pipelines use 226 and 512)" % L)
        dev, bf = self.device, torch.bfloat16
        D, F, H = c.d_model, c.d_ff, c.num_heads
        inner, T = H * c.d_kv, B * L
        ids = input_ids.to(torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int32).contiguous() if attention_mask is not None else None
        if L not in self._luts:
            self._luts[L] = self.bucket_lut(L).to(dev)
        lut = self._luts[L]
        x = torch.empty(T, D, device=dev, dtype=bf)
        _lib.embed_rows(ids, W["shared"], x)
        n = torch.empty(T, D, device=dev, dtype=bf)
        qkv = torch.empty(T, 3 * inner, device=dev, dtype=bf)
        att = torch.empty(T, inner, device=dev, dtype=bf)
        g = torch.empty(T, F, device=dev, dtype=bf)
        u = torch.empty(T, F, device=dev, dtype=bf)
        for i in range(c.num_layers):
            p = "encoder.block.%d." % i
            bias = W[p + "bias"] if self.per_layer_bias else W["encoder.block.0.bias"]
            _lib.t5_layernorm(x, W[p + "ln0"], n, T, D, c.layer_norm_epsilon)
            _lib.gemm(n, W[p + "qkv"], qkv, T, 3 * inner, D, D, D, 3 * inner)
            _lib.attn_bias(qkv, att, bias, lut, mask, B, H, L, scale=1.0)
            _lib.gemm(att, W[p + "o"], x, T, D, inner, inner, inner, D, R=x, ldr=D)
            _lib.t5_layernorm(x, W[p + "ln1"], n, T, D, c.layer_norm_epsilon)
            _lib.gemm(n, W[p + "wi_0"], g, T, F, D, D, D, F, act=_lib.ACT_GELU_TANH)
            _lib.gemm(n, W[p + "wi_1"], u, T, F, D, D, D, F)
            _lib.mul_bf16(g, u, g)
            _lib.gemm(g, W[p + "wo"], x, T, D, F, F, F, D, R=x, ldr=D)
        out = torch.empty(T, D, device=dev, dtype=bf)
        _lib.t5_layernorm(x, W["final_ln"], out, T, D, c.layer_norm_epsilon)
        out = out.view(B, L, D)
        return BaseModelOutput(last_hidden_state=out) if return_dict else (out,)

    def to(self, *_, **__):
        return self

    def eval(self):
        return self


class UMT5EncoderModel(T5EncoderModel):
    """transformers' UMT5EncoderModel (wan:185-234; UMT5-XXL: vocab 256384): every block owns its relative-position table."""
    per_layer_bias = True
