"""CLIP vision tower on MI355X -- the component behind the Wan pipeline's image embeddings:
`pipeline_wan_image2video_lowpass.py:228-234` runs `self.image_processor(images=image, return_tensors="pt")` and
`self.image_encoder(**image, output_hidden_states=True).hidden_states[-2]` on transformers' `CLIPVisionModel` (Wan 2.1:
ViT-H/14, 1280 wide, 32 layers, 16 heads of 80, exact GELU; 224 x 224 -> 257 tokens).  Same call signature, transformers
state-dict names (with or without the 4.x `vision_model.` prefix).

Launch order over the C ABI: `alg_patchify3d` + `alg_gemm_bf16` (the stride-14 patch convolution), `alg_lincomb` (+
positions), `alg_layernorm_mod_f32` (LayerNorm with bias, fp32 statistics), fused QKV `alg_gemm_bf16` with bias,
`alg_attn_bias` (eager graph, head_dim 80, scores * d^-0.5), output projection / fc2 with bias + residual in the GEMM
epilogue, `alg_gelu_erf`.  257 tokens once per video; the HIP extension is mandatory, there is no torch fallback.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib


@dataclass
class CLIPVisionEncoderConfig:
    """Defaults = the ViT-H/14 image encoder Wan2.1-I2V ships (`image_encoder/config.json`)."""
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    hidden_act: str = "gelu"


@dataclass
class CLIPVisionOutput:
    last_hidden_state: torch.Tensor
    hidden_states: Optional[List[torch.Tensor]] = None


class BatchFeature(dict):
    """What an image processor returns: a dict that can be moved (`.to(device)`) and splatted (`**image`)."""

    def to(self, device):
        return BatchFeature({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class CLIPImageProcessor:
    """transformers' CLIPImageProcessor defaults: RGB, resize the shortest edge to `size` (bicubic), centre crop,
    rescale by 1/255, normalise with the OpenAI CLIP mean / std.  PIL + numpy on the host (one image per video)."""

    def __init__(self, size=224, image_mean=(0.48145466, 0.4578275, 0.40821073), image_std=(0.26862954, 0.26130258, 0.27577711)):
        self.size, self.image_mean, self.image_std = size, image_mean, image_std

    @classmethod
    def from_pretrained(cls, path, subfolder="image_processor", **_):
        """`preprocessor_config.json` of the checkpoint (size as an int or {"shortest_edge": n}, mean, std)."""
        import json
        import os
        root = os.path.join(path, subfolder) if os.path.isdir(os.path.join(path, subfolder)) else path
        with open(os.path.join(root, "preprocessor_config.json")) as f:
            raw = json.load(f)
        size = raw.get("size", 224)
        if isinstance(size, dict):
            size = size.get("shortest_edge", size.get("height", 224))
        return cls(size=int(size), image_mean=tuple(raw.get("image_mean", cls().image_mean)),
                   image_std=tuple(raw.get("image_std", cls().image_std)))

    def __call__(self, images, return_tensors="pt"):
        import numpy as np
        from PIL import Image
        imgs = images if isinstance(images, (list, tuple)) else [images]
        out = []
        for im in imgs:
            if torch.is_tensor(im):                     # [3, H, W] in [0, 1]
                im = Image.fromarray((im.permute(1, 2, 0).clamp(0, 1).float().cpu().numpy() * 255).round().astype("uint8"))
            im = im.convert("RGB")
            w, h = im.size
            s = self.size / min(w, h)
            nw, nh = (self.size, max(self.size, int(h * s))) if w <= h else (max(self.size, int(w * s)), self.size)
            im = im.resize((nw, nh), resample=Image.BICUBIC)
            left, top = (nw - self.size) // 2, (nh - self.size) // 2
            im = im.crop((left, top, left + self.size, top + self.size))
            a = np.asarray(im, dtype=np.float32) * (1.0 / 255.0)
            a = (a - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
            out.append(torch.from_numpy(a).permute(2, 0, 1))
        return BatchFeature(pixel_values=torch.stack(out))


class CLIPVisionModel:
    def __init__(self, config: Optional[CLIPVisionEncoderConfig] = None, device="cuda", dtype=torch.bfloat16):
        self.config = config or CLIPVisionEncoderConfig()
        c = self.config
        if dtype != torch.bfloat16:
            raise ValueError("the HIP encoder computes in bfloat16")
        dh = c.hidden_size // c.num_attention_heads
        if dh not in (64, 80) or c.hidden_size % 64 or c.intermediate_size % 64 or c.hidden_act not in ("gelu", "quick_gelu") \
                or c.image_size % c.patch_size:
            raise ValueError("unsupported CLIP vision configuration (head_dim 64 / 80, GELU or quick-GELU, widths multiples of 64)")
        self.head_dim = dh
        self.tokens = (c.image_size // c.patch_size) ** 2 + 1
        # up to 448 tokens the eager-graph attention kernel (K resident in LDS) is used; longer sequences (ViT-L/14 at 336 px:
        # 577 tokens, the tower inside HunyuanVideo's Llava prompt encoder) go through the flash attention of the DiT
        self.flash = self.tokens > 448
        if self.flash and dh != 64:
            raise ValueError("CLIP towers with more than 448 tokens need head_dim 64 (flash attention path)")
        self.kpad = -(-3 * c.patch_size * c.patch_size // 64) * 64
        self.device, self.dtype = torch.device(device), dtype
        self.w = {}

    def param_shapes(self):
        c = self.config
        D, M = c.hidden_size, c.intermediate_size
        out = {"embeddings.class_embedding": (D,), "embeddings.patch_embedding.weight": (D, 3, c.patch_size, c.patch_size),
               "embeddings.position_embedding.weight": (self.tokens, D), "pre_layrnorm.weight": (D,),
               "pre_layrnorm.bias": (D,)}
        for i in range(c.num_hidden_layers):
            p = "encoder.layers.%d." % i
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                out[p + "self_attn.%s.weight" % nm], out[p + "self_attn.%s.bias" % nm] = (D, D), (D,)
            for nm in ("layer_norm1", "layer_norm2"):
                out[p + nm + ".weight"], out[p + nm + ".bias"] = (D,), (D,)
            out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (M, D), (M,)
            out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (D, M), (D,)
        out["post_layernorm.weight"], out["post_layernorm.bias"] = (D,), (D,)
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
            else:
                t = torch.randn(shape, generator=g) * torch.Size(shape[1:]).numel() ** -0.5
            sd[name] = t.bfloat16()
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, subfolder="image_encoder", torch_dtype=torch.bfloat16, device="cuda", **_):
        """transformers-format directory on local disk; a joint CLIP config's `vision_config` is unwrapped."""
        from .weights import component_from_pretrained
        return component_from_pretrained(cls, CLIPVisionEncoderConfig, path, subfolder, device=device, nested="vision_config")

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in sd.items()}
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd and not k.startswith("post_layernorm")]
        if missing and strict:
            raise KeyError("missing image-encoder weights: %s ..." % missing[:3])
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s: shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        c, dev, bf = self.config, self.device, torch.bfloat16
        D = c.hidden_size
        put = lambda t: t.to(dev, bf).contiguous()
        f32 = lambda t: t.to(bf).to(dev, torch.float32).contiguous()       # LayerNorm affine, bf16 values kept in fp32
        pw = sd["embeddings.patch_embedding.weight"].reshape(D, -1).to(torch.float32)
        pw = torch.nn.functional.pad(pw, (0, self.kpad - pw.shape[1]))
        pos = sd["embeddings.position_embedding.weight"].to(bf)
        W = {"patch": put(pw), "pos": put(pos[1:]),
             "cls": put(sd["embeddings.class_embedding"].to(bf) + pos[0]),  # cat([class, patches]) + positions, row 0
             "pre_ln": (f32(sd["pre_layrnorm.weight"]), f32(sd["pre_layrnorm.bias"]))}
        for i in range(c.num_hidden_layers):
            p, a = "encoder.layers.%d." % i, "encoder.layers.%d.self_attn." % i
            if self.flash:
                W[p + "qk"] = put(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"]], 0))
                W[p + "qk_b"] = put(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"]], 0))
                W[p + "v"], W[p + "v_b"] = put(sd[a + "v_proj.weight"]), put(sd[a + "v_proj.bias"])
            else:
                W[p + "qkv"] = put(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0))
                W[p + "qkv_b"] = put(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0))
            W[p + "o"], W[p + "o_b"] = put(sd[a + "out_proj.weight"]), put(sd[a + "out_proj.bias"])
            for n in ("layer_norm1", "layer_norm2"):
                W[p + n] = (f32(sd[p + n + ".weight"]), f32(sd[p + n + ".bias"]))
            for n in ("fc1", "fc2"):
                W[p + n], W[p + n + "_b"] = put(sd[p + "mlp.%s.weight" % n]), put(sd[p + "mlp.%s.bias" % n])
        self.w = W
        return self

    @torch.no_grad()
    def __call__(self, pixel_values=None, output_hidden_states=False, return_dict=True, **_):
        c, W = self.config, self.w
        if not (torch.is_tensor(pixel_values) and pixel_values.is_cuda and pixel_values.dim() == 4):
            raise _lib.AlgHipError("CLIPVisionModel: pixel_values must be a [B, 3, S, S] device tensor (HIP-only path)")
        B, C, S, S2 = pixel_values.shape
        if C != 3 or S != c.image_size or S2 != c.image_size:
            raise ValueError("pixel_values must be [B, 3, %d, %d]" % (c.image_size, c.image_size))
        dev, bf = self.device, torch.bfloat16
        D, M, H, L, dh = c.hidden_size, c.intermediate_size, c.num_attention_heads, self.tokens, self.head_dim
        T, P = B * L, L - 1
        px = pixel_values.to(bf).contiguous()
        patches = torch.empty(B * P, self.kpad, device=dev, dtype=bf)
        _lib.patchify3d(px, patches, B, 3, 1, S, S, c.patch_size, c.patch_size, self.kpad)
        tok = torch.empty(B, L, D, device=dev, dtype=bf)
        _lib.gemm(patches, W["patch"], tok, P, D, self.kpad, self.kpad, self.kpad, D, batch=B, strideA=P * self.kpad,
                  strideC=L * D, c_off=D)
        tok[:, 0] = W["cls"]
        for b in range(B):
            _lib.lincomb([(1.0, tok[b, 1:]), (1.0, W["pos"])], bf, out=tok[b, 1:])
        n_states = c.num_hidden_layers + 1
        hs = torch.empty(n_states, T, D, device=dev, dtype=bf)
        _lib.layernorm_mod_f32(tok, hs[0], W["pre_ln"][0], W["pre_ln"][1], None, None, 0, 1, T, D, c.layer_norm_eps)
        n = torch.empty(T, D, device=dev, dtype=bf)
        att = torch.empty(T, D, device=dev, dtype=bf)
        if self.flash:
            L_pad = (L + 127) // 128 * 128
            qk = torch.empty(T, 2 * D, device=dev, dtype=bf)
            vt = torch.zeros(B, D, L_pad, device=dev, dtype=bf)     # V^T written by a GEMM with swapped operands
        else:
            qkv = torch.empty(T, 3 * D, device=dev, dtype=bf)
        act_ = _lib.gelu_erf_ if c.hidden_act == "gelu" else _lib.quick_gelu_
        mid = torch.empty(T, M, device=dev, dtype=bf)
        for i in range(c.num_hidden_layers):
            p = "encoder.layers.%d." % i
            x, y = hs[i], hs[i + 1]
            _lib.layernorm_mod_f32(x, n, W[p + "layer_norm1"][0], W[p + "layer_norm1"][1], None, None, 0, 1, T, D,
                                   c.layer_norm_eps)
            if self.flash:
                _lib.gemm(n, W[p + "qk"], qk, T, 2 * D, D, D, D, 2 * D, bias=W[p + "qk_b"])
                _lib.gemm(W[p + "v"], n, vt, D, L, D, D, D, L_pad, bias=W[p + "v_b"], batch=B, strideB=L * D, strideC=D * L_pad,
                          flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
                _lib.flash_attn_d64(qk, qk, vt, att, B, H, L, L * 2 * D, 2 * D, D * L_pad, L_pad, L * D, D, dh ** -0.5, k_off=D)
            else:
                _lib.gemm(n, W[p + "qkv"], qkv, T, 3 * D, D, D, D, 3 * D, bias=W[p + "qkv_b"])
                _lib.attn_bias(qkv, att, None, None, None, B, H, L, scale=dh ** -0.5, head_dim=dh)
            _lib.gemm(att, W[p + "o"], y, T, D, D, D, D, D, bias=W[p + "o_b"], R=x, ldr=D)
            _lib.layernorm_mod_f32(y, n, W[p + "layer_norm2"][0], W[p + "layer_norm2"][1], None, None, 0, 1, T, D,
                                   c.layer_norm_eps)
            _lib.gemm(n, W[p + "fc1"], mid, T, M, D, D, D, M, bias=W[p + "fc1_b"])
            act_(mid)
            _lib.gemm(mid, W[p + "fc2"], y, T, D, M, M, M, D, bias=W[p + "fc2_b"], R=y, ldr=D)
        states = [hs[i].view(B, L, D) for i in range(n_states)]
        out = CLIPVisionOutput(last_hidden_state=states[-1], hidden_states=states if output_hidden_states else None)
        return out if return_dict else (out.last_hidden_state,)

    def to(self, *_, **__):
        return self

    def eval(self):
        return self
