"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch) of the e4m3 operand arithmetic of BASELINE config 5
("Wan2.1-I2V-14B fp8 weights (CDNA4 fp8 MFMA)").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; nothing under alg_amd/ does.

What it stands in for.  The reference picks the transformer's dtype from the YAML (`/root/reference/run.py:38` dtype string ->
`run.py:59-61` `WanTransformer3DModel.from_pretrained(..., torch_dtype=dtype)`) and runs every `nn.Linear` of the DiT behind
`pipeline_wan_image2video_lowpass.py:910-917` in it.  PyTorch has no fp8 `nn.Linear`; what "fp8 weights" means on a matrix core
is the scaled-operand scheme restated here, in torch eager op order:

    weights      one scale per OUTPUT CHANNEL (row of W [N, K]):  s_w[n] = amax_k |W[n, k]| / 448,  Wq = e4m3(W / s_w)
    activations  one scale per TOKEN (row of x [M, K]), re-derived in front of every linear:  s_x[m] = amax_k |x[m, k]| / 448
    product      y[m, n] = (sum_k xq[m, k] * Wq[n, k]) * s_x[m] * s_w[n] + bias[n]   (e4m3 x e4m3 products are exact in fp32;
                 the sum is an fp32 accumulation), rounded once to the activation dtype

with OCP e4m3 (`torch.float8_e4m3fn`: 4 exponent bits, 3 mantissa bits, max 448, no infinities), round-to-nearest-even, values
clamped to +-448 before the cast (the cast itself does not saturate), an all-zero row keeping scale 1.  The seven large linears
of a Wan block take this path (attn1.to_q / to_k / to_v / to_out.0, attn2.to_q / to_out.0, ffn.net.0.proj, ffn.net.2);
the text / image K and V projections of the cross-attention, the embedders, norms, attention and the residual stream stay in
the activation dtype -- `wan_oracle.wan_forward(..., fp8=True)` routes exactly those seven through `linear`.

PINNING.  e4m3 is a published format (OCP 8-bit floating point specification v1.0) and `torch.float8_e4m3fn` is PyTorch's
implementation of it: `tests/test_fp8_oracle_cpu.py` checks this file's `e4m3_round` (a from-the-definition rounding written with
integer / frexp arithmetic) against torch's cast on every one of the 256 codes and on a dense sweep, so the number format is
pinned by the third-party package itself.  The SCHEME (per-token / per-channel scales, amax / 448) is this build's choice --
the reference has no fp8 code to pin it against; parity of the scheme is therefore a statement about the product and its
oracle only, and says so wherever it is reported.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

E4M3_MAX = 448.0
F8 = torch.float8_e4m3fn


def e4m3_round(x: torch.Tensor) -> torch.Tensor:
    """Round float32 values to the nearest OCP e4m3 value (ties to even), from the format's definition: normal numbers
    2^e * (1 + m / 8) for e in [-6, 8], m in [0, 8) (the top code 2^8 * 1.875 = 480 is NaN, so the largest finite value is
    448), subnormals m * 2^-9.  Inputs are clamped to +-448 first.  Returns float32."""
    x = x.float().clamp(-E4M3_MAX, E4M3_MAX)
    mant, exp = torch.frexp(x)                       # x = mant * 2^exp, 0.5 <= |mant| < 1
    e = (exp - 1).clamp(min=-6)                      # exponent of the leading bit, floored at the subnormal exponent
    step = torch.ldexp(torch.ones_like(x), e - 3)    # spacing of e4m3 values in that binade: 2^(e - 3)
    y = torch.round(x / step) * step                 # torch.round is round-half-to-even; x / step is exact (power of two)
    return torch.where(x == 0, x, y).clamp(-E4M3_MAX, E4M3_MAX)


def quantize_rows(x: torch.Tensor, via_torch_cast: bool = True):
    """Row-wise e4m3 quantisation of x [..., K] -> (values as float32 [..., K] that are exactly representable in e4m3,
    scale float32 [...]).  The arithmetic order is the product's (alg_quantize_fp8_rows): amax in fp32, scale = amax * (1 / 448)
    rounded to fp32, reciprocal of the scale rounded to fp32, x * reciprocal in fp32, clamp, round to e4m3."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1)
    scale = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    inv = 1.0 / scale
    y = (xf * inv.unsqueeze(-1)).clamp(-E4M3_MAX, E4M3_MAX)
    q = y.to(F8).float() if via_torch_cast else e4m3_round(y)
    return q, scale


def dequantize(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q * scale.unsqueeze(-1)


def linear(x: torch.Tensor, weight: torch.Tensor, bias, out_dtype=None, chunk_rows: int = 16384) -> torch.Tensor:
    """y = e4m3(x per token) @ e4m3(weight per output channel)^T * scales + bias, fp32 accumulation, one rounding to
    `out_dtype` (default: x's dtype) -- F.linear with both operands quantise-dequantised.  Works on any device (the
    full-size C5 test runs it with torch's own fp32 ops on the GPU); rows are processed in chunks to bound memory."""
    out_dtype = out_dtype or x.dtype
    qw, sw = quantize_rows(weight)
    wd = dequantize(qw, sw)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    out = torch.empty(x2.shape[0], weight.shape[0], dtype=out_dtype, device=x.device)
    b = None if bias is None else bias.float()
    for r0 in range(0, x2.shape[0], chunk_rows):
        qx, sx = quantize_rows(x2[r0:r0 + chunk_rows])
        out[r0:r0 + chunk_rows] = F.linear(dequantize(qx, sx), wd, b).to(out_dtype)
    return out.reshape(*lead, weight.shape[0])


def quantization_error_bound(K: int) -> float:
    """Relative L2 error one expects from rounding both operands of a K-long dot product of independent zero-mean values to
    e4m3: each rounding has relative error uniform in +-2^-4 (3 mantissa bits) -> std 2^-4 / sqrt(3) per factor, sqrt(2) for
    the product; errors of the K terms are independent, so the relative error of the sum equals that of a term.  ~5.1 %."""
    return math.sqrt(2.0) * 2.0 ** -4 / math.sqrt(3.0)
