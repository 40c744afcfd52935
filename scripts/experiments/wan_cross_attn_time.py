#!/usr/bin/env python3
"""Times Wan's two cross-attention calls (512 text keys, 257 image keys; d = 128, 40 heads) and the add behind them at the C3 / C5
token counts.  usage: python scripts/experiments/wan_cross_attn_time.py [S]   (ALG_ATTN128_Q64 etc. select kernels)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32760
N, H, D = 1, 40, 5120
dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(5)
q = torch.randn(N, S, D, generator=g, device=dev).to(BF)
out = {}
for name, n_kv in (("text", 512), ("image", 257)):
    pad = (n_kv + 127) // 128 * 128
    k = torch.randn(N, n_kv, D, generator=g, device=dev).to(BF)
    vt = torch.zeros(N, D, pad, dtype=BF, device=dev)
    v = torch.randn(N, D, n_kv, generator=g, device=dev).to(BF)
    perm = torch.tensor([(i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1) for i in range(pad)], device=dev)   # V^T columns: key index bits 2 <-> 3
    vt[:, :, perm[:n_kv]] = v
    o = torch.empty(N, S, D, dtype=BF, device=dev)
    call = lambda: _lib.flash_attn_d128(q, k, vt, o, N, H, S, n_kv, S * D, D, n_kv * D, D, D * pad, pad, S * D, D, 128 ** -0.5)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.time()
    R = 50
    for _ in range(R):
        call()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / R * 1e3
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.view(N, S, H, 128).transpose(1, 2)[:, :, :4096].float(), k.view(N, n_kv, H, 128).transpose(1, 2).float(),
        v.view(N, H, 128, n_kv).transpose(2, 3).float())
    err = (o.view(N, S, H, 128).transpose(1, 2)[:, :, :4096].float() - ref).abs().max().item()
    print("%-6s n_kv %4d  %.3f ms  %.0f TFLOP/s  max err %.2e" % (name, n_kv, ms, 4.0 * N * H * S * n_kv * 128 / ms / 1e9, err))
    out[name] = o
t0 = time.time()
for _ in range(50):
    _lib.lincomb([(1.0, out["text"]), (1.0, out["image"])], BF, out=out["text"])
torch.cuda.synchronize()
print("add    %.3f ms" % ((time.time() - t0) / 50 * 1e3))
# the two as ONE launch (alg_flash_attn_d128_dual)
ks, vts = {}, {}
g = torch.Generator(device=dev).manual_seed(6)
for name, n_kv in (("text", 512), ("image", 257)):
    pad = (n_kv + 127) // 128 * 128
    ks[name] = torch.randn(N, n_kv, D, generator=g, device=dev).to(BF)
    vts[name] = torch.zeros(N, D, pad, dtype=BF, device=dev)
    vts[name][:, :, :n_kv] = torch.randn(N, D, n_kv, generator=g, device=dev).to(BF)
o = torch.empty(N, S, D, dtype=BF, device=dev)
dual = lambda: _lib.flash_attn_d128_dual(q, ks["text"], vts["text"], 512, 512 * D, D, D * 512, 512, ks["image"], vts["image"], 257,
                                         257 * D, D, D * 384, 384, o, N, H, S, S * D, D, S * D, D, 128 ** -0.5)
for _ in range(5):
    dual()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(50):
    dual()
torch.cuda.synchronize()
ms = (time.time() - t0) / 50 * 1e3
print("dual   %.3f ms  %.0f TFLOP/s" % (ms, 4.0 * N * H * S * 769 * 128 / ms / 1e9))
