#!/usr/bin/env python3
"""d = 64 attention at the C2 shape with model-like scores: every ALG_ATTN_PP main launch against fp32 SDPA rows (torch as the
checker) and against each other.  usage: python scripts/attn_pp_compare.py [S] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 17776
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev, BF, D, H = torch.device("cuda:0"), torch.bfloat16, 3072, 48
S_pad = (S + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(0)
qk = torch.randn(N, S, 2 * D, generator=g, device=dev).to(BF)
qk.view(N, S, 2, D)[:, :, 0] *= 0.125 * 1.4426950408889634
vt = torch.zeros(N, D, S_pad, dtype=BF, device=dev)
vt[:, :, :S] = torch.randn(N, D, S, generator=g, device=dev).to(BF)
outs = {}
for pp in ("4", "6", "0"):
    os.environ["ALG_ATTN_PP"] = pp
    _lib.reload_env()
    att = torch.full((N, S, D), 7.0, dtype=BF, device=dev)
    _lib.flash_attn_d64(qk, qk, vt, att, N, H, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D, q_prescaled=True)
    torch.cuda.synchronize()
    outs[pp] = att
perm = torch.tensor([(n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1) for n in range(S)], device=dev)
worst = {}
for b, hh in ((0, 0), (N - 1, 5), (N - 1, 47)):
    q = qk[b, :, hh * 64:(hh + 1) * 64].float()
    k = qk[b, :, D + hh * 64:D + (hh + 1) * 64].float()
    v = vt[b, hh * 64:(hh + 1) * 64][:, perm].t().float()
    rows = torch.cat([torch.arange(0, 600, device=dev), torch.arange(S - 600, S, device=dev), torch.randint(0, S, (400,), device=dev)])
    ref = torch.softmax(q[rows] @ k.t() * 0.6931471805599453, dim=-1) @ v        # scores are in log2 units
    for pp, att in outs.items():
        err = (att[b, rows, hh * 64:(hh + 1) * 64].float() - ref).abs().max().item()
        worst[pp] = max(worst.get(pp, 0.0), err)
print("max |out - fp32 SDPA| over sampled rows:", {k: round(v, 5) for k, v in worst.items()})
for pp in ("6", "0"):
    d = (outs[pp].float() - outs["4"].float()).abs()
    print("PP=%s vs PP=4: max |diff| %.5f, differing elements %.4f %%, finite %s" % (pp, d.max().item(), 100.0 * (d > 0).float().mean().item(),
                                                                                 bool(torch.isfinite(outs[pp].float()).all())))
