#!/usr/bin/env python3
"""Bit-compare the d = 64 pre-scaled attention of several library builds on the same tensors (each build in its own process).
usage: python scripts/attn_lib_compare.py out.pt   (ALG_HIP_LIB selects the build; prints a checksum, saves the output)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib  # noqa: E402

S, N, H, D = 17776, 2, 48, 3072
dev, BF = torch.device("cuda:0"), torch.bfloat16
S_pad = (S + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(3)
qk = torch.randn(N, S, 2 * D, generator=g, device=dev).to(BF)
qk.view(N, S, 2, D)[:, :, 0] *= 0.125 * 1.4426950408889634
qk[0, 5000:5003] *= 30.0          # a few rows / keys far out: non-zero offsets and a refused tile inside the statement
vt = torch.zeros(N, D, S_pad, dtype=BF, device=dev)
vt[:, :, :S] = torch.randn(N, D, S, generator=g, device=dev).to(BF)
o = torch.empty(N, S, D, dtype=BF, device=dev)
outs = []
for _ in range(3):
    o.fill_(7.0)
    _lib.flash_attn_d64(qk, qk, vt, o, N, H, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D, q_prescaled=True)
    torch.cuda.synchronize()
    outs.append(o.clone())
assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
torch.save(outs[0].cpu(), sys.argv[1])
print(os.environ.get("ALG_HIP_LIB", "default"), "checksum", int(outs[0].view(torch.int16).to(torch.int64).sum().item()), "finite", bool(torch.isfinite(outs[0].float()).all()))
