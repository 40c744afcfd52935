#!/usr/bin/env python3
"""One GEMM shape in a loop for rocprofv3 --pmc: python vendor_pmc_one.py [vendor|ours] [qk|ff1|ff2] [iters].
Measurement only (the product path never calls the vendor library)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

who = sys.argv[1] if len(sys.argv) > 1 else "ours"
shape = sys.argv[2] if len(sys.argv) > 2 else "ff2"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
M, N, K = {"qk": (35552, 6144, 3072), "ff1": (35552, 12288, 3072), "ff2": (35552, 3072, 12288)}[shape]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, K, generator=g, device=dev).bfloat16()
w = (torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()
b = torch.randn(N, generator=g, device=dev).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    if who == "vendor":
        F.linear(a, w, b)
    else:
        _lib.gemm(a, w, c, M, N, K, K, K, N, bias=b)
torch.cuda.synchronize()
