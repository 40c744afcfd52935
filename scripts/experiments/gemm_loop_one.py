#!/usr/bin/env python3
"""One plain bias GEMM looping for a few seconds (for power_probe.sh): gemm_loop_one.py [vendor|ours] [qk|ff1|ff2] [secs].
Measurement only (the product path never calls the vendor library)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

who, shape, secs = sys.argv[1], sys.argv[2], float(sys.argv[3])
M, N, K = {"qk": (35552, 6144, 3072), "ff1": (35552, 12288, 3072), "ff2": (35552, 3072, 12288)}[shape]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, K, generator=g, device=dev).bfloat16()
w = (torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()
b = torch.randn(N, generator=g, device=dev).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
fn = (lambda: F.linear(a, w, b)) if who == "vendor" else (lambda: _lib.gemm(a, w, c, M, N, K, K, K, N, bias=b))
fn(); torch.cuda.synchronize()
t0 = time.time(); rates = []
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    rates.append(2.0 * M * N * K * 20 / (e0.elapsed_time(e1) / 1e3) / 1e12)
print("%s %s pipe %s TFLOP/s first/min/median/last %.1f %.1f %.1f %.1f" % (who, shape, os.environ.get("ALG_GEMM_PIPE", "-"), rates[0], min(rates),
      sorted(rates)[len(rates) // 2], rates[-1]))
