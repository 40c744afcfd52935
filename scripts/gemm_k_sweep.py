#!/usr/bin/env python3
"""Per-tile overhead of the bf16 GEMM: same M x N (2 x 17,776 rows x 3072 columns = 1680 tiles), K swept; a fit of
time = tiles/256 * (nk * a + b) gives the steady-state k-tile time a and the per-tile prologue + epilogue b."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib

dev = torch.device("cuda:0")
BF = torch.bfloat16
N, S, D = 2, 17776, 3072
g = torch.Generator(device=dev).manual_seed(0)
res = []
VENDOR = "--vendor" in sys.argv
for K in (768, 1536, 3072, 6144, 12288):
    a = torch.randn(N, S, K, generator=g, device=dev).to(BF)
    w = (torch.randn(D, K, generator=g, device=dev) * 0.02).to(BF)
    x = torch.randn(N, S, D, generator=g, device=dev).to(BF)
    for mode in ("plain", "residual"):
        kw = dict(R=x, ldr=D, strideR=S * D) if mode == "residual" else {}
        fn = lambda: _lib.gemm(a, w, x, S, D, K, K, K, D, batch=N, strideA=S * K, strideC=S * D, **kw)
        if VENDOR:   # the vendor library on the same tensors (measurement only)
            if mode == "residual":
                continue
            fn = lambda: torch.nn.functional.linear(a.view(N * S, K), w)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("K=%5d %-8s %.3f ms  %.1f TFLOP/s  %.2f us per k-tile-round" % (
            K, mode, ms, 2.0 * N * S * D * K / ms / 1e9, ms * 1e3 / (K / 64) / (1680 / 256)), flush=True)
        res.append((K, mode, ms))
    del a, w, x
for mode in (("plain",) if VENDOR else ("plain", "residual")):
    pts = [(K / 64, ms) for K, m, ms in res if m == mode]
    (n1, t1), (n2, t2) = pts[1], pts[-1]
    a = (t2 - t1) / (n2 - n1)
    b = t1 - a * n1
    rounds = 1680 / 256
    print("%s: a = %.3f us per k-tile (steady state %.0f TFLOP/s), b = %.1f us per tile = %.1f k-tiles" % (
        mode, a * 1e3 / rounds, 2 * 256 * 256 * 64 * 256 / (a / rounds * 1e-3) / 1e12, b * 1e3 / rounds, b / a))
