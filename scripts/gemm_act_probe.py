#!/usr/bin/env python3
"""Cost of the activation epilogues on the ff1 shape (2 x 17,776 x 3072 -> 12288): none vs tanh-GELU vs SiLU.
Round 1: none 1105-1173, gelu_tanh 1051-1055, silu 1079 TFLOP/s -- the GELU epilogue (13 VALU + 2 transcendental per
element, 128 elements per thread per tile) costs 5-10 % of ff1, i.e. ~1 % of the step."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alg_amd import _lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
N, S, D = 2, 17776, 3072
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(N, S, D, generator=g, device=dev).to(BF)
w = (torch.randn(4 * D, D, generator=g, device=dev) * 0.02).to(BF)
b = torch.randn(4 * D, generator=g, device=dev).to(BF)
h = torch.empty(N, S, 4 * D, dtype=BF, device=dev)
for name, act in (("none", _lib.ACT_NONE), ("gelu_tanh", _lib.ACT_GELU_TANH), ("silu", _lib.ACT_SILU), ("none", _lib.ACT_NONE), ("gelu_tanh", _lib.ACT_GELU_TANH)):
    fn = lambda: _lib.gemm(a, w, h, S, 4 * D, D, D, D, 4 * D, bias=b, act=act, batch=N, strideA=S * D, strideC=S * 4 * D)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-10s %.3f ms  %.1f TFLOP/s" % (name, ms, 2.0 * N * S * D * 4 * D / ms / 1e9), flush=True)
