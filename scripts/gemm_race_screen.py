#!/usr/bin/env python3
"""Race screen for the default (ping-pong, counted-vmcnt) GEMM schedule: its results must equal the drain-and-barrier
schedule (ALG_GEMM_PIPE=0) bit for bit -- both accumulate K in the same order -- over random shapes and over repeated runs
of large shapes (an LDS-DMA / barrier ordering bug shows up as rare wrong tiles that come and go)."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
rng = random.Random(0)
g = torch.Generator(device=dev).manual_seed(0)


def run(pipe, a, w, bias, M, N, K):
    os.environ["ALG_GEMM_PIPE"] = str(pipe)     # 0 exists in the EXPERIMENTS build only (ALG_HIP_LIB=alg_amd/libalg_hip_exp.so)
    _lib.reload_env()                           # the library reads its options once, at load
    c = torch.empty(M, N, dtype=BF, device=dev)
    _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH)
    return c


bad = 0
cases = [(rng.randint(1, 4000), rng.randint(1, 375) * 4, rng.randint(1, 64) * 64) for _ in range(150)]
cases += [(35552, 3072, 3072)] * 10 + [(17776, 3072, 12288)] * 10 + [(5000, 12288, 3072)] * 5
for i, (M, N, K) in enumerate(cases):
    a = torch.randn(M, K, generator=g, device=dev).to(BF)
    w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(BF)
    bias = torch.randn(N, generator=g, device=dev).to(BF)
    ref = run(0, a, w, bias, M, N, K)
    for rep in range(3):
        got = run(6, a, w, bias, M, N, K)
        if not torch.equal(got, ref):
            bad += 1
            d = (got.float() - ref.float()).abs()
            print("MISMATCH", (M, N, K), "rep", rep, "max", d.max().item(), "count", int((d > 0).sum()))
            break
torch.cuda.synchronize()
print("race screen: %d shapes x 3 runs, %d mismatching" % (len(cases), bad))
sys.exit(1 if bad else 0)
