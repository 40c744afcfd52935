#!/usr/bin/env python3
"""Times the plain GEMM forms of the C2 step with row-major and with fragment-packed weights (alg_pack_b_bf16), interleaved."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
M = 35552
cases = {"ff1 (N 12288, K 3072, GELU)": (12288, 3072, _lib.ACT_GELU_TANH), "qk (N 6144, K 3072)": (6144, 3072, 0),
         "wan ff1 (N 13824, K 5120, GELU)": (13824, 5120, _lib.ACT_GELU_TANH), "wan q (N 5120, K 5120)": (5120, 5120, 0)}
for name, (N, K, act) in cases.items():
    a = torch.randn(M, K, generator=g, device=dev).to(BF)
    w = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(BF)
    b = torch.randn(N, generator=g, device=dev).to(BF)
    pk = _lib.PackedB(w)
    c = torch.empty(M, N, dtype=BF, device=dev)
    res = {}
    for rep in range(3):
        for label, B in (("rows", w), ("packed", pk)):
            for _ in range(3):
                _lib.gemm(a, B, c, M, N, K, K, K, N, bias=b, act=act)
            torch.cuda.synchronize()
            t0 = time.time()
            R = 30
            for _ in range(R):
                _lib.gemm(a, B, c, M, N, K, K, K, N, bias=b, act=act)
            torch.cuda.synchronize()
            res.setdefault(label, []).append(2.0 * M * N * K * R / (time.time() - t0) / 1e12)
    print("%-34s rows %s   packed %s   %+.1f %%" % (name, " ".join("%.0f" % x for x in res["rows"]), " ".join("%.0f" % x for x in res["packed"]),
                                                   100.0 * (sum(res["packed"]) / sum(res["rows"]) - 1.0)))
