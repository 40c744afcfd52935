#!/usr/bin/env python3
"""Times alg_quantize_fp8_rows at the C5 shapes.  ALG_HIP_LIB selects the build."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
for rows, K in ((75600, 5120), (75600, 13824), (2 * 75600, 13824)):
    x = torch.randn(rows, K, device=dev).to(BF)
    q = torch.empty(rows, K, dtype=torch.uint8, device=dev)
    s = torch.empty(rows, dtype=torch.float32, device=dev)
    for _ in range(5):
        _lib.quantize_fp8_rows(x, q, s, rows, K)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50):
        _lib.quantize_fp8_rows(x, q, s, rows, K)
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 20
    print("rows %6d K %5d  %.3f ms  %.0f GB/s (read + write once)" % (rows, K, ms, rows * K * 3 / ms / 1e6))
