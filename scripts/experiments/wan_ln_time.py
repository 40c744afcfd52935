#!/usr/bin/env python3
"""Times Wan's modulated LayerNorm (norm1 / norm3) at the C3 / C5 token counts.  ALG_HIP_LIB selects the build."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
D = 5120
for N, S in ((3, 32760), (2, 32760), (3, 75600)):
    x = torch.randn(N, S, D, device=dev).to(BF)
    y = torch.empty_like(x)
    mod = torch.randn(N, 2, D, device=dev) * 0.3
    call = lambda: _lib.layernorm_mod_f32(x, y, None, None, mod, mod, 2 * D, N, S, D, 1e-6, scale_off=0, shift_off=D)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(100):
        call()
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 10
    print("N %d S %6d  %.3f ms  %.0f GB/s" % (N, S, ms, 2 * N * S * D * 2 / ms / 1e6))
