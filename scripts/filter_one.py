#!/usr/bin/env python3
"""One low-pass case in a loop (for rocprofv3 --pmc): python scripts/filter_one.py [c2x8|c5x8|wanx8g] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import lp_utils  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "c2x8"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator().manual_seed(5)
if case == "c2x8":
    x, args = torch.randn(8, 13, 16, 60, 90, generator=g).to(torch.bfloat16), ("down_up", 0.0, 0, 0.25)
elif case == "c5x8":
    x, args = torch.randn(8, 20, 21, 90, 160, generator=g), ("down_up", 0.0, 0, 0.4)
else:
    x, args = torch.randn(8, 20, 21, 60, 104, generator=g), ("gaussian_blur", 15.0, 9, 1.0)
x = x.cuda()
for _ in range(iters):
    lp_utils.apply_low_pass_filter(x, *args)
torch.cuda.synchronize()
