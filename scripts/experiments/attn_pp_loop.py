#!/usr/bin/env python3
"""The C2 self-attention call looping for a few seconds (for power_probe.sh / rocprofv3): attn_pp_loop.py <ALG_ATTN_PP> <secs>"""
import os
import sys
import time

import torch

os.environ["ALG_ATTN_PP"] = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
N, H, S, D = 2, 48, 17776, 64
S_pad = (S + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(0)
qk = (torch.randn(N, S, 2 * H * D, generator=g, device=dev) * 0.5).to(BF)
vt = torch.zeros(N, H * D, S_pad, dtype=BF, device=dev)
vt[:, :, :S] = torch.randn(N, H * D, S, generator=g, device=dev).to(BF)
att = torch.empty(N, S, H * D, dtype=BF, device=dev)
fn = lambda: _lib.flash_attn_d64(qk, qk, vt, att, N, H, S, S * 2 * H * D, 2 * H * D, H * D * S_pad, S_pad, S * H * D, H * D, 0.125,
                                 k_off=H * D, q_prescaled=True)
fn(); torch.cuda.synchronize()
t0 = time.time(); rates = []
while time.time() - t0 < float(sys.argv[2]):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    rates.append(4.0 * N * H * S * S * 64 * 10 / (e0.elapsed_time(e1) / 1e3) / 1e12)
print("ALG_ATTN_PP=%s TFLOP/s first/min/median/last %.1f %.1f %.1f %.1f" % (sys.argv[1], rates[0], min(rates), sorted(rates)[len(rates) // 2], rates[-1]))
