#!/usr/bin/env python3
"""Does workgroup-count quantisation show in the d64 attention at the C2 sequence length?  Sweeps the number of heads
(=> workgroups = heads * N * ceil(S / 256)) and prints TFLOP/s per point; a sawtooth means a tail effect."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib

dev = torch.device("cuda:0")
BF = torch.bfloat16
S = 17776
S_pad = (S + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(0)
for N, heads in [(1, h) for h in (44, 46, 48, 50, 52, 55, 58, 62, 64)] + [(2, 48), (3, 48), (2, 40), (2, 44), (2, 52)]:
    D = heads * 64
    qk = torch.randn(N, S, 2 * D, generator=g, device=dev).to(BF)
    vt = torch.randn(N, D, S_pad, generator=g, device=dev).to(BF)
    att = torch.empty(N, S, D, dtype=BF, device=dev)
    fn = lambda: _lib.flash_attn_d64(qk, qk, vt, att, N, heads, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    wgs = N * heads * ((S + 255) // 256)
    print("N=%d heads=%d wgs=%d  wgs/256=%.2f  %.3f ms  %.1f TF  %.2f us/wg-slot" % (
        N, heads, wgs, wgs / 256, ms, 4.0 * N * heads * S * S * 64 / ms / 1e9, ms * 1e3 / (wgs / 256)), flush=True)
    del qk, vt, att
