#!/usr/bin/env python3
"""Ping-pong attention (ALG_ATTN_PP = 1 / 2: flash_attn_d64_kernel<42 / 43>) against the straight loop (<41>) at the C2 shape,
pre-scaled Q (the product's call): bit identity (same per-wave arithmetic, only the phase order across waves differs), then
timing.  python scripts/experiments/attn_pp_ab.py [rounds]"""
import os
import sys

import torch

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


def run(pp, small=None):
    os.environ["ALG_ATTN_PP"] = str(pp if pp < 10 else 0)
    os.environ["ALG_ATTN64_Q64"] = "1" if pp == 64 else "0"      # pp = 64: the 64-queries-per-wave kernel
    s = small or S
    att = torch.empty(N, s, H * D, dtype=BF, device=dev)
    _lib.flash_attn_d64(qk, qk, vt, att, N, H, s, S * 2 * H * D, 2 * H * D, H * D * S_pad, S_pad, s * H * D, H * D, 0.125,
                        k_off=H * D, q_prescaled=True)
    return att


def timeit(pp, iters=6):
    run(pp); run(pp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(pp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 4.0 * N * H * S * S * 64 / ms / 1e9


bad = 0
PPS = tuple(int(x) for x in os.environ.get('PPS', '1,2').split(','))
for small in (64, 100, 256, 640, 1000, 4097, S):
    ref = run(0, small)
    for pp in PPS:
        for rep in range(2):
            got = run(pp, small)
            if pp in (3, 4, 5, 64):   # row sums of unrounded probabilities inside the statement: close, not bit-equal
                err = (got.float() - ref.float()).abs().max().item()
                print('pp', pp, 'S', small, 'max abs diff vs straight', err, flush=True)
                if not (err < 2e-2):
                    bad += 1
                break
            if not torch.equal(got, ref):
                d = (got.float() - ref.float()).abs()
                bad += 1
                print("MISMATCH pp", pp, "S", small, "max", d.max().item(), "count", int((d > 0).sum()), flush=True)
                break
print("ping-pong vs straight loop: %d mismatching" % bad, flush=True)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for pp in (0,) + PPS:
        ms, tf = timeit(pp)
        print("round %d ALG_ATTN_PP=%d  %.3f ms  %.1f TFLOP/s" % (r, pp, ms, tf), flush=True)
