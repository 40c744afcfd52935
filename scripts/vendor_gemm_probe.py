#!/usr/bin/env python3
"""Probe: the vendor library (hipBLASLt through torch.nn.functional.linear) on the plain bias GEMMs of the C2 step, next
to alg_gemm_bf16 on the same tensors.  Measurement only -- the product path does not call torch for compute."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib  # noqa: E402


def bench(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K in (("gemm_qk", 35552, 6144, 3072), ("gemm_ff1_no_act", 35552, 12288, 3072),
                          ("gemm_ff2_no_res", 35552, 3072, 12288)):
        a = torch.randn(M, K, generator=g, device=dev).bfloat16()
        w = (torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()
        b = torch.randn(N, generator=g, device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_v = bench(lambda: F.linear(a, w, b))
        t_o = bench(lambda: _lib.gemm(a, w, c, M, N, K, K, K, N, bias=b))
        fl = 2.0 * M * N * K / 1e9
        print("%-16s vendor %.3f ms %7.1f TFLOP/s | alg_gemm_bf16 %.3f ms %7.1f TFLOP/s" % (name, t_v, fl / t_v, t_o, fl / t_o),
              flush=True)


if __name__ == "__main__":
    main()
