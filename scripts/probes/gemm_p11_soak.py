#!/usr/bin/env python3
"""Soak of GEMM schedule 11 at the C2 shapes: every call is repeated N times on the same inputs and every result compared on the device
with the first one (a rare race in the ring / register-set protocol would show as a run-to-run difference), and the first one with
schedule 10's row-major call.  usage: python scripts/probes/gemm_p11_soak.py [N=300]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from alg_amd import _lib  # noqa: E402

BF = torch.bfloat16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device=dev) * sc).to(BF)
    N, S, D, T = 2, 17776, 3072, 226
    F4 = 4 * D
    y, att, h = rn(N, S, D), rn(N, S, D), rn(N, S, F4, sc=0.3)
    x0 = rn(N, S, D)
    mod = rn(N, 12 * D, sc=0.1)
    wo, bo, wf1, bf1, wf2, bf2 = rn(D, D, sc=0.02), rn(D, sc=0.02), rn(F4, D, sc=0.02), rn(F4, sc=0.02), rn(D, F4, sc=0.02), rn(D, sc=0.02)
    cases = {
        "out": lambda w, c: _lib.gemm(att, w, c, S, D, D, D, D, D, bias=bo, R=x0, ldr=D, gate=mod, gate_off=4 * D, strideGate=12 * D, seg_split=T,
                                      batch=N, strideA=S * D, strideC=S * D, strideR=S * D),
        "ff1": lambda w, c: _lib.gemm(y, w, c, S, F4, D, D, D, F4, bias=bf1, act=_lib.ACT_GELU_TANH, batch=N, strideA=S * D, strideC=S * F4),
        "ff2": lambda w, c: _lib.gemm(h, w, c, S, D, F4, F4, F4, D, bias=bf2, R=x0, ldr=D, gate=mod, gate_off=10 * D, strideGate=12 * D, seg_split=T,
                                      batch=N, strideA=S * F4, strideC=S * D, strideR=S * D),
    }
    weights = {"out": wo, "ff1": wf1, "ff2": wf2}
    res = {}
    for name, call in cases.items():
        w = weights[name]
        shape = (N, S, F4) if name == "ff1" else (N, S, D)
        ref, first, c = torch.empty(shape, dtype=BF, device=dev), torch.empty(shape, dtype=BF, device=dev), torch.empty(shape, dtype=BF, device=dev)
        call(w, ref)
        pk = _lib.PackedB(w)
        call(pk, first)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(n):
            c.fill_(float("nan"))
            call(pk, c)
            bad += (c.view(torch.int16) != first.view(torch.int16)).any().to(torch.int64)
        torch.cuda.synchronize()
        res[name] = {"repeats": n, "runs_that_differ_from_the_first": int(bad.item()),
                     "first_equals_schedule10": bool(torch.equal(first, ref)), "finite": bool(torch.isfinite(first.float()).all())}
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
