#!/usr/bin/env python3
"""GEMM schedule 9 (asm main loop) against schedule 0 (drain-and-barrier, compiler-scheduled): bit identity over shapes that
exercise edge tiles, every epilogue form and the ring's wrap-around (K / 64 = 2 .. 200), then timing against schedule 6.
python scripts/experiments/p9_check.py [--time]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)


def run(pipe, fn):
    os.environ["ALG_GEMM_PIPE"] = str(pipe)
    return fn()


def forms(M, N, K):
    a, w, bias = rn(M, K), rn(N, K, sc=0.05), rn(N)
    x0 = rn(M, N)
    gate = rn(1, 2 * N, sc=0.5)
    brow = rn(M)

    def plain():
        c = torch.full((M, N), 7.0, dtype=BF, device=dev)
        _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias)
        return c

    def gelu():
        c = torch.full((M, N), 7.0, dtype=BF, device=dev)
        _lib.gemm(a, w, c, M, N, K, K, K, N, bias=bias, act=_lib.ACT_GELU_TANH)
        return c

    def silu():
        c = torch.full((M, N), 7.0, dtype=BF, device=dev)
        _lib.gemm(a, w, c, M, N, K, K, K, N, act=_lib.ACT_SILU)
        return c

    def res():
        x = x0.clone()
        _lib.gemm(a, w, x, M, N, K, K, K, N, bias=bias, R=x, ldr=N, gate=gate, strideGate=2 * N, seg_split=M // 3)
        return x

    def vt():   # the V^T projection: per-row bias, permuted columns, padded pitch
        npad = (N + 63) // 64 * 64
        c = torch.zeros(M, npad, dtype=BF, device=dev)
        _lib.gemm(a, w, c, M, N, K, K, K, npad, bias=brow, flags=_lib.GEMM_BIAS_PER_ROW | _lib.GEMM_PERMUTE_COLS)
        return c

    return {"plain": plain, "gelu": gelu, "silu": silu, "res": res, "vt": vt}


rng = random.Random(3)
cases = [(256, 256, 128), (512, 768, 256), (300, 520, 128), (17, 64, 512), (2, 1000, 192), (1111, 96, 3072), (70, 250, 192),
         (33, 8, 128), (257, 257 * 4, 64 * 7), (4096, 1024, 64 * 11), (1000, 3072, 12288)]
cases += [(rng.randint(1, 3000), rng.randint(1, 300) * 4, rng.randint(2, 48) * 64) for _ in range(30)]
bad = 0
for M, N, K in cases:
    for name, fn in forms(M, N, K).items():
        if name == "vt" and N % 4:
            continue
        ref = run(0, fn)
        for rep in range(2):
            got = run(9, fn)
            if not torch.equal(got, ref):
                d = (got.float() - ref.float()).abs()
                bad += 1
                print("MISMATCH", name, (M, N, K), "rep", rep, "max", d.max().item(), "count", int((d > 0).sum()),
                      "first", (d > 0).nonzero()[:3].tolist(), flush=True)
                break
print("p9 check: %d shapes x 5 epilogue forms, %d mismatching" % (len(cases), bad), flush=True)
big = [(35552, 3072, 3072)] * 3 + [(17776, 3072, 12288)] * 3 + [(17776, 12288, 3072)] * 2
for M, N, K in big:
    f = forms(M, N, K)
    for name in ("plain", "res"):
        ref = run(6, f[name])
        for rep in range(3):
            if not torch.equal(run(9, f[name]), ref):
                bad += 1
                print("MISMATCH big", name, (M, N, K), flush=True)
                break
print("p9 check incl. C2 shapes vs schedule 6: %d mismatching" % bad, flush=True)
if "--time" in sys.argv:
    import subprocess
    for pipe in ("6", "9", "6", "9"):
        env = dict(os.environ, ALG_GEMM_PIPE=pipe)
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kbench.py"),
                              "--only", "gemm_qk,gemm_vt,gemm_out,gemm_ff1,gemm_ff2", "--iters", "20"], env=env,
                             capture_output=True, text=True).stdout
        print("== ALG_GEMM_PIPE=%s\n%s" % (pipe, out), flush=True)
sys.exit(1 if bad else 0)
