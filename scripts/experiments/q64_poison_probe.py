#!/usr/bin/env python3
"""Does a kernel read a register (or LDS word) it never wrote?  A poison kernel (tests/helpers/reg_poison.hip) rewrites the
whole register file / LDS of every CU with one bit pattern right in front of the kernel under test, launch after launch; the
output is compared with a reference made without poison.  A kernel that initialises everything it reads cannot see the
difference.  (Round 4: the 64-queries-per-wave d = 128 attention fails only in the first round of workgroups behind OTHER
kernels -- what those kernels leave in the registers is the one thing a same-kernel stress loop never varies.)

    [ALG_HIP_LIB=alg_amd/libalg_hip_exp.so] python scripts/experiments/q64_poison_probe.py KERNEL [iters] [S]
KERNEL: d128_pipe | d128_q64:<arm> | d64_pipe | gemm9 | gemm9_res
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alg_amd import _lib  # noqa: E402

kernel = sys.argv[1] if len(sys.argv) > 1 else "d128_pipe"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev, BF = torch.device("cuda:0"), torch.bfloat16
poison = __import__("tests.helpers.poison", fromlist=["load"]).load()
g = torch.Generator(device=dev).manual_seed(5)
rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device=dev) * sc).to(BF)


def select(name, value):
    os.environ[name] = value
    _lib.reload_env()


if kernel.startswith("d128"):
    Hh, N = 40, 1
    D = Hh * 128
    S_pad = (S + 63) // 64 * 64
    qk = rn(N, S, 2 * D)
    vt = torch.zeros(N, D, S_pad, dtype=BF, device=dev)
    vt[:, :, :S] = rn(N, D, S)
    out = torch.empty(N, S, D, dtype=BF, device=dev)
    if kernel.startswith("d128_q64"):
        select("ALG_ATTN128_Q64", kernel.split(":")[1])
    run = lambda: _lib.flash_attn_d128(qk, qk, vt, out, N, Hh, S, S, S * 2 * D, 2 * D, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D,
                                       128 ** -0.5, k_off=D)
elif kernel == "d64_pipe":
    Hh, N = 48, 1
    D = Hh * 64
    S_pad = (S + 63) // 64 * 64
    qk = rn(N, S, 2 * D)
    qk.view(N, S, 2, D)[:, :, 0] *= 0.125 * 1.4426950408889634
    vt = torch.zeros(N, D, S_pad, dtype=BF, device=dev)
    vt[:, :, :S] = rn(N, D, S)
    out = torch.empty(N, S, D, dtype=BF, device=dev)
    run = lambda: _lib.flash_attn_d64(qk, qk, vt, out, N, Hh, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D,
                                      q_prescaled=True)
else:
    M, Nn, K = S, 3072, 3072
    a, w, bias = rn(M, K), rn(Nn, K, sc=0.02), rn(Nn)
    out = rn(M, Nn)
    x0 = out.clone()
    gate = rn(1, 2 * Nn, sc=0.5)
    if kernel == "gemm9_res":
        def run():
            out.copy_(x0)
            _lib.gemm(a, w, out, M, Nn, K, K, K, Nn, bias=bias, R=out, ldr=Nn, gate=gate, strideGate=2 * Nn, seg_split=226)
    else:
        run = lambda: _lib.gemm(a, w, out, M, Nn, K, K, K, Nn, bias=bias, act=_lib.ACT_GELU_TANH)

run()
torch.cuda.synchronize()
ref = out.clone()
for _ in range(3):          # same-kernel repeats: the classic stress loop
    run()
    assert torch.equal(out, ref), "not deterministic even without poison"
stream = torch.cuda.current_stream().cuda_stream
report = {"kernel": kernel, "S": S, "iters": iters, "experiments": _lib.experiments_build(), "arms": []}
PATTERNS = {"nan": 0x7FC00000, "big": 0x7F000000, "neg": 0xFF000000, "ones": 0x3F803F80, "allbits": 0xFFFFFFFF, "alt": 0xAAAAAAAA,
            "zero": 0, "eighty": 0x42A042A0, "minus80": 0xC2A0C2A0}
for parts, pname in [(0, "nan"), (31, "nan"), (31, "big"), (31, "neg"), (31, "ones"), (1, "nan"), (2, "nan"), (4, "nan"), (8, "nan"),
                     (16, "nan"), (32, "allbits"), (32, "alt"), (32, "zero"), (63, "allbits"), (63, "zero"), (63, "eighty"), (16, "eighty"), (16, "minus80"), (31, "eighty")]:
    bad, worst, nonfinite = 0, 0.0, 0
    for i in range(iters):
        rc = poison.reg_poison(PATTERNS[pname], parts, 512, stream)
        assert rc == 0, rc
        run()
        if not torch.equal(out, ref):
            bad += 1
            d = (out.float() - ref.float())
            nonfinite += int((~torch.isfinite(d)).sum().item() > 0)
            worst = max(worst, float(torch.nan_to_num(d.abs(), nan=0.0, posinf=0.0).max()))
    report["arms"].append({"parts": parts, "pattern": pname, "bad": bad, "nonfinite_launches": nonfinite, "worst_finite": round(worst, 4)})
    print(json.dumps(report["arms"][-1]), flush=True)
print(json.dumps(report))
