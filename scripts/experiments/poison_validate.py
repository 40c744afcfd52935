#!/usr/bin/env python3
"""Validation of tests/helpers/reg_poison.hip: a peek kernel reads v200 / a200 without writing them, right behind the poison."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = __import__("tests.helpers.poison", fromlist=["load"]).load()
out = torch.zeros(256 * 256 * 2, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(4096, 4096, device="cuda")
for pat in (0x7FC00000, 0x12345678):
    for between in (False, True):
        lib.reg_poison(pat, 31, 512, st)
        if between:
            (x @ x).sum().item()     # other kernels in between (rocBLAS + a reduction)
        lib.reg_peek(out.data_ptr(), 256, st)
        torch.cuda.synchronize()
        o = out.view(-1, 2)
        want = pat - (1 << 32) if pat >= (1 << 31) else pat
        print("pattern %08x  other kernels between: %s   v200 == pattern in %.1f %% of threads, a200 in %.1f %%"
              % (pat, between, 100.0 * (o[:, 0] == want).float().mean().item(), 100.0 * (o[:, 1] == want).float().mean().item()))
