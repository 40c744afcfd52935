#!/usr/bin/env python3
"""Does the Wan fp8 (C5) forward read memory it never wrote?  The caching allocator's free blocks are filled with a byte
pattern before each forward; outputs must not depend on the pattern.  python c5_uninit.py [fp8 0|1] [tokens-frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import WanTransformer3DModel, WanTransformerConfig  # noqa: E402

fp8 = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
DEV, BF = "cuda:0", torch.bfloat16
F, H, W = 21, 90, 160
cfg = WanTransformerConfig(num_layers=1)
model = WanTransformer3DModel.from_synthetic(cfg, seed=21, device=DEV, fp8=fp8)
g = torch.Generator(device=DEV).manual_seed(5)
x = torch.randn(3, 36, F, H, W, generator=g, device=DEV).to(BF)
txt = torch.randn(3, 512, 4096, generator=g, device=DEV).to(BF)
img = torch.randn(3, 257, 1280, generator=g, device=DEV).to(BF)
ts = torch.full((3,), 900.0, device=DEV)
run = lambda: model(hidden_states=x, timestep=ts, encoder_hidden_states=txt, encoder_hidden_states_image=img, return_dict=False)[0]


def poison(byte):
    """fill every free cached block (and some fresh memory) with `byte`"""
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    bufs = []
    try:
        # grab what the cache holds plus fresh memory, in chunks of decreasing size
        for sz in (8 << 30, 2 << 30, 512 << 20, 64 << 20, 8 << 20, 1 << 20):
            for _ in range(64):
                try:
                    if torch.cuda.memory_reserved() + sz > 150 * (1 << 30):
                        break
                    bufs.append(torch.empty(sz, dtype=torch.uint8, device=DEV))
                except RuntimeError:
                    break
        for b in bufs:
            b.fill_(byte)
    finally:
        del bufs
    torch.cuda.synchronize()


ref = run().clone()
for byte in (0x00, 0x7f, 0xff, 0x00):
    poison(byte)
    y = run()
    d = (y.float() - ref.float()).abs()
    nz = (d > 0) | torch.isnan(d)
    idx = nz.nonzero()
    print("pattern 0x%02x: %d elements differ, max %.4g, nan %d; first indices %s" % (
        byte, int(nz.sum()), float(torch.nan_to_num(d).max()), int(torch.isnan(y.float()).sum()), idx[:6].tolist()))
