#!/usr/bin/env python3
"""Run-to-run determinism of the C5 (Wan 14B fp8, 75,600 tokens) 2-block forward: N repeats against the first result.
python scripts/experiments/c5_determinism.py [repeats] [fp8 0|1] [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import WanTransformer3DModel, WanTransformerConfig  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fp8 = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
DEV, BF = "cuda:0", torch.bfloat16
if os.environ.get("SYNCPY"):   # debug: drain the device before (1) / after (2) every d = 128 attention call, from Python
    from alg_amd import _lib
    _orig = _lib.flash_attn_d128
    def _synced(*a, **k):
        if int(os.environ["SYNCPY"]) & 1:
            torch.cuda.synchronize()
        r = _orig(*a, **k)
        if int(os.environ["SYNCPY"]) & 2:
            torch.cuda.synchronize()
        return r
    _lib.flash_attn_d128 = _synced
F, H, W = 21, 90, 160
cfg = WanTransformerConfig(num_layers=layers)
model = WanTransformer3DModel.from_synthetic(cfg, seed=21, device=DEV, fp8=fp8)
g = torch.Generator(device=DEV).manual_seed(5)
x = torch.randn(3, 36, F, H, W, generator=g, device=DEV).to(BF)
txt = torch.randn(3, 512, 4096, generator=g, device=DEV).to(BF)
img = torch.randn(3, 257, 1280, generator=g, device=DEV).to(BF)
ts = torch.full((3,), 900.0, device=DEV)
run = lambda: model(hidden_states=x, timestep=ts, encoder_hidden_states=txt, encoder_hidden_states_image=img, return_dict=False)[0]
ref = run()
bad = []
for i in range(reps):
    y = run()
    if not torch.equal(y, ref):
        d = (y.float() - ref.float()).abs()
        nz = (d > 0).nonzero()
        info = {"per_sample": [int((d[k] > 0).sum()) for k in range(d.shape[0])]}
        for ax, nm in ((2, "f"), (3, "h"), (4, "w")):
            info[nm] = (int(nz[:, ax].min()), int(nz[:, ax].max()), int(nz[:, ax].unique().numel()))
        bad.append((i, int((d > 0).sum()), float(d.max()), info))
print("env", {k: v for k, v in os.environ.items() if k.startswith("ALG_")}, "fp8", fp8, "repeats", reps, "mismatching runs", bad)
