#!/usr/bin/env python3
"""Wan fp8 (C5) forward: poison the model's persistent workspace between two forwards; the second result must not change.
python c5_ws_poison.py [fp8 0|1] [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import WanTransformer3DModel, WanTransformerConfig  # noqa: E402

fp8 = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DEV, BF = "cuda:0", torch.bfloat16
F, H, W = 21, 90, 160
cfg = WanTransformerConfig(num_layers=layers)
model = WanTransformer3DModel.from_synthetic(cfg, seed=21, device=DEV, fp8=fp8)
g = torch.Generator(device=DEV).manual_seed(5)
x = torch.randn(3, 36, F, H, W, generator=g, device=DEV).to(BF)
txt = torch.randn(3, 512, 4096, generator=g, device=DEV).to(BF)
img = torch.randn(3, 257, 1280, generator=g, device=DEV).to(BF)
ts = torch.full((3,), 900.0, device=DEV)
run = lambda: model(hidden_states=x, timestep=ts, encoder_hidden_states=txt, encoder_hidden_states_image=img, return_dict=False)[0]
ref = run().clone()
ws = next(iter(model._ws.values()))
S = 75600
names = [k for k, v in vars(ws).items() if torch.is_tensor(v)]
print("workspace tensors:", names)
for name in names + ["ALL"]:
    for k, v in vars(ws).items():
        if not torch.is_tensor(v) or (name != "ALL" and k != name):
            continue
        if k in ("vt", "vti", "vtt"):
            continue
        if k == "vt":
            v[:, :, :S].fill_(float("nan"))       # the padding columns are zero by contract
        elif v.dtype == torch.uint8:
            v.fill_(0x7f)
        elif v.dtype.is_floating_point:
            v.fill_(float("nan"))
    y = run()
    d = (y.float() - ref.float()).abs()
    nz = (d > 0) | torch.isnan(d)
    print("poisoned %-10s: %d elements differ, nan %d, first %s" % (name, int(nz.sum()), int(torch.isnan(y.float()).sum()), nz.nonzero()[:3].tolist()))
