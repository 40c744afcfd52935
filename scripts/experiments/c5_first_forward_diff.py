#!/usr/bin/env python3
"""Which workspace tensors differ between the first and the second forward of a fresh process (Wan fp8, C5 tokens)?
python c5_first_forward_diff.py [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import WanTransformer3DModel, WanTransformerConfig  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
DEV, BF = "cuda:0", torch.bfloat16
F, H, W = 21, 90, 160
cfg = WanTransformerConfig(num_layers=layers)
model = WanTransformer3DModel.from_synthetic(cfg, seed=21, device=DEV, fp8=True)
g = torch.Generator(device=DEV).manual_seed(5)
x = torch.randn(3, 36, F, H, W, generator=g, device=DEV).to(BF)
txt = torch.randn(3, 512, 4096, generator=g, device=DEV).to(BF)
img = torch.randn(3, 257, 1280, generator=g, device=DEV).to(BF)
ts = torch.full((3,), 900.0, device=DEV)
run = lambda: model(hidden_states=x, timestep=ts, encoder_hidden_states=txt, encoder_hidden_states_image=img, return_dict=False)[0]
y1 = run().clone()
ws = next(iter(model._ws.values()))
snap = {k: v.clone() for k, v in vars(ws).items() if torch.is_tensor(v)}
y2 = run()
diff = []
for k, v in vars(ws).items():
    if torch.is_tensor(v):
        a, b = snap[k].view(torch.uint8) if snap[k].dtype != torch.uint8 else snap[k], v.view(torch.uint8) if v.dtype != torch.uint8 else v
        n = int((a != b).sum())
        if n:
            nz = (snap[k] != v).nonzero()
            diff.append((k, n, tuple(v.shape), nz[0].tolist(), nz[-1].tolist()))
print("output differs:", int((y1 != y2).sum()), "| workspace tensors that differ:", diff)
