#!/usr/bin/env python3
"""HBM GB/s of the low-pass kernels (both generations) at the BASELINE shapes: python scripts/filter_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
res = {}
for tag, env, env3 in (("v3", "0", "1"), ("v2", "0", "0"), ("v1", "1", "1")):
    os.environ["ALG_LOWPASS_V1"] = env
    os.environ["ALG_LOWPASS_V3"] = env3
    res[tag] = {k: {"us": round(v["ms"] * 1e3, 2), "gbs": round(v["gbs"]), "hbm_frac": round(v["hbm_frac"], 3)}
                for k, v in bench.filter_microbench(dev).items()}
print(json.dumps(res, indent=1))
