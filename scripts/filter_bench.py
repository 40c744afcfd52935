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
from alg_amd import _lib  # noqa: E402
for tag, path in (("v3", "0"), ("v2", "2"), ("v1", "1")):     # ALG_LOWPASS_PATH: 0 auto (v3 from 128 planes), 2 = v2, 1 = v1
    os.environ["ALG_LOWPASS_PATH"] = path
    _lib.reload_env()                                          # the library reads its options once, at load
    res[tag] = {k: {"us": round(v["ms"] * 1e3, 2), "gbs": round(v["gbs"]), "hbm_frac": round(v["hbm_frac"], 3)}
                for k, v in bench.filter_microbench(dev).items()}
print(json.dumps(res, indent=1))
