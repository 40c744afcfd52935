#!/usr/bin/env python3
"""Probe: do the two CFG samples of a CogVideoX forward run faster as ONE batched forward (N = 2, what the sampler does) or
as TWO single-sample forwards on two HIP streams?  The second form lets one stream's kernels fill the idle compute units
of the other's partial last rounds (GEMM: 6.5 rounds paid as 7; attention likewise), at the price of two kernels sharing
the L2s.  C2 shape (17,776 tokens), synthetic weights, `--layers` blocks.

    python scripts/two_stream_probe.py [--layers 8] [--iters 3]
"""
import argparse
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd.pipeline_cogvideox_image2video_lowpass import get_resize_crop_region_for_grid, rotary_tables  # noqa: E402
from alg_amd.transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = CogVideoXTransformer3DModel.from_synthetic(CogVideoXTransformerConfig(num_layers=a.layers), device=dev)
    twin = copy.copy(model)      # same weights, its own workspace
    twin._ws = {}
    g = torch.Generator(device=dev).manual_seed(0)
    Fr, C, H, W = 13, 16, 60, 90
    lat = torch.randn(1, Fr, C, H, W, generator=g, device=dev).to(torch.bfloat16)
    conds = [torch.randn(1, Fr, C, H, W, generator=g, device=dev).to(torch.bfloat16) for _ in range(2)]
    emb = torch.randn(2, 226, 4096, generator=g, device=dev).to(torch.bfloat16)
    ts = torch.full((2,), 999.0, device=dev)
    rope = tuple(t.to(dev) for t in rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), Fr))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]

    def batched():
        return model.forward_assembled(lat, conds, emb, ts, rope)

    def split():
        outs = []
        ev = torch.cuda.Event()
        ev.record()
        for k, m in enumerate((model, twin)):
            s[k].wait_event(ev)
            with torch.cuda.stream(s[k]):
                outs.append(m.forward_assembled(lat, conds[k:k + 1], emb[k:k + 1], ts[k:k + 1], rope))
        for k in range(2):
            torch.cuda.current_stream().wait_stream(s[k])
        return torch.cat(outs)

    ref = batched()
    got = split()
    torch.cuda.synchronize()
    res = {"layers": a.layers, "max_abs_diff": (ref.float() - got.float()).abs().max().item()}
    for name, fn in (("batched_n2", batched), ("two_streams_n1", split), ("batched_n2_again", batched),
                     ("two_streams_n1_again", split)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = round(e0.elapsed_time(e1) / a.iters, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
