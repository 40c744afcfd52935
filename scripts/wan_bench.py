#!/usr/bin/env python3
"""Full-size Wan2.1-I2V-14B DiT forward on one MI355X (BASELINE config C3 shape: 81 frames @
480x832 -> 21 x 60 x 104 latents -> 32,760 tokens), synthetic weights.  Prints per-kernel-family time and TFLOP/s.

    python scripts/wan_bench.py [--n 2] [--layers 40] [--iters 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd.transformer_wan import WanTransformer3DModel, WanTransformerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--frames", type=int, default=21)
    ap.add_argument("--height", type=int, default=60)
    ap.add_argument("--width", type=int, default=104)
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5: e4m3 block linears on the fp8 MFMA")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = WanTransformerConfig(num_layers=a.layers)
    t0 = time.time()
    model = WanTransformer3DModel.from_synthetic(cfg, device=dev, fp8=a.fp8)
    torch.cuda.synchronize()
    print("weights ready in %.1f s, %.1f GB allocated" % (time.time() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
    g = torch.Generator(device=dev).manual_seed(0)
    N, F, H, W = a.n, a.frames, a.height, a.width
    x = torch.randn(N, 36, F, H, W, generator=g, device=dev).to(torch.bfloat16)
    txt = torch.randn(N, 512, 4096, generator=g, device=dev).to(torch.bfloat16)
    img = torch.randn(N, 257, 1280, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((N,), 999.0, device=dev)
    S = F * (H // 2) * (W // 2)
    D, Ff, L = cfg.dim, cfg.ffn_dim, cfg.num_layers
    out = model(x, t, txt, img, return_dict=False)[0]  # warm-up (allocates the workspace)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    model.profile = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        model(x, t, txt, img, return_dict=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    flops = {"gemm_qk": 2.0 * N * S * D * 2 * D, "gemm_vt": 2.0 * N * S * D * D, "gemm_out": 2.0 * N * S * D * D,
             "gemm_cq": 2.0 * N * S * D * D, "gemm_cout": 2.0 * N * S * D * D, "gemm_ff1": 2.0 * N * S * D * Ff,
             "gemm_ff2": 2.0 * N * S * D * Ff, "attn_self": 4.0 * N * S * S * D, "attn_cross": 4.0 * N * S * (512 + 257) * D / 2}
    total_flop = L * (sum(v for k, v in flops.items() if k != "attn_cross") + 2 * flops["attn_cross"])
    res = {"ms_per_forward": round(ms, 2), "fp8": a.fp8, "samples": N, "tokens": S, "layers": L,
           "tflops_whole_forward": round(total_flop / ms / 1e9, 1), "kernels": {}}
    for name, evs in sorted(model.profile.items()):
        tms = sum(x0.elapsed_time(x1) for x0, x1 in evs) / a.iters
        per = tms / (len(evs) / a.iters)
        entry = {"ms_total": round(tms, 2), "share": round(tms / ms, 4), "ms_per_launch": round(per, 4)}
        if name in flops:
            entry["tflops"] = round(flops[name] / per / 1e9, 1)
        elif name in ("ln_mod", "rms_rope"):
            entry["gbs"] = round(2.0 * N * S * D * 2 / per / 1e6, 1)
        res["kernels"][name] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
