#!/usr/bin/env python3
"""Full-size HunyuanVideo-I2V DiT forward on one MI355X, synthetic weights.
Default shape = BASELINE config C4 per GPU: 129 frames @ 720x1280 -> 33 x 90 x 160 latents -> 118,800 latent tokens.

    python scripts/hy_bench.py [--n 1] [--frames 33] [--height 90] [--width 160] [--iters 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd.transformer_hunyuan_video import HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--frames", type=int, default=33)
    ap.add_argument("--height", type=int, default=90)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--text", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = HunyuanVideoTransformerConfig()
    t0 = time.time()
    model = HunyuanVideoTransformer3DModel.from_synthetic(cfg, device=dev)
    torch.cuda.synchronize()
    print("weights ready in %.1f s, %.1f GB allocated" % (time.time() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
    g = torch.Generator(device=dev).manual_seed(0)
    N, F, H, W, L = a.n, a.frames, a.height, a.width, a.text
    x = torch.randn(N, 16, F, H, W, generator=g, device=dev).to(torch.bfloat16)
    txt = torch.randn(N, L, 4096, generator=g, device=dev).to(torch.bfloat16)
    mask = torch.zeros(N, L, device=dev)
    mask[:, :64] = 1
    pooled = torch.randn(N, 768, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((N,), 996.0, device=dev)
    S = F * (H // 2) * (W // 2)
    J, D, M = S + L, cfg.dim, int(cfg.dim * cfg.mlp_ratio)
    out = model(x, t, txt, mask, pooled, return_dict=False)[0]
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        model(x, t, txt, mask, pooled, return_dict=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    lin_dual = 2.0 * N * J * (4 * D * D + 2 * D * M)
    lin_single = 2.0 * N * J * (3 * D * D + D * M + (D + M) * D)
    attn = 4.0 * N * J * (S + 64) * D
    flop = cfg.num_layers * (lin_dual + attn) + cfg.num_single_layers * (lin_single + attn)
    print(json.dumps({"ms_per_forward": round(ms, 1), "samples": N, "latent_tokens": S, "text_tokens": L,
                      "pflop_per_forward": round(flop / 1e15, 3), "tflops_whole_forward": round(flop / ms / 1e9, 1),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))


if __name__ == "__main__":
    main()
