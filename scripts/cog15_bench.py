#!/usr/bin/env python3
"""One full-size CogVideoX1.5-5B-I2V DiT forward on one MI355X (row a-6, `patch_size_t` variant), synthetic weights:
81 frames @ 768x1360 -> 22 padded latent frames x 96 x 170 -> 11 x 48 x 85 = 44,880 video tokens + 226 text tokens.

    python scripts/cog15_bench.py [--n 2] [--iters 2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd.pipeline_cogvideox_image2video_lowpass import rotary_tables  # noqa: E402
from alg_amd.transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = CogVideoXTransformerConfig(patch_size_t=2, ofs_embed_dim=512, use_learned_positional_embeddings=False,
                                     sample_height=96, sample_width=170, sample_frames=81)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    N, Fr, C, H, W = a.n, 22, 16, 96, 170
    lat = torch.randn(1, Fr, C, H, W, generator=g, device=dev).to(torch.bfloat16)
    conds = [torch.randn(1, Fr, C, H, W, generator=g, device=dev).to(torch.bfloat16) for _ in range(N)]
    emb = torch.randn(N, 226, 4096, generator=g, device=dev).to(torch.bfloat16)
    ts = torch.full((N,), 999.0, device=dev)
    cos, sin = rotary_tables(64, None, (48, 85), 11, max_size=(48, 85))
    rope = (cos.to(dev), sin.to(dev))
    ofs = torch.full((1,), 2.0, device=dev)
    out = model.forward_assembled(lat, conds, emb, ts, rope, ofs=ofs)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and out.shape == (N, Fr, C, H, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        model.forward_assembled(lat, conds, emb, ts, rope, ofs=ofs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    S, d = 11 * 48 * 85 + 226, 3072
    flop = N * 42 * (24.0 * S * d * d + 4.0 * S * S * d)
    print(json.dumps({"ms_per_forward": round(ms, 1), "samples": N, "tokens": S, "pflop_per_forward": round(flop / 1e15, 3),
                      "tflops_whole_forward": round(flop / ms / 1e9, 1),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))


if __name__ == "__main__":
    main()
