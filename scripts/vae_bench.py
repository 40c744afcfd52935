#!/usr/bin/env python3
"""Full-size CogVideoX VAE decode on one MI355X, synthetic weights.
Default shape = BASELINE config C2: 13 latent frames @ 60x90 -> 49 frames @ 480x720.

    python scripts/vae_bench.py [--frames 13] [--height 60] [--width 90] [--iters 3] [--profile]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX, _Level  # noqa: E402


def conv_flops(vae, L, h, w):
    """useful multiply-adds x 2 of every convolution / 1x1x1 projection of one decode (valid voxels only)."""
    c = vae.config
    rev = list(reversed(c.block_out_channels))
    lv = _Level(L, h, w, 1, 1)
    vox = lambda v: v.T * v.H * v.W
    fl = 2.0 * vox(lv) * 27 * c.latent_channels * rev[0]
    rs = vae._resnets()
    per_level = [rs[:2 + c.layers_per_block + 1]] + [rs[2 + (c.layers_per_block + 1) * i:2 + (c.layers_per_block + 1) * (i + 1)]
                                                      for i in range(1, len(rev))]
    rate, scale = 1, 1
    for i, group in enumerate(per_level):
        lv = _Level(L, h, w, rate, scale)
        for _, ci, co in group:
            fl += 2.0 * vox(lv) * (27 * ci * co + 27 * co * co + (ci * co if ci != co else 0))
        if i != len(rev) - 1:
            rate, scale = rate * (2 if i < 2 else 1), scale * 2
            fl += 2.0 * vox(_Level(L, h, w, rate, scale)) * 9 * rev[i] * rev[i]
    fl += 2.0 * vox(lv) * 27 * rev[-1] * c.out_channels
    return fl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=13)
    ap.add_argument("--height", type=int, default=60)
    ap.add_argument("--width", type=int, default=90)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    vae = AutoencoderKLCogVideoX.from_synthetic(device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    lat = torch.randn(1, a.frames, 16, a.height, a.width, generator=g, device=dev).to(torch.bfloat16)
    out = vae.decode_latents(lat, to_uint8=True)
    torch.cuda.synchronize()
    fr = vae.decode_latents(lat)
    assert torch.isfinite(fr.float()).all()
    print("output", tuple(out.shape), "frames std %.3f" % fr.float().std().item(), flush=True)
    del fr
    torch.cuda.reset_peak_memory_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        vae.decode_latents(lat, to_uint8=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = conv_flops(vae, a.frames, a.height, a.width)
    res = {"ms_per_decode": round(ms, 2), "latent": [a.frames, a.height, a.width], "frames": int(out.shape[1]),
           "conv_tflop": round(fl / 1e12, 2), "tflops_whole_decode": round(fl / ms / 1e9, 1),
           "frames_per_s_decode_only": round(out.shape[1] / ms * 1e3, 1),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    if a.profile:
        vae.profile = {}
        vae.decode_latents(lat, to_uint8=True)
        torch.cuda.synchronize()
        vae2, vae.profile = vae.profile, None
        flat = [(n, e) for n, lst in vae2.items() for e in lst]
        t0 = vae2["pack"][0]
        flat.sort(key=lambda p: t0.elapsed_time(p[1]))
        tot = {}
        for (n, e), (_, e2) in zip(flat[:-1], flat[1:]):
            tot[n] = tot.get(n, 0.0) + e.elapsed_time(e2)
        res["ms_by_op"] = {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
