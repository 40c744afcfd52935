#!/usr/bin/env python3
"""Full-size conditioning front-end on one MI355X, synthetic weights: T5 v1.1 XXL encoder on a
226-token prompt (cog:228-268), UMT5-XXL-width encoder on 512 tokens with a mask (wan:185-234), CLIP ViT-H/14 on one
224 x 224 image (wan:228-234), and the CogVideoX VAE encode of one 480 x 720 image (cog:388-391).

    python scripts/encoders_bench.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import (AutoencoderKLCogVideoX, CLIPVisionModel, T5EncoderConfig, T5EncoderModel,  # noqa: E402
                     UMT5EncoderModel)


def timed(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    res = {}
    t5 = T5EncoderModel.from_synthetic(device=dev)
    ids = torch.randint(0, 32128, (2, 226), generator=g).to(dev)            # prompt + negative prompt
    res["t5_xxl_2x226_tokens_ms"] = round(timed(lambda: t5(ids)), 2)
    del t5
    um = UMT5EncoderModel.from_synthetic(T5EncoderConfig(vocab_size=32128), device=dev)   # UMT5-XXL widths (vocab cut: table only)
    ids = torch.randint(0, 32128, (2, 512), generator=g).to(dev)
    mask = torch.ones(2, 512, dtype=torch.long, device=dev)
    mask[:, 80:] = 0
    res["umt5_xxl_2x512_tokens_ms"] = round(timed(lambda: um(ids, mask)), 2)
    del um
    clip = CLIPVisionModel.from_synthetic(device=dev)
    px = torch.randn(1, 3, 224, 224, generator=g).to(dev)
    res["clip_vit_h14_1_image_ms"] = round(timed(lambda: clip(pixel_values=px, output_hidden_states=True)), 2)
    del clip
    vae = AutoencoderKLCogVideoX.from_synthetic(device=dev, encoder=True)
    img = (torch.rand(1, 3, 1, 480, 720, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    res["cogvideox_vae_encode_480x720_ms"] = round(timed(lambda: vae.encode(img)), 2)
    res["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
