"""Full-size timing of the HunyuanVideo VAE (published widths, synthetic weights): encode of the conditioning image
(hy:578-582) and the temporally tiled decode of the final latents (hy:1291-1292) at the bucket sizes run.py produces."""
import sys
import time

import torch

from alg_amd import AutoencoderKLHunyuanVideo

BF = torch.bfloat16
dev = "cuda:0"
vae = AutoencoderKLHunyuanVideo.from_synthetic(device=dev)
cases = [(544, 960, 129), (720, 1280, 129)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]
for H, W, F in cases:
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 1, H, W, generator=g).clamp(-1, 1).to(BF).to(dev)
    z = (torch.randn(1, 16, (F - 1) // 4 + 1, H // 8, W // 8, generator=g) * 0.7).to(BF).to(dev)
    for what, fn in (("encode", lambda: vae.encode(img).latent_dist.mode()), ("decode", lambda: vae.decode(z).sample)):
        torch.cuda.reset_peak_memory_stats()
        out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(what, tuple(out.shape), "%.3f s" % dt, "peak GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9),
              bool(torch.isfinite(out.float()).all()), flush=True)
        del out
