#!/usr/bin/env python3
"""Headline benchmark: frames/sec of CogVideoX-5B-I2V 49-frame x 50-step ALG sampling (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one iteration of the ALG denoising loop (reference cog:1005-1140) of the C2 workload: schedule
strength -> low-pass filter of the conditioning latents -> 2- or 3-sample DiT forward -> fused CFG combine + DDIM
step.  Step s of the timed region is loop iteration s mod 50 of a video, so K = 50 (the default) is exactly one
whole video per GPU: 2 three-pass + 48 two-pass steps = 102 DiT sample-forwards.  frames/s = 49 * (K / 50) * N / T.

Inputs are synthetic and resident in HBM before the timed region: seeded random-init weights at the true
CogVideoX-5B-I2V shapes (5.55e9 parameters, bf16), seeded latents / conditioning latents / T5-shaped embeddings.
Ranks are independent replicas of the workload over different seeds (weak scaling); rank 0 broadcasts the weights
once over RCCL, there is no per-step collective.

The JSON line also carries
  roofline      -- the dominant kernel (flash attention): algorithmic FLOPs per launch / mean launch duration, timed
                   with HIP events on the launch stream inside the timed region; `extra` holds the same for the GEMMs
                   and the HBM GB/s of the filter and step kernels.
  cpu_baseline  -- the CPU oracle (oracle/dit_oracle.py, a port of the reference's PyTorch path; the reference
                   pipeline itself cannot run without diffusers) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(frames=49, steps=50, height=480, width=720, guidance_scale=6.0, lp_resize_factor=0.25,
          schedule_interval_end_time=0.04)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def event_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def pmc_traffic(kernel_substr, profile="profiles/r1_pmc_summary.txt"):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate passes and reported in KiB).  gfx950 correction from MI355X_MICROARCH.md section HBM: a wide
    coalesced 16-B/lane stream (our global_load_lds staging) is tallied at half its bytes in FETCH_SIZE -> x2.
    The PMC passes profile the kernel micro-benchmark (scripts/kbench.py) at the N = 2 C2 shape, i.e. a two-pass step."""
    path = os.path.join(ROOT, profile)
    if not os.path.exists(path):
        return None
    vals = {}
    with open(path) as f:
        for line in f:
            if kernel_substr in line:
                for name in ("FETCH_SIZE", "WRITE_SIZE"):
                    if (" " + name + " ") in line:
                        vals[name] = float(line.rsplit("mean=", 1)[1])
    if len(vals) != 2:
        return None
    return dict(bytes_per_launch=(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, fetch_kib_raw=vals["FETCH_SIZE"],
                write_kib_raw=vals["WRITE_SIZE"], samples_per_launch=2, source=profile)


def filter_microbench(dev):
    """HBM GB/s of the low-pass kernels at the BASELINE shapes: algorithmic bytes = 2 * planes * H * W *
    sizeof(dtype) per call.  One video is launch-bound (208 workgroups), so the 8-video batch is reported too."""
    from alg_amd import lp_utils

    out = {}
    g = torch.Generator().manual_seed(5)
    cases = {
        "down_up_c2_1video_bf16": (torch.randn(1, 13, 16, 60, 90, generator=g).to(torch.bfloat16), "down_up", 0.0, 0, 0.25),
        "down_up_c2_8videos_bf16": (torch.randn(8, 13, 16, 60, 90, generator=g).to(torch.bfloat16), "down_up", 0.0, 0, 0.25),
        "down_up_wan480p_f32": (torch.randn(1, 20, 21, 60, 104, generator=g), "down_up", 0.0, 0, 0.4),
        "down_up_c5_f32": (torch.randn(1, 20, 21, 90, 160, generator=g), "down_up", 0.0, 0, 0.4),
        "gaussian_wan480p_k9_f32": (torch.randn(1, 20, 21, 60, 104, generator=g), "gaussian_blur", 15.0, 9, 1.0),
        "gaussian_wan480p_8videos_f32": (torch.randn(8, 20, 21, 60, 104, generator=g), "gaussian_blur", 15.0, 9, 1.0),
    }
    for name, (x, kind, sigma, k, f) in cases.items():
        xd = x.to(dev)
        for _ in range(3):
            lp_utils.apply_low_pass_filter(xd, kind, sigma, k, f)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            lp_utils.apply_low_pass_filter(xd, kind, sigma, k, f)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        nbytes = 2.0 * x.numel() * x.element_size()
        out[name] = dict(ms=ms, mbytes=nbytes / 1e6, gbs=nbytes / (ms / 1e3) / 1e9, hbm_frac=nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS)
    return out


def cpu_filter_baseline():
    """the filters on the host cores through the CPU oracle (fp32; ATen's own op for down_up)."""
    from oracle import loop_oracle, lp_oracle
    import numpy as np

    g = torch.Generator().manual_seed(5)
    res = {}
    x = torch.randn(1, 16, 13, 60, 90, generator=g)
    loop_oracle.apply_low_pass_filter_torch(x, "down_up", 0.0, 0, 0.25)
    t0 = time.perf_counter()
    for _ in range(10):
        loop_oracle.apply_low_pass_filter_torch(x, "down_up", 0.0, 0, 0.25)
    res["down_up_c2_f32_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    w = torch.randn(1, 20, 21, 60, 104, generator=g).numpy()
    t0 = time.perf_counter()
    lp_oracle.gaussian_blur(w.astype(np.float32), 9, 15.0, np.float32)
    res["gaussian_wan480p_k9_f32_ms"] = (time.perf_counter() - t0) * 1e3
    return res


def cpu_baseline(budget_s=30.0):
    """One of the 42 DiT blocks of one sample-forward at the C2 token count, fp32, on the host cores, through the CPU
    oracle; scaled to frames/s of the whole workload (x 42 layers x 102 forwards per 49 frames)."""
    from oracle import dit_oracle

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    cfg = dit_oracle.DiTConfig(num_layers=1)
    full = dit_oracle.DiTConfig()
    w = dit_oracle.init_weights(cfg, seed=1, std=0.02)
    g = torch.Generator().manual_seed(0)
    tokens = 17776
    # sample: the block alone (patch embed / final projection are < 0.01 % of the FLOPs)
    hs = torch.randn(1, 13, 32, 60, 90, generator=g)
    ehs = torch.randn(1, 226, 4096, generator=g)
    rope = dit_oracle.rope_tables(full, 480, 720, 13)
    t0 = time.perf_counter()
    with torch.no_grad():
        dit_oracle.dit_forward(cfg, w, hs, ehs, torch.tensor([999]), rope)
    dt = time.perf_counter() - t0
    per_video = dt * full.num_layers * 102
    return dict(value=49.0 / per_video, unit="frames/s", cores=threads, kind="port",
                sample="one 1-layer CogVideoX-5B forward at the C2 token count (17,776 tokens, fp32 oracle, %.1f s) "
                       "scaled x42 layers x102 sample-forwards per 49-frame video" % dt,
                seconds_sampled=dt, flops_sampled=dit_oracle.flops_per_forward(cfg, tokens))


def vae_decode_microbench(dev, latents, loop_seconds_per_video):
    """The step after the loop (cog:427-433): CogVideoX VAE decode of the final latents to uint8 frames,
    outside the timed region and NOT part of `value` (BASELINE's metric counts the denoising loop); reported so the
    end-to-end figure is on record."""
    from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    vae = AutoencoderKLCogVideoX.from_synthetic(device=dev)
    z = (latents.float() * 0.3).to(torch.bfloat16).contiguous()
    frames = vae.decode_latents(z, to_uint8=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        vae.decode_latents(z, to_uint8=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = int(frames.shape[1])
    return {"ms_per_video": round(ms, 1), "frames": n, "conv_tflop": 312.98, "tflops": round(312.98 / ms * 1e3, 1),
            "frames_per_s_loop_plus_decode": round(n / (loop_seconds_per_video + ms / 1e3), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=42, help="debug only: fewer layers invalidates the metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg-split", action="store_true",
                    help="opt-in: cond/uncond CFG passes of one video on a pair of GPUs (latency mode, one all-reduce per step)")
    ap.add_argument("--filters-only", action="store_true", help="debug: only the low-pass kernel micro-benchmark")
    args = ap.parse_args()
    if args.filters_only:
        print(json.dumps(filter_microbench(torch.device("cuda:0"))))
        return

    import alg_amd
    from alg_amd import parallel, weights as W
    from alg_amd import CogVideoXDDIMScheduler, CogVideoXTransformer3DModel, CogVideoXTransformerConfig, lp_utils
    from alg_amd.pipeline_cogvideox_image2video_lowpass import get_resize_crop_region_for_grid, rotary_tables

    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ALG hot path is HIP-only")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = CogVideoXTransformerConfig(num_layers=args.layers)
    shapes = W.parameter_shapes(cfg)
    sd = parallel.broadcast_state_dict(lambda: W.synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev), shapes,
                                       dev)
    model = CogVideoXTransformer3DModel(cfg, sd, device=dev)
    del sd
    sched = CogVideoXDDIMScheduler()
    sched.set_timesteps(C2["steps"])
    timesteps = sched.timesteps

    # per-rank synthetic video inputs (seed 42 + rank, run.py:94 uses 42)
    split = None
    if args.cfg_split:
        # BASELINE config 5 style: the cond / uncond CFG passes of ONE video on a pair of GPUs (one small all-reduce per
        # step); pairs are independent videos.  Opt-in: the default bench is pure data parallelism over videos.
        if world % 2:
            raise SystemExit("--cfg-split needs an even number of GPUs")
        split = parallel.CFGPairSplit.from_world()
    g = torch.Generator().manual_seed(42 + (rank // 2 if split else rank))
    F_lat, C, Hh, Ww = 13, 16, 60, 90
    latents0 = torch.randn(1, F_lat, C, Hh, Ww, generator=g).to(dev, torch.bfloat16)
    image_latents = torch.zeros(1, F_lat, C, Hh, Ww, device=dev, dtype=torch.bfloat16)
    image_latents[:, 0] = (torch.randn(1, C, Hh, Ww, generator=g) * 0.7).to(dev, torch.bfloat16)
    pos = torch.randn(1, 226, 4096, generator=g).to(dev, torch.bfloat16)
    neg = torch.randn(1, 226, 4096, generator=g).to(dev, torch.bfloat16)
    emb2 = torch.cat([neg, pos]).contiguous()
    emb3 = torch.cat([neg, neg, pos]).contiguous()
    crops = get_resize_crop_region_for_grid((30, 45), 45, 30)
    rope = tuple(t.to(dev) for t in rotary_tables(64, crops, (30, 45), F_lat))
    latents = latents0.clone()

    kinds = {}  # kernel family -> list of event pairs

    def one_step(i, prof):
        """Loop iteration i (mod 50) of the C2 sampler on `latents` (in place)."""
        i = i % C2["steps"]
        t = timesteps[i]
        s = lp_utils.get_lp_strength(i, C2["steps"], "interval", 0.0, C2["schedule_interval_end_time"], 1.0, 0.0, 0.5,
                                     10.0)
        two_pass = s == 0
        factor = 1.0 - (1.0 - C2["lp_resize_factor"]) * s
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        lp = lp_utils.apply_low_pass_filter(image_latents, "down_up", 15.0 * s, 0.02734375, factor)
        if prof is not None and lp is not image_latents:
            e1.record()
            prof.setdefault("down_up", []).append((e0, e1))
        conds = [lp, lp] if two_pass else [image_latents, lp, lp]
        n = len(conds)
        ts = torch.full((n,), float(t), device=dev)
        if split is not None:
            rows = split.my_passes(n)
            emb = emb2 if two_pass else emb3
            local = model.forward_assembled(latents, [conds[r] for r in rows], emb[rows].contiguous(), ts[:len(rows)],
                                            rope)
            pred = split.merge(local, n, 1)
            n_local = len(rows)
        else:
            pred = model.forward_assembled(latents, conds, emb2 if two_pass else emb3, ts, rope)
            n_local = n
        if prof is not None:
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e2.record()
        sched.fused_cfg_step_(pred, latents, n, C2["guidance_scale"], t)
        if prof is not None:
            e3.record()
            prof.setdefault("cfg_step_%d" % n, []).append((e2, e3))
        return n_local

    for i in range(args.warmup):
        one_step(i, None)
    # steps 0/1 are the 3-pass steps: make sure both workspaces exist before timing even when warmup < 3
    if args.warmup < 3:
        one_step(0, None)
        one_step(2, None)
    latents.copy_(latents0)

    model.profile = kinds
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    forwards = 0
    for i in range(args.steps):
        forwards += one_step(i, kinds)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    model.profile = None
    elapsed = parallel.max_over_ranks(elapsed, dev)
    finite = bool(torch.isfinite(latents.float()).all().item())

    n_videos_parallel = world // 2 if split else world
    frames = C2["frames"] * args.steps / C2["steps"] * n_videos_parallel
    value = frames / elapsed

    # ---- per-kernel rooflines from the HIP events of the timed region --------------------------------------
    S, D, Hn, T = 17776, 3072, 48, 226
    ms = {k: event_ms(v) for k, v in kinds.items()}
    mean = {k: sum(v) / len(v) for k, v in ms.items() if v}
    n_of = {}  # samples per launch: attention / GEMM launches of 2- and 3-pass steps differ -> use per-launch totals
    total_samples = forwards
    attn_flops_total = 4.0 * S * S * 64 * Hn * total_samples * cfg.num_layers
    attn_time_total = sum(ms.get("attn", [])) / 1e3
    gemm_flops = dict(gemm_qk=2.0 * S * D * 2 * D, gemm_vt=2.0 * S * D * D, gemm_out=2.0 * S * D * D,
                      gemm_ff1=2.0 * S * D * 4 * D, gemm_ff2=2.0 * S * D * 4 * D)
    extra = {}
    for k, f in gemm_flops.items():
        tt = sum(ms.get(k, [])) / 1e3
        if tt > 0:
            extra[k + "_tflops"] = f * total_samples * cfg.num_layers / tt / 1e12
    gemm_time_total = sum(sum(ms.get(k, [])) for k in gemm_flops) / 1e3
    if gemm_time_total > 0:
        extra["gemm_all_tflops"] = sum(gemm_flops.values()) * total_samples * cfg.num_layers / gemm_time_total / 1e12
    numel = F_lat * C * Hh * Ww
    if "down_up" in mean:
        extra["down_up_gbs"] = 2.0 * numel * 2 / (mean["down_up"] / 1e3) / 1e9   # read + write, bf16
    for n in (2, 3):
        k = "cfg_step_%d" % n
        if k in mean:
            extra[k + "_gbs"] = (n + 2) * numel * 2 / (mean[k] / 1e3) / 1e9       # n bf16 preds + latents r/w
    for k in ("ln_mod", "qk_norm_rope"):
        tt = sum(ms.get(k, [])) / 1e3
        if tt > 0:
            per = (2.0 * S * D * 2) if k == "ln_mod" else (2.0 * S * 2 * D * 2)
            launches_per_layer = 2 if k == "ln_mod" else 1
            extra[k + "_gbs"] = per * total_samples * cfg.num_layers * launches_per_layer / tt / 1e9
    extra["time_share"] = {k: round(sum(v) / 1e3 / elapsed, 4) for k, v in ms.items()}
    attn_tflops = attn_flops_total / attn_time_total / 1e12 if attn_time_total > 0 else 0.0
    roofline = dict(bound="mfma", kernel="flash_attn_d64_kernel", achieved=attn_tflops, peak=MFMA_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=attn_tflops / MFMA_PEAK_TFLOPS, traffic=None,
                    launches=len(ms.get("attn", [])),
                    mean_launch_ms=(sum(ms["attn"]) / len(ms["attn"])) if ms.get("attn") else None, extra=extra)
    tr = pmc_traffic("flash_attn_d64_kernel")
    if tr is not None:  # `traffic` = HBM bytes per launch (PMC, corrected); how it was derived goes next to it
        roofline["traffic"] = tr["bytes_per_launch"]
        roofline["traffic_detail"] = tr
    flops_total = (attn_flops_total + sum(gemm_flops.values()) * total_samples * cfg.num_layers)
    # measured on this chip (profiles/r1_power_and_issue_rates.txt): a register-only MFMA loop on random bf16 operands
    # sustains 1765 TFLOP/s under the 1400 W package cap -- the matrix-pipe ceiling for real data; `peak` stays the guide's
    roofline["extra"]["mfma_sustained_random_operands_tflops"] = 1765.0
    roofline["extra"]["whole_step_mfma_frac"] = flops_total * world / elapsed / 1e12 / MFMA_PEAK_TFLOPS / world
    # tile-count quantisation of the persistent GEMM (256x256 tiles on `cus` workgroups): rounds the
    # last partial round costs as a whole one, for the 2-sample launches that make up 96 % of the steps
    cus = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
    tiles = lambda m, n, batch: batch * ((m + 255) // 256) * ((n + 255) // 256)
    tail = {}
    for name, (m, n, batch) in dict(gemm_qk=(S, 2 * D, 2), gemm_vt=(D, S, 2), gemm_out=(S, D, 2), gemm_ff1=(S, 4 * D, 2),
                                    gemm_ff2=(S, D, 2)).items():
        t = tiles(m, n, batch)
        tail[name] = {"tiles": t, "rounds_of_work": round(t / cus, 3), "rounds_paid": -(-t // cus)}
    roofline["extra"]["gemm_tile_rounds"] = tail

    out = {
        "metric": "frames/sec (whole node) CogVideoX-5B-I2V 49f x 50-step ALG",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (seeded random-init weights at CogVideoX-5B-I2V shapes, seeded latents/embeddings)",
        "config": {"workload": "BASELINE config 2: CogVideoX-5B-I2V bf16, 49 frames @ 480x720, 50 steps, ALG interval "
                               "down_up (resize_factor 0.25, interval [0, 0.04]), guidance 6.0; one video per GPU",
                   "layers": cfg.num_layers, "tokens": S, "dit_sample_forwards": forwards, "parallelism": ("cfgpair2xdp%d" % (world // 2)) if split else ("dp%d" % world),
                   "videos": args.steps / C2["steps"] * n_videos_parallel},
        "seconds": elapsed, "finite": finite, "roofline": roofline,
    }
    if cfg.num_layers != 42:
        out["INVALID"] = "debug run with %d layers" % cfg.num_layers
    if rank == 0 and world == 1:
        out["roofline"]["extra"]["filters"] = filter_microbench(dev)
        out["roofline"]["extra"]["vae_decode"] = vae_decode_microbench(dev, latents, elapsed / args.steps * C2["steps"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline"]["filters"] = cpu_filter_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
