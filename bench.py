#!/usr/bin/env python3
"""Headline benchmark: frames/sec of CogVideoX-5B-I2V 49-frame x 50-step ALG sampling (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5] [--cfg-split]

Invoked plainly with N > 1 it launches its own ranks (re-exec through torch.distributed.run, one process per GPU,
rendezvous on 127.0.0.1); invoked BY torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

A "step" is one iteration of the ALG denoising loop (reference cog:1005-1140) of the workload, executed THROUGH the
drop-in pipeline's ``__call__`` (``CogVideoXImageToVideoPipeline.__call__`` for c2): schedule strength -> low-pass filter
of the conditioning latents -> 2- or 3-sample DiT forward -> fused CFG combine + scheduler step -> callback plumbing.  The
timed region is a sequence of ``__call__``s covering exactly K steps (a call is cut short after its share by the
reference's own ``interrupt`` mechanism, set from ``callback_on_step_end``); step s of the region is loop iteration
s mod steps_per_video, so K = 50 (the default) is exactly one whole C2 video per GPU: 2 three-pass + 48 two-pass steps
= 102 DiT sample-forwards.  frames/s = frames_per_video * (K / steps_per_video) * videos_in_flight / T.

Inputs are synthetic and resident in HBM before the timed region: seeded random-init weights at the true model
shapes (bf16), seeded latents / conditioning latents / encoder-shaped embeddings.  Ranks are independent replicas of
the workload over different seeds (weak scaling); rank 0 materialises the weights and broadcasts them once over RCCL,
there is no per-step collective (``--cfg-split``: the cond / uncond passes of one video on a GPU pair, one all-gather
per step).

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
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X (MI355X_MICROARCH.md)
FP8_PEAK_TFLOPS = 5000.0   # dense fp8 through the scaled (MX) MFMA
HBM_PEAK_GBS = 8000.0


def event_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def pmc_traffic(kernel_substr, profiles=("profiles/r6_pmc_bench_run.txt", "profiles/r5_pmc_summary.txt", "profiles/r4_pmc_summary.txt", "profiles/r3_pmc_summary.txt", "profiles/r2_pmc_summary.txt", "profiles/r1_pmc_summary.txt"), samples=2):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate passes and reported in KiB).  gfx950 correction from MI355X_MICROARCH.md section HBM: a wide
    coalesced 16-B/lane stream (our global_load_lds staging) is tallied at half its bytes in FETCH_SIZE -> x2.
    Round 6: the first profile is a PMC pass over the BENCH command itself (scripts/gpu_r6_pmc.sh), summarised per (kernel, grid):
    the `samples`-sample launch of the step is picked by its grid (a 3-sample C2 launch has 1.5 x the workgroups of a 2-sample
    one); the older profiles are kernel micro-benchmarks (scripts/kbench.py) at the N = 2 shape."""
    for profile in profiles:
        path = os.path.join(ROOT, profile)
        if not os.path.exists(path):
            continue
        per_kernel = {}   # kernel name as printed (+ grid) -> {counter: mean}
        with open(path) as f:
            for line in f:
                if kernel_substr in line and "mean=" in line:
                    for name in ("FETCH_SIZE", "WRITE_SIZE"):
                        if (" " + name + " ") in line:
                            kname = line.split(name)[0].strip()
                            per_kernel.setdefault(kname, {})[name] = float(line.rsplit("mean=", 1)[1])
        full = {k: v for k, v in per_kernel.items() if len(v) == 2}
        if not full:
            continue
        by_grid = {}
        for k, v in full.items():
            if "[grid=" in k:
                try:
                    by_grid[int(k.split("[grid=")[1].rstrip("]"))] = v
                except ValueError:
                    pass
        if by_grid:
            # the main launches of a bench run: the largest grids; 3-sample = the largest, 2-sample = the next distinct size
            grids = sorted(by_grid, reverse=True)
            main = [g for g in grids if g >= 0.5 * grids[0]]
            pick = main[0] if (samples == 3 or len(main) == 1) else main[1]
            vals, n_samples = by_grid[pick], (3 if pick == main[0] and len(main) > 1 else samples)
        else:
            vals, n_samples = max(full.values(), key=lambda v: v["FETCH_SIZE"]), 2
        return dict(bytes_per_launch=(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                    fetch_kib_raw=vals["FETCH_SIZE"], write_kib_raw=vals["WRITE_SIZE"], samples_per_launch=n_samples,
                    source=profile)
    return None


class SmiSampler:
    """Package power / shader clock of GPU `index` sampled by a background thread (librocm_smi64 through ctypes: no process
    spawn, ~50 us per sample; falls back to parsing `rocm-smi` output once per second, and to nothing).  `with SmiSampler(i) as
    s: ...; s.summary()` -> {"power_w": {mean, max, n}, "sclk_mhz": {...}, "power_cap_w": ..., "source": ...} or None."""

    def __init__(self, index=0, period=0.2):
        import threading
        self.index, self.period = index, period
        self.samples, self.cap, self.source = [], None, None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._read = self._open_rsmi() or self._open_cli()

    def _open_rsmi(self):
        import ctypes as C
        for path in ("/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so", "librocm_smi64.so.1"):
            try:
                lib = C.CDLL(path)
                break
            except OSError:
                lib = None
        if lib is None:
            return None
        try:
            if lib.rsmi_init(C.c_uint64(0)) != 0:
                return None

            class Freq(C.Structure):
                _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                            ("frequency", C.c_uint64 * 33)]

            idx = C.c_uint32(self.index)
            cap = C.c_uint64(0)
            if lib.rsmi_dev_power_cap_get(idx, C.c_uint32(0), C.byref(cap)) == 0 and cap.value:
                self.cap = cap.value / 1e6

            def read():
                pw, kind, f = C.c_uint64(0), C.c_int(0), Freq()
                watts = mhz = None
                if lib.rsmi_dev_power_get(idx, C.byref(pw), C.byref(kind)) == 0 and pw.value:
                    watts = pw.value / 1e6
                elif lib.rsmi_dev_current_socket_power_get(idx, C.byref(pw)) == 0 and pw.value:
                    watts = pw.value / 1e6
                if lib.rsmi_dev_gpu_clk_freq_get(idx, C.c_int(0), C.byref(f)) == 0 and f.current < 33:
                    mhz = f.frequency[f.current] / 1e6
                return watts, mhz

            w, m = read()
            if w is None and m is None:
                return None
            self.source = "librocm_smi64 (rsmi_dev_power_get / rsmi_dev_gpu_clk_freq_get SYS)"
            return read
        except Exception:
            return None

    def _open_cli(self):
        import re
        import shutil
        exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        if exe is None:
            return None

        def read():
            try:
                out = subprocess.run([exe, "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=10).stdout
            except Exception:
                return None, None
            w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
            return (float(w.group(1)) if w else None), (float(m.group(1)) if m else None)

        self.period = max(self.period, 1.0)
        self.source = "rocm-smi --showpower --showclocks"
        return read

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self._read is not None:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=15)
        return False

    def summary(self):
        if self._read is None:
            return None
        stat = lambda v: {"mean": round(sum(v) / len(v), 1), "max": round(max(v), 1), "min": round(min(v), 1), "n": len(v)} if v else None
        return {"power_w": stat([w for w, _ in self.samples if w is not None]),
                "sclk_mhz": stat([m for _, m in self.samples if m is not None]), "power_cap_w": self.cap, "source": self.source}


def calibrate_box(dev, seconds=3.0):
    """what THIS chip's matrix pipe sustains right now -- a register-only v_mfma_f32_32x32x16_bf16 loop on
    pseudo-random operands (csrc/calibrate.hip; two waves per SIMD, no memory traffic) run for `seconds` -- with the shader clock
    it ran at (cycle counter / constant-rate counter, read by every workgroup) and the package power meanwhile.  The number
    replaces round 4's hard-coded 1765 TFLOP/s; it is the matrix-pipe ceiling under the package power cap on real data, i.e. the
    practical denominator next to the guide's 2500."""
    from alg_amd import _lib
    iters = 4000
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks = 2 * cus
    clocks = torch.zeros(blocks, 4, dtype=torch.int64, device=dev)
    flop = 2.0 * 32 * 32 * 16 * 32 * iters * 4 * blocks
    _lib.calib_mfma_bf16(sink, iters, 1, blocks, clocks)
    torch.cuda.synchronize()
    rates, t0 = [], time.perf_counter()
    with SmiSampler(dev.index or 0) as smi:
        while time.perf_counter() - t0 < seconds:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(4):
                _lib.calib_mfma_bf16(sink, iters, 2 + i, blocks, clocks)
            b.record()
            b.synchronize()
            rates.append(4 * flop / (a.elapsed_time(b) / 1e3) / 1e12)
    tail = rates[2 * len(rates) // 3:] or rates      # the last third: the package has reached its power limit by then
    khz = _lib.wall_clock_khz()
    out = {"mfma_sustained_random_operands_tflops": sum(tail) / len(tail), "first_tflops": rates[0], "launch_groups": len(rates),
           "seconds": round(time.perf_counter() - t0, 2), "frac_of_peak": sum(tail) / len(tail) / MFMA_PEAK_TFLOPS,
           "shader_clock_mhz": _lib.clock_mhz_from_taps(clocks, khz), "wall_clock_khz": khz, "smi": smi.summary(),
           "what": "register-only v_mfma_f32_32x32x16_bf16 loop, pseudo-random bf16 operands, 2 waves per SIMD on %d CUs" % cus}
    clk = out["shader_clock_mhz"]
    if clk:   # 32 cycles per MFMA per SIMD: what the pipe would deliver at that clock if it never idled
        out["pipe_busy_at_that_clock"] = out["mfma_sustained_random_operands_tflops"] / (cus * 4 * 32768.0 / 32.0 * clk["mean"] * 1e6 / 1e12)
    return out


def filter_microbench(dev):
    """HBM GB/s of the low-pass kernels at the BASELINE shapes: algorithmic bytes = 2 * planes * H * W *
    sizeof(dtype) per call.  One video is launch-bound (208 planes), so the 8-video batch is reported too."""
    from alg_amd import lp_utils

    out = {}
    g = torch.Generator().manual_seed(5)
    cases = {
        "down_up_c2_1video_bf16": (torch.randn(1, 13, 16, 60, 90, generator=g).to(torch.bfloat16), "down_up", 0.0, 0, 0.25),
        "down_up_c2_8videos_bf16": (torch.randn(8, 13, 16, 60, 90, generator=g).to(torch.bfloat16), "down_up", 0.0, 0, 0.25),
        "down_up_wan480p_f32": (torch.randn(1, 20, 21, 60, 104, generator=g), "down_up", 0.0, 0, 0.4),
        "down_up_c5_f32": (torch.randn(1, 20, 21, 90, 160, generator=g), "down_up", 0.0, 0, 0.4),
        "down_up_c5_8videos_f32": (torch.randn(8, 20, 21, 90, 160, generator=g), "down_up", 0.0, 0, 0.4),
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
        pmc = FILTER_PMC_BYTES.get(name)
        if pmc is not None:   # rocprofv3 FETCH_SIZE (x 2, gfx950) + WRITE_SIZE of the same launch shape, committed under profiles/
            out[name].update(pmc_mbytes=pmc / 1e6, pmc_over_algorithmic=pmc / nbytes, pmc_gbs=pmc / (ms / 1e3) / 1e9,
                             pmc_source="profiles/r6_pmc_filters_hbm.txt")
    return out


# HBM-side bytes per launch of the three batched filter shapes from the committed --pmc passes (profiles/r6_pmc_filters_hbm.txt, re-taken in round 6:
# 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes): they equal the algorithmic bytes to 0.1-1.1 % -- nothing is re-read.
FILTER_PMC_BYTES = {
    "down_up_c5_8videos_f32": (2 * 9.462e4 + 1.89e5) * 1024.0,
    "gaussian_wan480p_8videos_f32": (2 * 4.101e4 + 8.19e4) * 1024.0,
    "down_up_c2_8videos_bf16": (2 * 8976.0 + 1.758e4) * 1024.0,
}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle -- a port of the reference's PyTorch path -- on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def _thread_candidates():
    n = os.cpu_count() or 1
    return sorted({t for t in (4, 8, 16, 32, 64, n) if t <= n})


def cpu_filter_baseline():
    """the filters on the host cores through the CPU oracle (fp32; ATen's own op for down_up).
    These tensors are 2-10 MB: oversubscribing a 256-core host makes ATen 100x slower than 8 threads, so the leg sweeps
    the thread count and reports the best one."""
    from oracle import loop_oracle, lp_oracle
    import numpy as np

    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 13, 60, 90, generator=g)
    w = torch.randn(1, 20, 21, 60, 104, generator=g).numpy().astype(np.float32)
    res = {}
    best = None
    for th in _thread_candidates():
        torch.set_num_threads(th)
        loop_oracle.apply_low_pass_filter_torch(x, "down_up", 0.0, 0, 0.25)
        t0 = time.perf_counter()
        for _ in range(5):
            loop_oracle.apply_low_pass_filter_torch(x, "down_up", 0.0, 0, 0.25)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if best is None or ms < best[0]:
            best = (ms, th)
    res["down_up_c2_f32_ms"], res["down_up_c2_threads"] = best
    # gaussian: what lp:47 executes -- torchvision's reflect pad + ONE depthwise F.conv2d (ATen's threaded kernel), best of the
    # same thread sweep; the single-threaded numpy statement of the oracle stays next to it for reference
    wt = torch.from_numpy(w)
    best = None
    for th in _thread_candidates():
        torch.set_num_threads(th)
        loop_oracle.gaussian_blur_torch(wt, 9, 15.0)
        t0 = time.perf_counter()
        for _ in range(3):
            loop_oracle.gaussian_blur_torch(wt, 9, 15.0)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        if best is None or ms < best[0]:
            best = (ms, th)
    res["gaussian_wan480p_k9_f32_ms"], res["gaussian_wan480p_threads"] = best
    t0 = time.perf_counter()
    lp_oracle.gaussian_blur(w, 9, 15.0, np.float32)
    res["gaussian_wan480p_k9_f32_numpy_1thread_ms"] = (time.perf_counter() - t0) * 1e3
    return res


def cpu_c1_call(threads, budget_s=150.0):
    """BASELINE.md CPU row 2 / BASELINE config 1: the whole ALG sampler __call__ at CogVideoX-5B widths, fp32, 9 frames @
    256x256 (994 tokens), 2 steps = 1 three-pass + 1 two-pass = 5 sample-forwards, through the CPU oracle
    (loop_oracle.alg_denoise_loop driving dit_oracle.dit_forward).  The 42 blocks SHARE one block's seeded weights (the
    timing is that of 42 distinct blocks up to cache effects in the CPU's favour; 22 GB of fp32 weights are not
    materialised on the host).  A 2-layer probe comes first: when it projects the whole call beyond `budget_s` the leg
    reports the projection instead of running (the default bench has to finish within minutes; --c1-budget lifts it)."""
    from oracle import ddim_oracle, dit_oracle, loop_oracle

    torch.set_num_threads(threads)
    one = dit_oracle.DiTConfig(num_layers=1, sample_height=32, sample_width=32, sample_frames=9)
    w1 = dit_oracle.init_weights(one, seed=1, std=0.02)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 3, 16, 32, 32, generator=g)
    cond = torch.zeros(1, 3, 16, 32, 32)
    cond[:, :1] = torch.randn(1, 1, 16, 32, 32, generator=g) * 0.7
    pe, ne = torch.randn(1, 226, 4096, generator=g), torch.randn(1, 226, 4096, generator=g)
    args = dict(num_inference_steps=2, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
                lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_end_time=0.04)
    what = ("BASELINE config 1: CogVideoX-5B widths fp32, 9 frames @ 256x256 (994 tokens), 2 steps, ALG down_up in "
            "latent, whole sampler through the CPU oracle")

    def call(n_layers):
        cfg = dit_oracle.DiTConfig(num_layers=n_layers, sample_height=32, sample_width=32, sample_frames=9)
        w = dict(w1)
        for k, v in w1.items():
            if k.startswith("transformer_blocks.0."):
                for i in range(1, n_layers):
                    w["transformer_blocks.%d.%s" % (i, k[len("transformer_blocks.0."):])] = v
        rope = dit_oracle.rope_tables(cfg, 256, 256, 3)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = loop_oracle.alg_denoise_loop(lambda a, b, c, d: dit_oracle.dit_forward(cfg, w, a, b, c, d),
                                               ddim_oracle.DDIMOracle(), lat, cond, pe, ne, image_rotary_emb=rope, **args)
        return time.perf_counter() - t0, out, cfg

    call(1)                                  # page in the weights / spin up the thread pool
    t2, _, _ = call(2)
    estimate = t2 / 2 * 42
    if estimate > budget_s:
        return dict(skipped="2-layer probe %.1f s projects the 42-layer call to %.0f s > budget %.0f s" % (t2, estimate, budget_s),
                    projected_seconds=estimate, projected_frames_per_s=9.0 / estimate, threads=threads, what=what)
    dt, out, full = call(42)
    flop = 5 * dit_oracle.flops_per_forward(full, 994)
    return dict(seconds=dt, frames=9, frames_per_s=9.0 / dt, sample_forwards=5, tflops=flop / dt / 1e12, threads=threads,
                finite=bool(torch.isfinite(out).all()), what=what)


def cpu_baseline(with_c1=True, c1_budget=150.0):
    """One of the 42 DiT blocks of one sample-forward at the C2 token count, fp32, on the host cores, through the CPU
    oracle; PROJECTED to frames/s of the whole workload (x 42 layers x 102 forwards per 49 frames)."""
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
    out = dict(value=49.0 / per_video, unit="frames/s", cores=threads,
               kind="port (projected: 1 of 42 layers timed, scaled)", projected=True,
               sample="one 1-layer CogVideoX-5B forward at the C2 token count (17,776 tokens, fp32 oracle, %.1f s) "
                      "scaled x42 layers x102 sample-forwards per 49-frame video (projected, BASELINE.md CPU row 3)" % dt,
               seconds_sampled=dt, flops_sampled=dit_oracle.flops_per_forward(cfg, tokens))
    del w, hs, ehs
    if with_c1:
        try:
            # 994 tokens: 256 threads are 6x SLOWER than 8 here (measured on the GPU box: 43.9 s vs 6.8 s for the 2-layer
            # probe) -- the small GEMMs drown in synchronisation; 32 threads is the advisor's / VERDICT's setting
            out["c1_call"] = cpu_c1_call(min(threads, 32), c1_budget)
        except Exception as e:  # the C1 leg must never take the bench line down
            out["c1_call"] = {"error": repr(e)}
    out["filters"] = cpu_filter_baseline()
    return out


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


# ---------------------------------------------------------------------------------------------------------------------
# workloads: each builds the drop-in pipeline + the __call__ kwargs of one video and knows its algorithmic work
# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    name = ""
    metric = ""
    frames = 0
    steps_per_video = 0
    dtype = "bf16"
    attn_kernel = ""
    peak = MFMA_PEAK_TFLOPS

    def __init__(self, args, dev, rank, world, split):
        self.args, self.dev, self.rank, self.world, self.split = args, dev, rank, world, split

    def seed(self):
        # run.py:94 uses 42; --seed-offset lets a 1-rank run reproduce rank r of an N-rank run (tests/test_gpu_dp_launch.py)
        return 42 + self.args.seed_offset + (self.rank // 2 if self.split is not None else self.rank)


class C2(Workload):
    """BASELINE config 2 = the metric's own configuration."""
    name = "c2"
    metric = "frames/sec (whole node) CogVideoX-5B-I2V 49f x 50-step ALG"
    frames, steps_per_video = 49, 50
    attn_kernel = "flash_attn_d64_pipe_kernel"   # main launch of the pre-scaled call (ALG_ATTN_PP=4, the default)
    S, D, Hn, T = 17776, 3072, 48, 226
    describe = ("BASELINE config 2: CogVideoX-5B-I2V bf16, 49 frames @ 480x720, 50 steps, ALG interval down_up "
                "(resize_factor 0.25, interval [0, 0.04]), guidance 6.0; one video per GPU")
    data = "synthetic (seeded random-init weights at CogVideoX-5B-I2V shapes, seeded latents/embeddings)"

    def build(self):
        from alg_amd import parallel, weights as W
        from alg_amd import (CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel,
                             CogVideoXTransformerConfig)
        dev = self.dev
        cfg = self.cfg = CogVideoXTransformerConfig(num_layers=self.args.layers or 42)
        self.layers = cfg.num_layers
        sd = parallel.broadcast_state_dict(lambda: W.synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev),
                                           W.parameter_shapes(cfg), dev)
        self.model = CogVideoXTransformer3DModel(cfg, sd, device=dev)
        del sd
        self.sched = CogVideoXDDIMScheduler()
        self.pipe = CogVideoXImageToVideoPipeline(transformer=self.model, scheduler=self.sched).to(dev)
        g = torch.Generator().manual_seed(self.seed())
        F_lat, C, Hh, Ww = 13, 16, 60, 90
        self.numel = F_lat * C * Hh * Ww
        self.latents0 = torch.randn(1, F_lat, C, Hh, Ww, generator=g).to(dev, torch.bfloat16)
        first = (torch.randn(1, 1, C, Hh, Ww, generator=g) * 0.7).to(dev, torch.bfloat16)
        pos = torch.randn(1, 226, 4096, generator=g).to(dev, torch.bfloat16)
        neg = torch.randn(1, 226, 4096, generator=g).to(dev, torch.bfloat16)
        self.kwargs = dict(image_latents=first, latents=self.latents0, prompt_embeds=pos, negative_prompt_embeds=neg,
                           height=480, width=720, num_frames=49, num_inference_steps=50, guidance_scale=6.0,
                           use_low_pass_guidance=True, lp_filter_type="down_up", lp_filter_in_latent=True,
                           lp_resize_factor=0.25, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
                           schedule_interval_end_time=0.04, output_type="latent", cfg_split=self.split)
        self.min_warmup = 3  # steps 0/1 are the 3-pass steps, step 2 the first 2-pass one: both workspaces exist after 3

    def instrument(self, kinds):
        """Bench-side HIP-event brackets around the two non-DiT launches of a step (the DiT has its own `profile` hook)."""
        from alg_amd import lp_utils
        self.model.profile = kinds
        if kinds is None:
            lp_utils.apply_low_pass_filter = self._orig_filter
            self.sched.fused_cfg_step_ = self._orig_step
            return
        self._orig_filter, self._orig_step = lp_utils.apply_low_pass_filter, self.sched.fused_cfg_step_

        def filt(x, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self._orig_filter(x, *a, **k)
            if y is not x:
                e1.record()
                kinds.setdefault("down_up", []).append((e0, e1))
            return y

        def step(pred, lat, n, gs, t):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig_step(pred, lat, n, gs, t)
            e1.record()
            kinds.setdefault("cfg_step_%d" % n, []).append((e0, e1))
            return r

        lp_utils.apply_low_pass_filter, self.sched.fused_cfg_step_ = filt, step

    def roofline(self, ms, forwards_local, elapsed):
        S, D, Hn = self.S, self.D, self.Hn
        L = self.layers
        attn_flops_total = 4.0 * S * S * 64 * Hn * forwards_local * L
        attn_time_total = sum(ms.get("attn", [])) / 1e3
        gemm_flops = dict(gemm_qk=2.0 * S * D * 2 * D, gemm_vt=2.0 * S * D * D, gemm_out=2.0 * S * D * D,
                          gemm_ff1=2.0 * S * D * 4 * D, gemm_ff2=2.0 * S * D * 4 * D)
        if "gemm_qkv" in ms:   # the Q|K and V^T projections as ONE alg_gemm_bf16_pair launch (the default since round 4)
            gemm_flops["gemm_qkv"] = gemm_flops.pop("gemm_qk") + gemm_flops.pop("gemm_vt")
        extra = {}
        for k, f in gemm_flops.items():
            tt = sum(ms.get(k, [])) / 1e3
            if tt > 0:
                extra[k + "_tflops"] = f * forwards_local * L / tt / 1e12
        gemm_time_total = sum(sum(ms.get(k, [])) for k in gemm_flops) / 1e3
        if gemm_time_total > 0:
            extra["gemm_all_tflops"] = sum(gemm_flops.values()) * forwards_local * L / gemm_time_total / 1e12
        mean = {k: sum(v) / len(v) for k, v in ms.items() if v}
        if "down_up" in mean:
            extra["down_up_gbs"] = 2.0 * self.numel * 2 / (mean["down_up"] / 1e3) / 1e9   # read + write, bf16
        for n in (2, 3):
            k = "cfg_step_%d" % n
            if k in mean:
                extra[k + "_gbs"] = (n + 2) * self.numel * 2 / (mean[k] / 1e3) / 1e9       # n bf16 preds + latents r/w
        for k in ("ln_mod", "qk_norm_rope"):
            tt = sum(ms.get(k, [])) / 1e3
            if tt > 0:
                per = (2.0 * S * D * 2) if k == "ln_mod" else (2.0 * S * 2 * D * 2)
                launches_per_layer = 2 if k == "ln_mod" else 1
                extra[k + "_gbs"] = per * forwards_local * L * launches_per_layer / tt / 1e9
        if "qk_norm_rope" not in ms and "gemm_qkv" in ms:
            extra["qk_norm_rope"] = "inside gemm_qkv's store loop (alg_gemm_bf16_pair_qk): gemm_qkv's time includes it"
        extra["time_share"] = {k: round(sum(v) / 1e3 / elapsed, 4) for k, v in ms.items()}
        attn_tflops = attn_flops_total / attn_time_total / 1e12 if attn_time_total > 0 else 0.0
        roofline = dict(bound="mfma", kernel=self.attn_kernel, achieved=attn_tflops, peak=self.peak,
                        unit="TFLOP/s", frac=attn_tflops / self.peak, traffic=None,
                        launches=len(ms.get("attn", [])),
                        mean_launch_ms=(sum(ms["attn"]) / len(ms["attn"])) if ms.get("attn") else None, extra=extra)
        # A launch's duration is proportional to the samples in its forward (3 on an ALG step, 2 otherwise): the mean per kind, and
        # the mean rocprofv3 --stats would print for the same command (its average runs over warm-up + timed launches alike)
        passes = getattr(self, "step_passes", [])
        if ms.get("attn") and len(ms["attn"]) == len(passes) * L:
            by = {}
            for j, n in enumerate(passes):
                by.setdefault(str(n), []).extend(ms["attn"][j * L:(j + 1) * L])
            roofline["mean_launch_ms_by_samples"] = {k: sum(v) / len(v) for k, v in by.items()}
            roofline["launches_by_samples"] = {k: len(v) for k, v in by.items()}
            allp = list(getattr(self, "warmup_passes", [])) + list(passes)
            per = {int(k): sum(v) / len(v) for k, v in by.items()}
            if all(n in per for n in allp):
                roofline["mean_launch_ms_incl_warmup_expected"] = sum(per[n] for n in allp) / len(allp)
        tr = pmc_traffic(self.attn_kernel)
        if tr is not None:  # `traffic` = HBM bytes per launch (PMC, corrected); how it was derived goes next to it
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_detail"] = tr
        flops_total = attn_flops_total + sum(gemm_flops.values()) * forwards_local * L
        # (the matrix-pipe ceiling of THIS box on random operands is measured in the run: main() -> calibrate_box(), `calibration`)
        extra["whole_step_mfma_frac"] = flops_total / elapsed / 1e12 / MFMA_PEAK_TFLOPS
        # tile-count quantisation of the persistent GEMM (256x256 tiles on `cus` workgroups)
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count & ~7
        tiles = lambda m, n, batch: batch * ((m + 255) // 256) * ((n + 255) // 256)
        tail = {}
        for name, (m, n, batch) in dict(gemm_qk=(S, 2 * D, 2), gemm_vt=(D, S, 2), gemm_out=(S, D, 2),
                                        gemm_ff1=(S, 4 * D, 2), gemm_ff2=(S, D, 2)).items():
            t = tiles(m, n, batch)
            tail[name] = {"tiles": t, "rounds_of_work": round(t / cus, 3), "rounds_paid": -(-t // cus)}
        if "gemm_qkv" in ms:
            t = tail.pop("gemm_qk")["tiles"] + tail.pop("gemm_vt")["tiles"]
            tail["gemm_qkv"] = {"tiles": t, "rounds_of_work": round(t / cus, 3), "rounds_paid": -(-t // cus)}
        extra["gemm_tile_rounds"] = tail
        return roofline

    def config(self, forwards):
        return {"workload": self.describe, "layers": self.layers, "tokens": self.S, "dit_sample_forwards": forwards}


class _WanBase(Workload):
    attn_kernel = "flash_attn_d128_q64_kernel"   # the default d = 128 self-attention since round 4 (attention128_q64.hip)
    D, heads, ffn = 5120, 40, 13824
    fp8 = False

    def build(self):
        from alg_amd import UniPCMultistepScheduler, WanImageToVideoPipeline, WanTransformer3DModel, WanTransformerConfig
        dev = self.dev
        cfg = self.cfg = WanTransformerConfig(num_layers=self.args.layers or 40)
        self.layers = cfg.num_layers
        # weights exist on rank 0 only and reach the other ranks through ONE bucketed RCCL broadcast (BASELINE configs 3 - 5:
        # "RCCL weight bcast only"); world size 1: the dict itself
        from alg_amd import parallel
        from alg_amd.transformer_wan import synthetic_state_dict
        sd = synthetic_state_dict(cfg, device=dev) if self.rank == 0 else None
        self.model = WanTransformer3DModel(cfg, parallel.broadcast_loaded_state_dict(sd, dev), device=dev, fp8=self.fp8)
        del sd
        self.pipe = WanImageToVideoPipeline(transformer=self.model, scheduler=UniPCMultistepScheduler(flow_shift=5.0)).to(dev)
        g = torch.Generator().manual_seed(self.seed())
        h, w_, nf = self.height, self.width, 81
        f_lat = (nf - 1) // 4 + 1
        self.S = f_lat * (h // 16) * (w_ // 16)
        self.numel = 16 * f_lat * (h // 8) * (w_ // 8)
        bf = torch.bfloat16
        cond = torch.randn(1, 20, f_lat, h // 8, w_ // 8, generator=g) * 0.7
        cond[:, :4] = 0.0
        cond[:, :4, 0] = 1.0                      # first-frame mask channels (wan:444-456)
        self.kwargs = dict(prompt_embeds=torch.randn(1, 512, 4096, generator=g).to(dev, bf),
                           negative_prompt_embeds=torch.randn(1, 512, 4096, generator=g).to(dev, bf),
                           image_embeds=torch.randn(1, 257, 1280, generator=g).to(dev, bf),
                           image_condition=cond.to(dev), latents=torch.randn(1, 16, f_lat, h // 8, w_ // 8, generator=g).to(dev),
                           height=h, width=w_, num_frames=nf, guidance_scale=5.0, output_type="latent",
                           use_low_pass_guidance=True, lp_filter_in_latent=True, cfg_split=self.split, **self.alg)
        self.min_warmup = 1

    def instrument(self, kinds):
        self.model.profile = kinds

    def roofline(self, ms, forwards_local, elapsed):
        S, D, Ff, L = self.S, self.D, self.ffn, self.layers
        per_fwd = {"gemm_qk": 2.0 * S * D * 2 * D, "gemm_vt": 2.0 * S * D * D, "gemm_out": 2.0 * S * D * D,
                   "gemm_cq": 2.0 * S * D * D, "gemm_cout": 2.0 * S * D * D, "gemm_ff1": 2.0 * S * D * Ff,
                   "gemm_ff2": 2.0 * S * D * Ff, "attn_self": 4.0 * S * S * D}
        if "gemm_qkv" in ms:   # bf16: Q|K and V^T projections as one alg_gemm_bf16_pair launch
            per_fwd["gemm_qkv"] = per_fwd.pop("gemm_qk") + per_fwd.pop("gemm_vt")
        extra = {}
        for k, f in per_fwd.items():
            tt = sum(ms.get(k, [])) / 1e3
            if tt > 0:
                extra[k + "_tflops"] = f * forwards_local * L / tt / 1e12
        extra["time_share"] = {k: round(sum(v) / 1e3 / elapsed, 4) for k, v in ms.items()}
        total = sum(per_fwd.values()) + 4.0 * S * (512 + 257) * D
        extra["whole_step_tflops"] = total * forwards_local * L / elapsed / 1e12
        a = extra.get("attn_self_tflops", 0.0)
        roofline = dict(bound="mfma", kernel=self.attn_kernel, achieved=a, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=a / MFMA_PEAK_TFLOPS, traffic=None, launches=len(ms.get("attn_self", [])),
                        mean_launch_ms=(sum(ms["attn_self"]) / len(ms["attn_self"])) if ms.get("attn_self") else None,
                        extra=extra)
        # the committed PMC pass profiles the 2-sample Wan-480p launch (scripts/kbench.py --only attn128): that IS the C3 launch
        tr = pmc_traffic(self.attn_kernel) if self.S == 32760 else None
        if tr is not None:
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_detail"] = tr
        return roofline

    def config(self, forwards):
        return {"workload": self.describe, "layers": self.layers, "tokens": self.S, "dit_sample_forwards": forwards}


class C3(_WanBase):
    name = "c3"
    metric = "frames/sec (whole node) Wan2.1-I2V-14B 81f x 40-step ALG (gaussian_blur, linear decay)"
    frames, steps_per_video = 81, 40
    height, width = 480, 832
    alg = dict(num_inference_steps=40, lp_filter_type="gaussian_blur", lp_blur_sigma=15.0, lp_blur_kernel_size=9,
               lp_strength_schedule_type="linear", schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0,
               schedule_linear_end_time=0.5)
    describe = ("BASELINE config 3: Wan2.1-I2V-14B bf16, 81 frames @ 832x480, 40 steps, ALG gaussian_blur in latent "
                "(sigma 15, k 9, linear decay to 0 at 0.5), guidance 5.0; one video per GPU")
    data = "synthetic (seeded random-init weights at Wan2.1-I2V-14B shapes, seeded latents/condition/embeddings)"


class C5(_WanBase):
    name = "c5"
    metric = "frames/sec (whole node) Wan2.1-I2V-14B fp8 81f @ 1280x720 x 50-step ALG (interval)"
    frames, steps_per_video = 81, 50
    height, width = 720, 1280
    fp8 = True
    dtype = "fp8 (e4m3 block linears; bf16 attention / norms / residual)"
    alg = dict(num_inference_steps=50, lp_filter_type="down_up", lp_resize_factor=0.4, lp_strength_schedule_type="interval",
               schedule_interval_start_time=0.0, schedule_interval_end_time=0.2)
    describe = ("BASELINE config 5: Wan2.1-I2V-14B fp8 weights, 81 frames @ 1280x720, 50 steps, ALG interval down_up "
                "(factor 0.4, interval [0, 0.2]), guidance 5.0; --cfg-split puts the cond/uncond pair on 2 GPUs")
    data = "synthetic (seeded random-init weights at Wan2.1-I2V-14B shapes quantised to e4m3, seeded inputs)"


class C4(Workload):
    name = "c4"
    metric = "frames/sec (whole node) HunyuanVideo-I2V 129f @ 1280x720 x 50-step ALG (down_up)"
    frames, steps_per_video = 129, 50
    attn_kernel = "flash_attn_d128_q64_kernel"   # the default d = 128 self-attention since round 4 (attention128_q64.hip)
    describe = ("BASELINE config 4: HunyuanVideo-I2V bf16, 129 frames @ 1280x720, 50 steps, ALG interval down_up (factor "
                "0.625, interval [0, 0.04]), embedded guidance 6.0 (single-pass ALG branch); one prompt per GPU")
    data = "synthetic (seeded random-init weights at HunyuanVideo-I2V shapes, seeded latents/embeddings)"

    def build(self):
        from alg_amd import (FlowMatchEulerDiscreteScheduler, HunyuanVideoImageToVideoPipeline, HunyuanVideoTransformer3DModel,
                             HunyuanVideoTransformerConfig)
        dev = self.dev
        cfg = self.cfg = HunyuanVideoTransformerConfig()
        self.layers = cfg.num_layers + cfg.num_single_layers
        from alg_amd import parallel
        from alg_amd.transformer_hunyuan_video import synthetic_state_dict
        sd = synthetic_state_dict(cfg, device=dev) if self.rank == 0 else None     # rank 0 only; one RCCL broadcast
        self.model = HunyuanVideoTransformer3DModel(cfg, parallel.broadcast_loaded_state_dict(sd, dev), device=dev)
        del sd
        self.pipe = HunyuanVideoImageToVideoPipeline(transformer=self.model,
                                                     scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0)).to(dev)
        g = torch.Generator().manual_seed(self.seed())
        bf = torch.bfloat16
        h, w_ = 720, 1280
        n_tok, self.n_valid = 256, 48
        self.S = 33 * (h // 16) * (w_ // 16)
        self.J = self.S + n_tok
        mask = torch.cat([torch.ones(1, self.n_valid), torch.zeros(1, n_tok - self.n_valid)], dim=1)
        self.kwargs = dict(prompt_embeds=torch.randn(1, n_tok, 4096, generator=g).to(dev, bf),
                           pooled_prompt_embeds=torch.randn(1, 768, generator=g).to(dev, bf), prompt_attention_mask=mask,
                           negative_prompt=None, image_latents=(torch.randn(1, 16, 1, h // 8, w_ // 8, generator=g) * 0.7).to(dev),
                           height=h, width=w_, num_frames=129, num_inference_steps=50, guidance_scale=6.0, true_cfg_scale=1.0,
                           i2v_stable=True, use_low_pass_guidance=True, lp_filter_type="down_up", lp_filter_in_latent=True,
                           lp_resize_factor=0.625, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
                           schedule_interval_end_time=0.04, output_type="latent",
                           generator=torch.Generator().manual_seed(self.seed()))
        self.min_warmup = 1

    def instrument(self, kinds):
        self.model.profile = kinds

    def roofline(self, ms, forwards_local, elapsed):
        D = 3072
        per_fwd = 4.0 * self.J * (self.S + self.n_valid) * D
        tt = sum(ms.get("attn_self", [])) / 1e3
        a = per_fwd * forwards_local * self.layers / tt / 1e12 if tt > 0 else 0.0
        extra = {"time_share": {k: round(sum(v) / 1e3 / elapsed, 4) for k, v in ms.items()}}
        # the large GEMM families (latent rows of the 20 dual-stream blocks, joint rows of the 40 single-stream blocks; the prompt
        # rows' own small GEMMs are not bracketed): FLOP summed over the blocks of ONE forward
        S, J, M, nd, ns = self.S, self.J, 4 * D, self.cfg.num_layers, self.cfg.num_single_layers
        fam = {"gemm_qk": 2.0 * D * 2 * D * (nd * S + ns * J), "gemm_vt": 2.0 * D * D * (nd * S + ns * J),
               "gemm_out": 2.0 * D * D * nd * S, "gemm_ff1": 2.0 * D * M * (nd * S + ns * J), "gemm_ff2": 2.0 * M * D * nd * S,
               "gemm_out_mlp": 2.0 * (D + M) * D * ns * J}
        for k, f in fam.items():
            t_k = sum(ms.get(k, [])) / 1e3
            if t_k > 0:
                extra[k + "_tflops"] = f * forwards_local / t_k / 1e12
        # every block is 12 D^2 multiply-adds per token of linears (dual: qkv 3 + out 1 + mlp 8; single: 7 + 5) + its attention
        extra["whole_step_tflops"] = (24.0 * self.J * D * D + per_fwd) * forwards_local * self.layers / elapsed / 1e12
        return dict(bound="mfma", kernel=self.attn_kernel, achieved=a, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=a / MFMA_PEAK_TFLOPS, traffic=None, launches=len(ms.get("attn_self", [])),
                    mean_launch_ms=(sum(ms["attn_self"]) / len(ms["attn_self"])) if ms.get("attn_self") else None,
                    extra=extra)

    def config(self, forwards):
        return {"workload": self.describe, "layers": self.layers, "tokens": self.J, "dit_sample_forwards": forwards}


WORKLOADS = {"c2": C2, "c3": C3, "c4": C4, "c5": C5}


# ---------------------------------------------------------------------------------------------------------------------
def run_steps(wl, k_steps):
    """`k_steps` loop iterations through pipe.__call__: whole videos while they fit, the last call cut short by the
    reference's own interrupt flag (cog:1006 / wan:845 / hy:1127) from callback_on_step_end.  Returns the number of DiT
    sample-forwards THIS rank executed."""
    forwards = 0
    left = k_steps
    while left > 0:
        take = min(left, wl.steps_per_video)
        trace = []

        def cb(pipe, i, t, kw, take=take):
            if i + 1 >= take:
                pipe._interrupt = True   # what diffusers' callbacks do to stop a run: the loop `continue`s from here on
            return {}

        wl.last_out = wl.pipe(callback_on_step_end=cb, step_trace=trace, **wl.kwargs).frames
        for rec in trace:
            n = rec[2]
            mine = len(wl.split.my_passes(n)) if wl.split is not None else n
            forwards += mine
            wl.__dict__.setdefault("step_passes", []).append(mine)      # samples per DiT forward of this step, in launch order (roofline: launch mix)
        left -= take
    return forwards


def private_loop_crosscheck(wl, k_steps):
    """Round 1's bench-private restatement of the C2 step (forward_assembled + fused step, no pipeline plumbing), kept as
    a cross-check of the __call__ timing only."""
    from alg_amd import lp_utils
    from alg_amd.pipeline_cogvideox_image2video_lowpass import get_resize_crop_region_for_grid, rotary_tables
    dev, kw = wl.dev, wl.kwargs
    sched = wl.sched
    sched.set_timesteps(50)
    timesteps = sched.timesteps
    lat = kw["latents"].clone()
    image_latents = torch.zeros_like(lat)
    image_latents[:, :1] = kw["image_latents"]
    emb2 = torch.cat([kw["negative_prompt_embeds"], kw["prompt_embeds"]]).contiguous()
    emb3 = torch.cat([kw["negative_prompt_embeds"], kw["negative_prompt_embeds"], kw["prompt_embeds"]]).contiguous()
    crops = get_resize_crop_region_for_grid((30, 45), 45, 30)
    rope = tuple(t.to(dev) for t in rotary_tables(64, crops, (30, 45), 13))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k_steps):
        i = i % 50
        t = timesteps[i]
        s = lp_utils.get_lp_strength(i, 50, "interval", 0.0, 0.04, 1.0, 0.0, 0.5, 10.0)
        lp = lp_utils.apply_low_pass_filter(image_latents, "down_up", 15.0 * s, 0.02734375, 1.0 - 0.75 * s)
        conds = [lp, lp] if s == 0 else [image_latents, lp, lp]
        ts = torch.full((len(conds),), float(t), device=dev)
        pred = wl.model.forward_assembled(lat, conds, emb2 if s == 0 else emb3, ts, rope)
        sched.fused_cfg_step_(pred, lat, len(conds), 6.0, t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k_steps * 1e3


def timed_region(wl, warmup, steps, parallel, measure_box=False):
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; HIP-event brackets of the
    kernel families live only inside the timed region (their cost is measured by the `events` arm of --ab).  Returns (elapsed
    seconds = max over ranks, sample-forwards of this rank, per-family event times in ms).  measure_box: the attention
    kernel's own clock taps are switched on (two scalar counter reads in every 64th workgroup; results unaffected) and a
    background thread samples package power / sclk for the length of the region -> wl.box."""
    from alg_amd import _lib
    wl.step_passes = []
    run_steps(wl, max(warmup, wl.min_warmup))
    wl.warmup_passes = list(wl.step_passes)
    kinds = {}  # kernel family -> list of event pairs
    wl.step_passes = []
    wl.instrument(kinds)
    taps = smi = None
    try:
        # the tap is a raw device pointer inside the library: whatever happens in the region, it is taken out again before
        # `taps` can be freed; the library itself refuses the tap on a capturing stream
        if measure_box:
            taps = torch.zeros(512, 4, dtype=torch.int64, device=wl.dev)
            _lib.attn_clock_tap(taps)
        parallel.barrier()
        torch.cuda.synchronize()
        if measure_box:
            smi = SmiSampler(wl.dev.index or 0)
            smi.__enter__()
        t0 = time.perf_counter()
        forwards = run_steps(wl, steps)
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
    finally:
        if smi is not None:
            smi.__exit__(None, None, None)
        wl.instrument(None)
        if measure_box:
            _lib.attn_clock_tap(None)
            torch.cuda.synchronize()
    if measure_box:
        wl.box = {"attn_kernel_shader_clock_mhz": _lib.clock_mhz_from_taps(taps, _lib.wall_clock_khz()),
                  "attn_kernel_clock_note": "cycle counter / constant-rate counter over the life of every 64th workgroup of the "
                                            "LAST attention launch of the timed region (%s)" % wl.attn_kernel,
                  "smi_during_timed_region": smi.summary()}
    elapsed = parallel.max_over_ranks(elapsed, wl.dev)
    return elapsed, forwards, {k: event_ms(v) for k, v in kinds.items()}


# A/B arms of the default bench line: each switches ONE schedule choice off (or on), 5 steps after 3 warm-up
# steps (loop iterations 0-2: both the 3-sample and the 2-sample launch shapes have run under the arm before the clock starts), same process, same box, no event brackets; the default arm is timed before and after.  kind "attr": attribute of the
# transformer; "env": an ALG_* option of the library (re-read with alg_reload_env); "events": the bench's own HIP-event brackets
# switched ON (what the instrumented headline region pays for them).
AB_ARMS = [
    ("attn_m16_statement", "env", "ALG_ATTN_PP", "7", "round 6: the d = 64 8-wave statement on v_mfma_f32_16x16x32_bf16 (attention64_m16.hip, ALG_ATTN_PP=7) as the OFF arm; > 0: the default (32x32x16 statement) is faster"),
    ("attn_pipelined", "env", "ALG_ATTN_PP", "0", "round 3: pipelined d = 64 attention vs the straight loop"),
    ("attn_split_tail", "env", "ALG_ATTN_SPLIT_TAIL", "0", "round 2: split-KV tail of the attention launch vs a single launch"),
    ("gemm_schedule11", "attr", "packed_weights", 0, "round 6: out / ff1 / ff2 on GEMM schedule 11 (1 x 4 waves, the weight pre-packed and loaded straight into registers) vs schedule 10 for all"),
    ("gemm_schedule10", "env", "ALG_GEMM_PIPE", "9", "round 6: GEMM schedules 10 / 11 (the asm K loop on v_mfma_f32_16x16x32_bf16) vs schedule 9 (the same loop on 32x32x16) for all"),
    ("gemm_schedule9", "env", "ALG_GEMM_PIPE", "6", "round 3: the asm K loop (here: schedule 10) vs the 8-wave ping-pong"),
    ("pair_qkv", "attr", "pair_qkv", 0, "round 4: Q|K and V^T projections as one persistent launch vs two launches"),
    ("events", "events", None, None, "the bench's own HIP-event brackets around every kernel family, switched ON (headline region has them)"),
]


def ab_arms(wl, arms, steps=5, warmup=3):
    from alg_amd import _lib

    box = {}     # arm -> package power / sclk during its timed steps (the arms' clocks say whether a delta is cycles or clock)

    def timed(with_events=False, tag=None):
        run_steps(wl, warmup)
        kinds = {} if with_events else None
        if with_events:
            wl.instrument(kinds)
        smi = SmiSampler(wl.dev.index or 0)
        smi.__enter__()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            run_steps(wl, steps)
            torch.cuda.synchronize()
        finally:
            smi.__exit__(None, None, None)
            if with_events:
                wl.instrument(None)
        dt = (time.perf_counter() - t0) / steps * 1e3
        sm = smi.summary()
        if tag is not None and sm:
            box[tag] = {"power_w": round(sm["power_w"]["mean"], 1), "sclk_mhz": round(sm["sclk_mhz"]["mean"], 1)}
        return dt

    res = {"steps": steps, "warmup": warmup, "default_ms_per_step": [timed(tag="default_first")], "arms": {}}
    for name, kind, key, off, what in arms:
        rec = {"what": what}
        try:
            if kind == "attr":
                if not hasattr(wl.model, key):
                    continue
                old = getattr(wl.model, key)
                setattr(wl.model, key, type(old)(off))
                try:
                    rec["off_ms_per_step"] = timed(tag=name)
                finally:
                    setattr(wl.model, key, old)
            elif kind == "env":
                old = os.environ.get(key)
                os.environ[key] = off
                _lib.reload_env()
                try:
                    rec["off_ms_per_step"] = timed(tag=name)
                finally:
                    if old is None:
                        os.environ.pop(key, None)
                    else:
                        os.environ[key] = old
                    _lib.reload_env()
                rec["off"] = "%s=%s" % (key, off)
            else:
                rec["off_ms_per_step"] = timed(with_events=True, tag=name)
        except Exception as e:
            rec["error"] = repr(e)
        res["arms"][name] = rec
    res["default_ms_per_step"].append(timed(tag="default_last"))
    res["smi_per_arm"] = box
    base = sum(res["default_ms_per_step"]) / 2
    res["default_spread_pct"] = round(abs(res["default_ms_per_step"][0] - res["default_ms_per_step"][1]) / base * 100, 2)
    for rec in res["arms"].values():
        if "off_ms_per_step" in rec:   # > 0: the default (change ON) is faster by that much
            rec["default_gain_pct"] = round((rec["off_ms_per_step"] - base) / base * 100, 2)
    return res


def schedule_passes(wl):
    """passes per loop iteration of the workload's schedule: 3 while the reference's lp strength is non-zero, else 2 (wan:882-894);
    the HunyuanVideo configuration runs its single-pass branch throughout (hy:1196-1235)"""
    from alg_amd import lp_utils
    steps = wl.steps_per_video
    alg = getattr(wl, "alg", None)
    if alg is None:
        return [1] * steps
    sig = dict(lp_strength_schedule_type="none", schedule_interval_start_time=0.0, schedule_interval_end_time=0.05,
               schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0, schedule_linear_end_time=0.5,
               schedule_exp_decay_rate=10.0)
    sig.update({k: v for k, v in alg.items() if k in sig})
    return [3 if lp_utils.get_lp_strength(step_index=i, total_steps=steps, **sig) != 0.0 else 2 for i in range(steps)]


def time_two_pass_step(wl, parallel):
    """the C3 / C5 legs time loop iterations 0-1 (3-pass); a whole video also has 2-pass steps, whose launches
    quantise differently on 256 CUs.  One 2-pass step, MEASURED: the pipeline's loop entered at the schedule's first 2-pass
    iteration (`_first_step`: earlier iterations are skipped the way interrupted ones are), one warm-up step, one timed step.
    Returns seconds per 2-pass step, or None when the schedule has no 2-pass step (or the pipeline no such hook)."""
    passes = schedule_passes(wl)
    if 2 not in passes or not hasattr(wl.pipe, "_first_step"):
        return None
    first = passes.index(2)
    if first + 2 > len(passes) or passes[first + 1] != 2:
        return None
    wl.pipe._first_step = first
    last = {}

    def run(n_exec):
        trace = []

        def cb(pipe, i, t, kw):
            if i + 1 >= first + n_exec:
                pipe._interrupt = True            # (iterations are counted from 0: this one was number first + n_exec - 1)
            return {}
        wl.last_out = wl.pipe(callback_on_step_end=cb, step_trace=trace, **wl.kwargs).frames
        last["passes"] = [r[2] for r in trace]
    try:
        run(1)                                    # warm-up: the 2-sample launch shapes, the filter at this strength
        torch.cuda.synchronize()
        parallel.barrier()
        t0 = time.perf_counter()
        run(2)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        wl.pipe._first_step = 0
    if last["passes"] != [2, 2]:
        raise RuntimeError("expected two 2-pass steps from iteration %d on, the pipeline ran %r" % (first, last["passes"]))
    return elapsed / 2.0


def whole_video_projection(wl, elapsed, forwards, two_pass_seconds=None):
    """PROJECTED whole-video rate of a C3 - C5 leg from MEASURED per-step times: the leg's timed steps are loop iterations 0-1
    (3-pass for Wan, the single pass of the HunyuanVideo branch), `two_pass_seconds` one measured 2-pass step (time_two_pass_step);
    a video = n3 x t3 + n2 x t2.  Without a measured 2-pass step the old assumption stands in (a 2-sample forward costs 2 / 3 of
    a 3-sample one) and the record says so."""
    passes = schedule_passes(wl)
    total = sum(passes)
    n3, n2, n1 = passes.count(3), passes.count(2), passes.count(1)
    t_leg_step = elapsed / 2.0                                     # the leg times 2 steps
    if n2 and two_pass_seconds is not None:
        seconds = (n3 + n1) * t_leg_step + n2 * two_pass_seconds
        what = "projected from two MEASURED step times: %d x %.3f s (loop iterations 0-1) + %d x %.3f s (one 2-pass step, iteration %d)" % (
            n3 + n1, t_leg_step, n2, two_pass_seconds, passes.index(2) + 1)
    else:
        seconds = total * elapsed / forwards
        what = "projected: schedule's passes per step x this leg's measured seconds per sample-forward"
    return {"sample_forwards_per_video": total, "frames_per_s": wl.frames / seconds, "seconds_per_video": seconds,
            "three_pass_steps": n3, "two_pass_steps": n2, "one_pass_steps": n1,
            "seconds_per_three_pass_step": t_leg_step if n3 else None, "seconds_per_two_pass_step": two_pass_seconds, "what": what}


def other_workloads(args, dev, parallel):
    """BASELINE configs 3 / 4 / 5 on the driver's clock.  After the C2 region (its weights freed), each of
    the other workloads is built at FULL depth and timed for 2 steps after 1 warm-up step through its pipeline's __call__ --
    the same code path and the same value formula as `bench.py --workload cX --steps 2 --warmup 1` (loop iterations 0-1, i.e.
    the 3-pass ALG steps of c3 / c5 and the single-pass branch of c4: the conservative end of each schedule)."""
    import gc
    res = {}
    for name in ("c3", "c4", "c5"):
        t_build = time.perf_counter()
        try:
            wl = WORKLOADS[name](args, dev, 0, 1, None)
            wl.build()
            torch.cuda.synchronize()
            t_build = time.perf_counter() - t_build
            elapsed, forwards, ms = timed_region(wl, 1, 2, parallel)
            rl = wl.roofline(ms, forwards, elapsed)
            two_pass = time_two_pass_step(wl, parallel)      # one measured 2-pass step of the schedule (None for c4)
            # same process, same box, same steps: the 64-query statement kernel (default) against the 32-query pipelined kernel
            # (ALG_ATTN128_Q64=0; round 4's compiler-scheduled 64-query kernel measured +2.6 ... 3.2 % over it at C4)
            ab = None
            if not args.no_ab:
                from alg_amd import _lib
                old_env = os.environ.get("ALG_ATTN128_Q64")
                os.environ["ALG_ATTN128_Q64"] = "0"
                _lib.reload_env()
                try:
                    e0, _, ms0 = timed_region(wl, 1, 2, parallel)
                    ab = {"off": "ALG_ATTN128_Q64=0 (flash_attn_d128_pipe_kernel)", "off_ms_per_step": e0 / 2 * 1e3,
                          "default_gain_pct": round((e0 - elapsed) / elapsed * 100, 2),
                          "off_attn_ms_per_step": sum(ms0.get("attn_self", [])) / 2, "default_attn_ms_per_step": sum(ms.get("attn_self", [])) / 2}
                finally:
                    if old_env is None:
                        os.environ.pop("ALG_ATTN128_Q64", None)
                    else:
                        os.environ["ALG_ATTN128_Q64"] = old_env
                    _lib.reload_env()
            res[name] = {
                "metric": wl.metric, "frames_per_s": wl.frames * 2 / wl.steps_per_video / elapsed, "ms_per_step": elapsed / 2 * 1e3,
                "steps": 2, "warmup": 1, "dit_sample_forwards": forwards, "layers": wl.layers, "tokens": wl.config(forwards)["tokens"],
                "dtype": wl.dtype, "attn_kernel": rl["kernel"], "attn_tflops": rl["achieved"], "attn_frac": rl["frac"],
                "whole_step_tflops": rl["extra"].get("whole_step_tflops"), "time_share": rl["extra"].get("time_share"),
                "gemm_tflops": {k: round(v, 1) for k, v in rl["extra"].items() if k.startswith("gemm_") and k.endswith("_tflops")},
                "finite": bool(torch.isfinite(wl.last_out.float()).all().item()), "build_seconds": round(t_build, 1),
                "peak_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                "ab_attn128_q64_statement": ab,
                "whole_video_projected": whole_video_projection(wl, elapsed, forwards, two_pass),
            }
        except Exception as e:   # an auxiliary workload must never take the headline line down
            res[name] = {"error": repr(e)}
        wl = None
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
    return res


def coerce_like(current, text, name):
    """--set ATTR=VALUE: the value converted to the attribute's CURRENT type, or SystemExit ('pair_qkv=false' used
    to be stored as the truthy string 'false')."""
    t = text.strip().lower()
    if isinstance(current, bool):
        if t in ("1", "true", "on", "yes"):
            return True
        if t in ("0", "false", "off", "no"):
            return False
        raise SystemExit("--set %s=%s: expected a boolean (0/1/true/false/on/off)" % (name, text))
    try:
        if isinstance(current, int):
            return int(t, 0)
        if isinstance(current, float):
            return float(t)
    except ValueError:
        raise SystemExit("--set %s=%s: cannot convert to %s" % (name, text, type(current).__name__))
    if isinstance(current, str) or current is None:
        return text
    raise SystemExit("--set %s: attributes of type %s cannot be set from the command line" % (name, type(current).__name__))


def self_launch(args):
    """`python bench.py --gpus N` run plainly: become the launcher of N ranks (one per GPU) on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = the metric's configuration (default); c3 / c4 / c5 = the other BASELINE configs as bench lines")
    ap.add_argument("--layers", type=int, default=0, help="debug only: fewer layers invalidates the metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c1", action="store_true", help="skip the timed C1 __call__ leg of the CPU baseline")
    ap.add_argument("--seed-offset", type=int, default=0, help="debug: added to every rank's video seed")
    ap.add_argument("--dump-latents", default="", metavar="PREFIX",
                    help="debug: every rank saves the final latents of its last __call__ to PREFIX.rank<r>.pt")
    ap.add_argument("--no-calibration", action="store_true", help="skip the 3 s matrix-pipe calibration of the box")
    ap.add_argument("--ab", action="store_true", help="in-run A/B arms (roofline.extra.ab; workloads.*.ab_attn128_q64_statement): ON by "
                                                      "default for c2 at 1 GPU -- the flag exists so that a command line can say so")
    ap.add_argument("--no-ab", action="store_true", help="c2 at 1 GPU: skip the in-run A/B arms (roofline.extra.ab, ~40 s)")
    ap.add_argument("--c1-budget", type=float, default=150.0,
                    help="seconds the C1 CPU leg may take (a 2-layer probe projects it first; beyond the budget the projection is reported)")
    ap.add_argument("--cfg-split", action="store_true",
                    help="opt-in: cond/uncond CFG passes of one video on a pair of GPUs (latency mode, one all-gather per step)")
    ap.add_argument("--cross-check", action="store_true", help="c2: also time round 1's bench-private loop")
    ap.add_argument("--filters-only", action="store_true", help="debug: only the low-pass kernel micro-benchmark")
    ap.add_argument("--set", action="append", default=[], metavar="ATTR=VALUE",
                    help="A/B runs: set an attribute of the transformer (e.g. pair_qkv=0, attn_prescale=0) before the warm-up")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="c2 at 1 GPU: skip the c3 / c4 / c5 legs (2 timed full-depth steps each, ~3 min) that ride behind the headline")
    args = ap.parse_args()
    if args.filters_only:
        print(json.dumps(filter_microbench(torch.device("cuda:0"))))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import alg_amd  # noqa: F401
    from alg_amd import parallel

    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ALG hot path is HIP-only")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    split = None
    if args.cfg_split:
        # BASELINE config 5 style: the cond / uncond CFG passes of ONE video on a pair of GPUs (one small all-gather per
        # step); pairs are independent videos.  Opt-in: the default bench is pure data parallelism over videos.
        if world % 2:
            raise SystemExit("--cfg-split needs an even number of GPUs")
        if args.workload == "c4":
            raise SystemExit("--cfg-split: the HunyuanVideo ALG branch is single-pass (no CFG pair to split)")
        split = parallel.CFGPairSplit.from_world()

    wl = WORKLOADS[args.workload](args, dev, rank, world, split)
    t_b0 = time.perf_counter()
    wl.build()
    torch.cuda.synchronize()
    build_seconds = time.perf_counter() - t_b0
    for kv in args.set:
        k, v_ = kv.split("=", 1)
        if not hasattr(wl.model, k):
            raise SystemExit("--set: the transformer has no attribute %r" % k)
        setattr(wl.model, k, coerce_like(getattr(wl.model, k), v_, k))
    elapsed, forwards, ms = timed_region(wl, args.warmup, args.steps, parallel, measure_box=(world == 1))
    calibration = None
    if rank == 0 and world == 1 and not args.no_calibration:
        try:
            calibration = calibrate_box(dev)      # right BEHIND the timed region: the chip in the thermal state the headline ran in
        except Exception as e:                    # an auxiliary leg must never take the headline line down
            calibration = {"error": repr(e)}

    if args.dump_latents:
        torch.save(wl.last_out.detach().cpu(), "%s.rank%d.pt" % (args.dump_latents, rank))
    n_videos_parallel = world // 2 if split else world
    value = wl.frames * args.steps / wl.steps_per_video * n_videos_parallel / elapsed
    roofline = wl.roofline(ms, forwards, elapsed)

    cfgd = wl.config(forwards)
    cfgd["parallelism"] = ("cfgpair2xdp%d" % (world // 2)) if split else ("dp%d" % world)
    cfgd["videos"] = args.steps / wl.steps_per_video * n_videos_parallel
    cfgd["timed_through"] = "%s.__call__ (interrupt after the step budget)" % type(wl.pipe).__name__
    out = {
        "metric": wl.metric, "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": wl.dtype, "data": wl.data, "config": cfgd, "seconds": elapsed,
        "finite": bool(torch.isfinite(wl.last_out.float()).all().item()), "roofline": roofline,
    }
    if world > 1 or parallel.FORCE_COLLECTIVES:
        # proof that N ranks sat on N distinct GPUs, and what the ONE collective of the data-parallel path cost (start-up only);
        # ALG_DIST_FORCE=1 runs the same collectives on a one-rank group (the one-GPU rehearsal of the RCCL path)
        out["dist_backend"] = torch.distributed.get_backend()
        out["rccl_version"] = parallel.rccl_version() if out["dist_backend"] == "nccl" else None
        seen = parallel.ranks_seen(dev)
        out["ranks_seen"] = seen["ranks"]
        out["distinct_gpus"] = seen["distinct_gpus"]
        out["bcast_seconds"] = parallel.max_over_ranks(parallel.BCAST_STATS["seconds"], dev)
        out["bcast_gbytes"] = parallel.BCAST_STATS["bytes"] / 1e9
        out["bcast_collectives"] = parallel.BCAST_STATS["collectives"]
    out["build_seconds"] = build_seconds
    if calibration is not None:
        out["calibration"] = calibration
    if getattr(wl, "box", None):
        out["roofline"]["extra"]["box"] = wl.box
        clk = wl.box.get("attn_kernel_shader_clock_mhz")
        if clk and out["roofline"].get("achieved"):
            # matrix-pipe busy share of the attention kernel AT THE CLOCK IT RAN AT: achieved / (CUs x 4 SIMDs x 1024 FLOP/cycle x f)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            out["roofline"]["extra"]["attn_pipe_busy_at_its_clock"] = out["roofline"]["achieved"] / (cus * 4 * 1024.0 * clk["mean"] * 1e6 / 1e12)
        if calibration is not None and calibration.get("mfma_sustained_random_operands_tflops") and out["roofline"].get("achieved"):
            out["roofline"]["extra"]["attn_frac_of_box_sustained_mfma"] = out["roofline"]["achieved"] / calibration["mfma_sustained_random_operands_tflops"]
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_ab and not args.layers and not args.set:
        try:
            out["roofline"]["extra"]["ab"] = ab_arms(wl, AB_ARMS)
        except Exception as e:
            out["roofline"]["extra"]["ab"] = {"error": repr(e)}
    if args.set:
        out["config"]["overrides"] = args.set
    if args.layers:
        out["INVALID"] = "debug run with %d layers" % args.layers
    if rank == 0 and world == 1 and args.workload == "c2":
        if args.cross_check:
            out["roofline"]["extra"]["private_loop_ms_per_step"] = private_loop_crosscheck(wl, args.steps)
        out["roofline"]["extra"]["filters"] = filter_microbench(dev)
        out["roofline"]["extra"]["vae_decode"] = vae_decode_microbench(dev, wl.latents0,
                                                                       elapsed / args.steps * wl.steps_per_video)
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_other_workloads and not args.layers:
        import gc
        wl.model = wl.pipe = wl.sched = wl.last_out = None
        gc.collect()
        torch.cuda.empty_cache()
        out["roofline"]["extra"]["workloads"] = other_workloads(args, dev, parallel)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(with_c1=not args.no_c1, c1_budget=args.c1_budget)
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
