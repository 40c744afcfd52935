#!/usr/bin/env python3
"""Probe (round 6): does the C2 step run differently with and without the bench's HIP-event brackets, under GEMM schedule 9 / 10?
The in-run A/B of the first schedule-10 bench line showed the un-instrumented default arm 6 % SLOWER than the instrumented one.
    python scripts/probes/events_vs_plain.py [steps] [rounds]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from alg_amd import _lib, parallel  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    ap_args = type("A", (), dict(layers=0, seed_offset=0, steps=steps, warmup=2, gpus=1))()
    wl = bench.WORKLOADS["c2"](ap_args, dev, 0, 1, None)
    wl.build()
    torch.cuda.synchronize()

    def timed(events, warm=3):
        bench.run_steps(wl, warm)
        kinds = {} if events else None
        if events:
            wl.instrument(kinds)
        smi = bench.SmiSampler(0)
        smi.__enter__()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(wl, steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        smi.__exit__(None, None, None)
        if events:
            wl.instrument(None)
        s = smi.summary()
        return dt, s["power_w"]["mean"], s["sclk_mhz"]["mean"]

    for r in range(rounds):
        for pipe in ("10", "9"):
            os.environ["ALG_GEMM_PIPE"] = pipe
            _lib.reload_env()
            for events in (False, True, False):
                dt, pw, clk = timed(events)
                print(json.dumps({"round": r, "gemm_pipe": pipe, "events": events, "ms_per_step": round(dt, 2), "power_w": round(pw, 1),
                                  "sclk_mhz": round(clk, 1)}), flush=True)


if __name__ == "__main__":
    main()
