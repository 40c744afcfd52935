#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own code.

Run only in the build container (needs /root/reference); the fixtures it writes are
data (inputs + expected outputs) and are committed; nothing of the reference travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported: /root/reference/lp_utils.py.  It imports
``torchvision.transforms.functional`` at module top (lp_utils.py:4); torchvision is not
installed here, so an empty stub module is registered for the import only.  That makes
``get_lp_strength``, ``apply_low_pass_filter('none'|'down_up')`` and the Hunyuan bucket
helpers executable against this container's torch (ATen CPU ``_upsample_bilinear2d_aa``);
``gaussian_blur`` cannot execute (needs the real torchvision) and has no fixture.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference_lp_utils():
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.path.insert(0, REF)
    import lp_utils  # noqa: E402

    sys.path.pop(0)
    return lp_utils


def main():
    import numpy as np
    import torch

    torch.set_num_threads(1)
    lp = import_reference_lp_utils()

    # ---- 1. schedule tables -------------------------------------------------------------
    sched_cases = []
    param_sets = [
        # shipped YAML values
        dict(kind="interval", start=0.0, end=0.04),  # cogvideox_alg / hunyuan_video_alg
        dict(kind="interval", start=0.0, end=0.20),  # wan_alg
        dict(kind="interval", start=0.0, end=0.05),  # __call__ default (cog:765)
        dict(kind="interval", start=0.1, end=0.3),
        dict(kind="linear", w0=1.0, w1=0.0, t1=0.5),  # defaults cog:768-770
        dict(kind="linear", w0=0.8, w1=0.2, t1=0.25),
        dict(kind="linear", w0=1.0, w1=0.0, t1=0.0),  # end_time <= 0 -> start weight
        dict(kind="linear", w0=1.0, w1=0.0, t1=-1.0),
        dict(kind="exponential", rate=10.0),  # default cog:773
        dict(kind="exponential", rate=-3.0),  # negative -> abs (+ warning)
        dict(kind="none"),
        dict(kind="bogus"),  # unknown -> 1.0 (+ warning)
    ]
    import contextlib
    import io

    for ps in param_sets:
        for total in (1, 2, 3, 10, 40, 50):
            vals = []
            for i in range(total):
                with contextlib.redirect_stdout(io.StringIO()):
                    v = lp.get_lp_strength(
                        step_index=i,
                        total_steps=total,
                        lp_strength_schedule_type=ps["kind"],
                        schedule_interval_start_time=ps.get("start", 0.0),
                        schedule_interval_end_time=ps.get("end", 0.05),
                        schedule_linear_start_weight=ps.get("w0", 1.0),
                        schedule_linear_end_weight=ps.get("w1", 0.0),
                        schedule_linear_end_time=ps.get("t1", 0.5),
                        schedule_exp_decay_rate=ps.get("rate", 10.0),
                    )
                vals.append(float(v).hex())
            sched_cases.append(dict(params=ps, total_steps=total, strength_hex=vals))
    with open(os.path.join(HERE, "schedule_tables.json"), "w") as f:
        json.dump(sched_cases, f, indent=0)

    # ---- 2. down_up tensors (ATen CPU fp32 through the reference function) ---------------
    rng = np.random.default_rng(20250929)
    tensors = {}
    meta = []

    def run_case(name, shape, factor):
        x = rng.standard_normal(shape).astype(np.float32)
        xt = torch.from_numpy(x.copy())
        y = lp.apply_low_pass_filter(xt, "down_up", 0.0, 0, factor)
        assert y is not xt
        h0, w0 = shape[-2:]
        h1 = max(1, int(round(h0 * factor)))
        w1 = max(1, int(round(w0 * factor)))
        tensors[name + "_in"] = x
        tensors[name + "_out"] = y.numpy().astype(np.float32)
        meta.append(dict(name=name, shape=list(shape), factor=factor, h1=h1, w1=w1))

    run_case("c1_5d", (1, 16, 3, 32, 32), 0.25)  # BASELINE config 1 latent shape
    for f in (0.25, 0.4, 0.625, 0.9):
        run_case("p60x90_f%s" % str(f).replace(".", "p"), (1, 2, 2, 60, 90), f)
    run_case("p60x104_f0p4", (1, 3, 1, 60, 104), 0.4)  # Wan-480p plane
    run_case("p90x160_f0p625", (1, 2, 1, 90, 160), 0.625)  # Hunyuan-720p plane
    run_case("deg_7x5_f0p1", (2, 3, 7, 5), 0.1)  # -> (1, 1)
    run_case("ragged_4d_13x17_f0p5", (2, 3, 13, 17), 0.5)
    # schedule-modulated factors (1 - (1-f)*s) for a linear schedule
    for s in (0.9487179487179487, 0.5128205128205128, 0.02564102564102566):
        f_eff = 1.0 - (1.0 - 0.25) * s
        run_case("p60x90_s%.3f" % s, (1, 1, 2, 60, 90), f_eff)
    np.savez_compressed(os.path.join(HERE, "down_up_vectors.npz"), **tensors)

    # ---- 3. identity exits (lp:23-28) ------------------------------------------------------
    xt = torch.zeros(1, 2, 3, 4, 5)
    ident = dict(
        none=lp.apply_low_pass_filter(xt, "none", 1.0, 3, 0.5) is xt,
        down_up_factor_1=lp.apply_low_pass_filter(xt, "down_up", 1.0, 3, 1.0) is xt,
        gaussian_sigma_0=lp.apply_low_pass_filter(xt, "gaussian_blur", 0, 3, 0.5) is xt,
    )
    # non-contiguous 5-D input raises in .view (lp:35)
    try:
        lp.apply_low_pass_filter(torch.zeros(1, 3, 2, 4, 5).permute(0, 2, 1, 3, 4), "down_up", 0.0, 0, 0.5)
        ident["noncontig_raises"] = False
    except RuntimeError:
        ident["noncontig_raises"] = True

    # ---- 4. size table (banker's rounding) -------------------------------------------------
    sizes = []
    for (h0, w0) in ((60, 90), (60, 104), (90, 160), (32, 32), (44, 78), (7, 5)):
        for f in (0.25, 0.4, 0.625, 0.1, 0.5, 0.9, 0.2884615384615385):
            y = lp.apply_low_pass_filter(torch.zeros(1, 1, h0, w0), "down_up", 0.0, 0, f)
            assert tuple(y.shape[-2:]) == (h0, w0)
            sizes.append(dict(h0=h0, w0=w0, factor=f, h1=max(1, int(round(h0 * f))), w1=max(1, int(round(w0 * f)))))

    # ---- 5. Hunyuan buckets ---------------------------------------------------------------
    class _Img:  # get_hunyuan_video_size only reads ``.size`` (lp_utils.py:182)
        def __init__(self, w, h):
            self.size = (w, h)

    buckets = []
    for (w, h) in ((832, 480), (887, 512), (1280, 720), (720, 1280), (512, 512)):
        for res in ("360p", "540p", "720p"):
            th, tw = lp.get_hunyuan_video_size(res, _Img(w, h))
            buckets.append(dict(image_wh=[w, h], resolution=res, height=int(th), width=int(tw)))
    crop_list = [list(map(int, p)) for p in lp._generate_crop_size_list(480, 32)]

    with open(os.path.join(HERE, "lp_misc.json"), "w") as f:
        json.dump(dict(down_up_meta=meta, identity=ident, sizes=sizes, hunyuan_buckets=buckets,
                       crop_size_list_480_32=crop_list, torch_version=torch.__version__), f, indent=0)
    print("wrote fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
