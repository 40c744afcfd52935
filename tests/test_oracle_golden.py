"""The CPU oracle against the golden vectors generated from the reference's own lp_utils
(tests/golden/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import lp_oracle


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _strength(ps, i, total):
    return lp_oracle.get_lp_strength(
        i, total, ps["kind"], ps.get("start", 0.0), ps.get("end", 0.05), ps.get("w0", 1.0), ps.get("w1", 0.0),
        ps.get("t1", 0.5), ps.get("rate", 10.0))


def test_schedule_bit_exact(golden_dir):
    cases = _load(golden_dir, "schedule_tables.json")
    assert len(cases) >= 60
    for c in cases:
        for i, hx in enumerate(c["strength_hex"]):
            got = _strength(c["params"], i, c["total_steps"])
            assert float(got).hex() == hx, (c["params"], c["total_steps"], i)


def test_active_step_tables(golden_dir):
    """Derived integer tables quoted in SURVEY.md 8a-3 (which steps run three DiT passes)."""
    def active(ps, total):
        return [i for i in range(total) if not lp_oracle.two_pass_flag(_strength(ps, i, total), ps["kind"])]
    assert active(dict(kind="interval", start=0.0, end=0.04), 50) == [0, 1]
    assert active(dict(kind="interval", start=0.0, end=0.20), 50) == list(range(10))
    assert active(dict(kind="interval", start=0.0, end=0.20), 40) == list(range(8))
    assert active(dict(kind="interval", start=0.0, end=0.04), 2) == [0]
    assert active(dict(kind="linear", w0=1.0, w1=0.0, t1=0.5), 40) == list(range(20))
    assert len(active(dict(kind="exponential", rate=10.0), 50)) == 12
    assert _strength(dict(kind="interval", start=0.0, end=0.04), 0, 1) == 1.0


def test_down_up_against_reference_vectors(golden_dir):
    misc = _load(golden_dir, "lp_misc.json")
    vec = np.load(os.path.join(golden_dir, "down_up_vectors.npz"))
    assert len(misc["down_up_meta"]) >= 10
    for m in misc["down_up_meta"]:
        x, ref = vec[m["name"] + "_in"], vec[m["name"] + "_out"]
        assert lp_oracle.down_up_size(x.shape[-2], x.shape[-1], m["factor"]) == (m["h1"], m["w1"])
        # float32 tap arithmetic (what ATen does for float tensors): agreement to fp32 rounding
        got32 = lp_oracle.down_up(x, m["factor"], np.float32)
        assert np.abs(got32 - ref).max() <= 1e-6, m["name"]
        # float64 taps differ from ATen's float taps by O(1e-5): documents why the kernels use fp32 taps
        got64 = lp_oracle.down_up(x, m["factor"], np.float64)
        assert np.abs(got64 - ref).max() <= 5e-5, m["name"]


def test_5d_equals_per_plane(golden_dir):
    vec = np.load(os.path.join(golden_dir, "down_up_vectors.npz"))
    x, ref = vec["c1_5d_in"], vec["c1_5d_out"]
    got = lp_oracle.apply_low_pass_filter(x, "down_up", 0.0, 0, 0.25, ftype=np.float32)
    assert got.shape == x.shape and np.abs(got - ref).max() <= 1e-6


def test_size_table_and_identity(golden_dir):
    misc = _load(golden_dir, "lp_misc.json")
    for s in misc["sizes"]:
        assert lp_oracle.down_up_size(s["h0"], s["w0"], s["factor"]) == (s["h1"], s["w1"])
    assert lp_oracle.down_up_size(60, 90, 0.25) == (15, 22)  # 22.5 -> 22 (banker's rounding)
    assert all(misc["identity"].values())
    x = np.zeros((1, 2, 3, 4, 5), np.float32)
    assert lp_oracle.apply_low_pass_filter(x, "none", 1.0, 3, 0.5) is x
    assert lp_oracle.apply_low_pass_filter(x, "down_up", 1.0, 3, 1.0) is x
    assert lp_oracle.apply_low_pass_filter(x, "gaussian_blur", 0, 3, 0.5) is x


def test_hunyuan_buckets(golden_dir):
    misc = _load(golden_dir, "lp_misc.json")
    for b in misc["hunyuan_buckets"]:
        assert lp_oracle.get_hunyuan_video_size(b["resolution"], tuple(b["image_wh"])) == (b["height"], b["width"])
    assert [list(p) for p in lp_oracle.generate_crop_size_list(480, 32)] == misc["crop_size_list_480_32"]


def test_gaussian_matches_independent_formulation():
    """torchvision is absent -> no reference vector; cross-check the restatement against scipy's mirror
    correlation (an independent implementation of reflect-pad + separable correlation)."""
    from scipy import ndimage

    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 60, 104))
    for k, sigma in ((9, 15.0), (3, 0.7), (13, 4.2)):
        g = lp_oracle.gaussian_kernel1d(k, sigma)
        assert abs(g.sum() - 1) < 1e-12 and np.allclose(g, g[::-1])
        ref = ndimage.correlate1d(ndimage.correlate1d(x, g, axis=-1, mode="mirror"), g, axis=-2, mode="mirror")
        assert np.abs(lp_oracle.gaussian_blur(x, k, sigma) - ref).max() < 1e-13
    # the threaded torch statement bench.py times as the CPU baseline (torchvision's own op order: k x k kernel, one depthwise
    # conv2d) agrees with the separable numpy one to fp32 rounding
    import torch
    from oracle import loop_oracle
    xt = torch.from_numpy(x.astype(np.float32))
    for k, sigma in ((9, 15.0), (3, 0.7), (13, 4.2)):
        got = loop_oracle.gaussian_blur_torch(xt, k, sigma).numpy()
        assert np.abs(got - lp_oracle.gaussian_blur(x, k, sigma)).max() < 2e-6
    # kernel-size rule (lp_utils.py:41-46): float = fraction of H, int = absolute, even -> +1
    assert lp_oracle.gaussian_kernel_size(0.02734375, 60) == 1
    assert lp_oracle.gaussian_kernel_size(0.02734375, 480) == 13
    assert lp_oracle.gaussian_kernel_size(0.02734375, 720) == 19
    assert lp_oracle.gaussian_kernel_size(8, 60) == 9
    assert lp_oracle.gaussian_kernel_size(15 * 1.0, 60) == 901  # quirk a-Q3: int*float -> float -> fraction of H
    with pytest.raises(ValueError):
        lp_oracle.gaussian_blur(x, 901, 1.0)


def test_bf16_round_helper():
    import torch

    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 37
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(lp_oracle.bf16_round(x), ref)
