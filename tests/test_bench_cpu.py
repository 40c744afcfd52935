"""Host-side pieces of bench.py that can be checked without a GPU: the --set coercion ('pair_qkv=false' used to be
stored as a truthy string), the power / clock sampler degrading to "no reading" instead of raising, the A/B arm table."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_set_values_are_coerced_to_the_attributes_type_or_refused():
    assert bench.coerce_like(True, "false", "pair_qkv") is False and bench.coerce_like(False, "ON", "x") is True
    assert bench.coerce_like(True, "0", "x") is False and bench.coerce_like(False, "1", "x") is True
    assert bench.coerce_like(3, "0x10", "x") == 16 and bench.coerce_like(3, "-2", "x") == -2
    assert bench.coerce_like(1.5, "2.5", "x") == 2.5
    assert bench.coerce_like("a", "b", "x") == "b" and bench.coerce_like(None, "b", "x") == "b"
    for cur, bad in ((True, "maybe"), (3, "three"), (1.0, "fast"), ([], "1")):
        with pytest.raises(SystemExit):
            bench.coerce_like(cur, bad, "x")


def test_the_sampler_never_raises_and_reports_nothing_when_there_is_nothing_to_read():
    s = bench.SmiSampler(0, period=0.01)
    with s:
        pass
    out = s.summary()
    assert out is None or (set(out) == {"power_w", "sclk_mhz", "power_cap_w", "source"})
    if out is not None and out["power_w"] is not None:      # on a GPU box: plausible numbers
        assert 10.0 < out["power_w"]["mean"] < 2000.0


def test_ab_arm_table_is_well_formed():
    names = [a[0] for a in bench.AB_ARMS]
    assert len(names) == len(set(names)) and "events" in names
    for name, kind, key, off, what in bench.AB_ARMS:
        assert kind in ("env", "attr", "events") and isinstance(what, str) and len(what) > 20
        if kind == "env":
            assert key.startswith("ALG_") and isinstance(off, str) and off.lstrip("-").isdigit()


def test_workload_table_matches_baseline_json():
    import json
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert set(bench.WORKLOADS) == {"c2", "c3", "c4", "c5"}
    c2 = bench.WORKLOADS["c2"]
    assert c2.frames == 49 and c2.steps_per_video == 50 and "CogVideoX-5B-I2V" in c2.metric and "CogVideoX-5B-I2V" in base["metric"]
    assert (bench.WORKLOADS["c3"].steps_per_video, bench.WORKLOADS["c4"].frames, bench.WORKLOADS["c5"].fp8) == (40, 129, True)


def test_whole_video_projection_counts_the_schedules_passes():
    """C3: linear decay to 0 at half of 40 steps = 20 three-pass + 20 two-pass steps; C5: interval [0, 0.2] of 50 steps = 10 + 40; C4 runs
    its single-pass branch throughout"""
    class W:
        pass
    want = {"c3": 100, "c5": 110, "c4": 50}
    for name, total in want.items():
        cls = bench.WORKLOADS[name]
        w = W()
        w.steps_per_video, w.frames = cls.steps_per_video, cls.frames
        if hasattr(cls, "alg"):
            w.alg = cls.alg
        out = bench.whole_video_projection(w, elapsed=6.0, forwards=6)
        assert out["sample_forwards_per_video"] == total
        assert abs(out["frames_per_s"] - cls.frames / total) < 1e-12        # one second per sample-forward
        # with a MEASURED 2-pass step (round 6): n3 steps at the leg's step time (elapsed / 2) + n2 steps at the measured time
        out = bench.whole_video_projection(w, elapsed=6.0, forwards=6, two_pass_seconds=2.5)
        n3, n2, n1 = out["three_pass_steps"], out["two_pass_steps"], out["one_pass_steps"]
        assert (n3, n2, n1) == {"c3": (20, 20, 0), "c5": (10, 40, 0), "c4": (0, 0, 50)}[name]
        if n2:                                                               # (c4 has no 2-pass step: the per-forward formula stands)
            assert abs(out["seconds_per_video"] - (n3 * 3.0 + n2 * 2.5)) < 1e-9 and "MEASURED" in out["what"]
