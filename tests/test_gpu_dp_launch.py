"""Keeps the multi-GPU launch path warm on a ONE-GPU box (BASELINE.json configs 4 - 5):
the exact command line the driver uses for its scaling runs -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- with two ranks sharing device 0 (ALG_DIST_ONE_GPU=1) over
gloo (ALG_DIST_BACKEND=gloo: RCCL refuses two ranks on one device).  Checked: the JSON line of an N = 2 run (two ranks seen, the
weight broadcast happened and moved bytes, weak scaling), and that data parallelism changes no bit -- rank r of the 2-rank run
ends on exactly the latents a single-process run with that rank's seed ends on (no per-step collective exists to change them)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "3", "--layers", "1", "--no-cpu-baseline", "--no-calibration", "--no-other-workloads", "--no-ab"]


def _run(cmd, env, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-1500:]
    return json.loads(lines[-1])


def test_two_ranks_through_the_drivers_launch_line(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ALG_DIST_BACKEND="gloo", ALG_DIST_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--dump-latents", str(tmp_path / "dp"), *COMMON], env)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["finite"] and two["steps"] == 2
    assert two["config"]["parallelism"] == "dp2" and two["config"]["videos"] == pytest.approx(2 * 2 / 50)
    assert len(two["ranks_seen"]) == 2 and {r["rank"] for r in two["ranks_seen"]} == {0, 1}
    assert two["bcast_gbytes"] > 0 and two["bcast_collectives"] >= 1 and two["bcast_seconds"] > 0
    assert two["distinct_gpus"] == 1                      # the test hook: both ranks on device 0 (the driver's node has N)
    assert two["value"] > 0 and two["roofline"]["achieved"] > 0 and "cpu_baseline" not in two
    solo_env = {k: v for k, v in env.items() if k not in ("ALG_DIST_BACKEND", "ALG_DIST_ONE_GPU")}
    for rank in (0, 1):
        one = _run([sys.executable, "bench.py", "--gpus", "1", "--seed-offset", str(rank), "--dump-latents",
                    str(tmp_path / ("solo%d" % rank)), *COMMON], solo_env)
        assert one["n_gpus"] == 1 and "ranks_seen" not in one
        a = torch.load(tmp_path / ("dp.rank%d.pt" % rank))
        b = torch.load(tmp_path / ("solo%d.rank0.pt" % rank))
        assert a.shape == (1, 13, 16, 60, 90) and torch.equal(a, b), rank
    assert not torch.equal(torch.load(tmp_path / "dp.rank0.pt"), torch.load(tmp_path / "dp.rank1.pt"))   # different videos
