"""The RCCL backend, executed on the ONE GPU a gpurun box has (BASELINE.json configs 4 - 5).

Every multi-process test of the data-parallel path runs gloo (two ranks cannot share a device under RCCL), so until this file
`backend="nccl"` -- the weight broadcast buckets, `all_gather_object`, the MAX all-reduce, the barrier, the CFG pair's
all-gather -- had never been executed anywhere: the driver's first 8-GPU run would have been the first time `librccl` was
loaded.  Here a ONE-rank RCCL group runs the very same helpers with their world-size-1 short-circuits forced off
(`alg_amd/parallel.py: force=True / ALG_DIST_FORCE=1`): the communicator is created, every collective is enqueued on the device
by RCCL kernels and completes, every dtype the product broadcasts (bf16, fp32, e4m3 viewed as bytes, int64, uint8) comes back
bit-identical, and the bench line of such a run carries `rccl_version`.  The reference has nothing to match here
(`/root/reference/run.py:40` picks one device); this is the build's own start-up collective."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_group():
    from alg_amd import parallel
    assert not dist.is_initialized()
    saved = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    try:
        rank, local, world = parallel.init_distributed(backend="nccl", force=True)
        assert (rank, local, world) == (0, 0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
        yield parallel
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _librccl_mapped():
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if "librccl" in ln or "libnccl" in ln})


def test_loaded_state_dict_buckets_every_dtype_through_rccl(rccl_group):
    parallel = rccl_group
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7)
    sd = {
        "w.bf16": torch.randn(300, 257, generator=g).to(torch.bfloat16),
        "w.f32": torch.randn(129, 65, generator=g),
        "w.fp8": torch.randn(64, 200, generator=g).to(torch.float8_e4m3fn),
        "scale.f32": torch.rand(64, generator=g),
        "ids.i64": torch.randint(-2 ** 40, 2 ** 40, (33,), generator=g),
        "bytes.u8": torch.randint(0, 255, (1001,), generator=g, dtype=torch.uint8),
        "empty.bf16": torch.empty(0, 8, dtype=torch.bfloat16),
        "big.bf16": torch.randn(1 << 20, generator=g).to(torch.bfloat16),        # 2 MiB: stays a view of its bucket
    }
    before = dict(parallel.BCAST_STATS)
    out = parallel.broadcast_loaded_state_dict({k: v.clone() for k, v in sd.items()}, dev, bucket_bytes=1 << 20, force=True)
    assert list(out) == list(sd)
    for k, v in sd.items():
        assert out[k].device == dev and out[k].dtype == v.dtype and out[k].shape == v.shape, k
        assert torch.equal(out[k].cpu().view(torch.uint8), v.contiguous().view(torch.uint8)), k    # bits, not values
    # several buckets per dtype at a 1 MiB bucket size, each ONE RCCL broadcast of uint8
    assert parallel.BCAST_STATS["collectives"] - before["collectives"] >= 6
    assert parallel.BCAST_STATS["bytes"] - before["bytes"] >= sum(v.numel() * v.element_size() for v in sd.values())
    assert parallel.BCAST_STATS["seconds"] > before["seconds"]
    # the short-circuit is what an unforced world-1 call takes: the same object back, no collective
    n = parallel.BCAST_STATS["collectives"]
    assert parallel.broadcast_loaded_state_dict(sd, dev) is sd and parallel.BCAST_STATS["collectives"] == n
    assert _librccl_mapped(), "the nccl backend ran without librccl in the process map"


def test_synthetic_state_dict_and_failure_status_through_rccl(rccl_group):
    parallel = rccl_group
    dev = torch.device("cuda", 0)
    shapes = {"a": (100, 33), "b": (7,), "c": (64, 64, 3)}
    make = lambda: {k: torch.full(s, float(i + 1)) for i, (k, s) in enumerate(shapes.items())}
    out = parallel.broadcast_state_dict(make, shapes, dev, bucket_bytes=4096, force=True)
    for i, (k, s) in enumerate(shapes.items()):
        assert out[k].shape == s and out[k].dtype == torch.bfloat16 and bool((out[k] == i + 1).all()), k
    # a failed read on the source rank travels as a status object (broadcast_object_list over RCCL) and raises everywhere
    with pytest.raises(RuntimeError, match="failed to load"):
        parallel.broadcast_loaded_state_dict(OSError("no such checkpoint"), dev, force=True)


def test_ranks_seen_max_barrier_and_pair_gather_through_rccl(rccl_group):
    parallel = rccl_group
    dev = torch.device("cuda", 0)
    seen = parallel.ranks_seen(dev, force=True)                       # all_gather_object on the CUDA-only backend
    assert seen["distinct_gpus"] == 1 and len(seen["ranks"]) == 1
    r = seen["ranks"][0]
    assert r["rank"] == 0 and r["device"] == 0 and r["name"] and (r["uuid"] or r["pci"])
    assert parallel.max_over_ranks(3.25, dev, force=True) == 3.25     # float64 MAX all-reduce
    parallel.barrier(force=True)
    # the CFG pair's only data-path collective, on the default (one-rank) group: a pure copy of the prediction rows
    pred = torch.randn(2, 16, 21, 8, 8, device=dev).to(torch.bfloat16)
    parts = [torch.empty_like(pred)]
    parallel.CFGPairSplit(group=None, pair_rank=0).all_gather(parts, pred)
    torch.cuda.synchronize()
    assert torch.equal(parts[0], pred)
    v = parallel.rccl_version()
    assert v and v[0].isdigit()


def test_bench_line_of_a_forced_one_rank_rccl_run_carries_rccl_version():
    """bench.py end to end on the RCCL backend: the driver's launch line with ONE rank, collectives forced on."""
    env = dict(os.environ, ALG_DIST_FORCE="1", ALG_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "3", "--layers", "1",
           "--no-cpu-baseline", "--no-calibration", "--no-other-workloads", "--no-ab"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["finite"] and line["dist_backend"] == "nccl"
    assert line["rccl_version"] and line["rccl_version"][0].isdigit()
    assert len(line["ranks_seen"]) == 1 and line["distinct_gpus"] == 1
    assert line["bcast_collectives"] >= 1 and line["bcast_gbytes"] > 0     # the synthetic weights went through RCCL buckets
