"""Multi-process (gloo, world_size 2) tests of the data-parallel plumbing, and the YAML schema <-> __call__ contract."""
import inspect
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

from alg_amd import CogVideoXImageToVideoPipeline, parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    shapes = {"a.weight": (7, 5), "b.bias": (11,), "c.weight": (3, 2, 2, 2), "d.weight": (1000, 33)}

    def make():
        g = torch.Generator().manual_seed(123)
        return {k: torch.randn(s, generator=g).to(torch.bfloat16) for k, s in shapes.items()}

    # tiny bucket size forces several buckets, like the 11 GB broadcast does with 1 GiB buckets
    sd = parallel.broadcast_state_dict(make, shapes, "cpu", src=0, bucket_bytes=4096)
    ref = make()
    ok = all(torch.equal(sd[k], ref[k]) and tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    mine = parallel.shard_videos(5, rank, world)
    t = parallel.max_over_ranks(1.0 + rank, "cpu")
    parallel.barrier()
    out.put((rank, ok, mine, t))
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, True, [0, 2, 4], 2.0)
    assert res[1] == (1, True, [1, 3], 2.0)


def test_single_process_is_a_noop():
    assert parallel.shard_videos(3, 0, 1) == [0, 1, 2]
    assert parallel.max_over_ranks(3.5, "cpu") == 3.5
    sd = parallel.broadcast_state_dict(lambda: {"w": torch.ones(2)}, {"w": (2,)}, "cpu")
    assert torch.equal(sd["w"], torch.ones(2))


def test_yaml_schema_matches_call_signature():
    """Every generation/alg key of every shipped CogVideoX YAML must be a __call__ keyword (run.py passes them
    verbatim, reference run.py:102-106); section names follow the reference schema."""
    params = set(inspect.signature(CogVideoXImageToVideoPipeline.__call__).parameters)
    for name in os.listdir(os.path.join(ROOT, "configs")):
        with open(os.path.join(ROOT, "configs", name)) as f:
            cfg = yaml.safe_load(f)
        assert set(cfg) <= {"model", "generation", "alg", "video"} and {"model", "generation", "video"} <= set(cfg)
        assert {"path", "dtype"} <= set(cfg["model"]) and "fps" in cfg["video"]
        if name.startswith("cogvideox"):
            for key in {**cfg.get("generation", {}), **cfg.get("alg", {})}:
                assert key in params, (name, key)
    # reference defaults of the ALG block (cog:753-773)
    sig = inspect.signature(CogVideoXImageToVideoPipeline.__call__).parameters
    expect = dict(num_frames=49, num_inference_steps=50, guidance_scale=6.0, use_low_pass_guidance=False,
                  lp_filter_type="none", lp_filter_in_latent=False, lp_blur_sigma=15.0,
                  lp_blur_kernel_size=0.02734375, lp_resize_factor=0.25, lp_strength_schedule_type="none",
                  schedule_blur_kernel_size=False, schedule_interval_start_time=0.0, schedule_interval_end_time=0.05,
                  schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0, schedule_linear_end_time=0.5,
                  schedule_exp_decay_rate=10.0, max_sequence_length=226, output_type="pil", eta=0.0)
    for k, v in expect.items():
        assert sig[k].default == v, k


def _cfg_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    split = parallel.CFGPairSplit.from_world()
    ok = True
    for n_pass, batch in ((2, 1), (3, 1), (3, 2)):
        g = torch.Generator().manual_seed(7 + n_pass + batch)
        full = torch.randn(n_pass * batch, 4, 3, 5, generator=g).to(torch.bfloat16)   # what one GPU would compute
        mine = split.my_passes(n_pass)
        local = torch.cat([full[p * batch:(p + 1) * batch] for p in mine])
        merged = split.merge(local, n_pass, batch)
        ok = ok and torch.equal(merged, full)
    out.put((rank, split.pair_rank, split.my_passes(2), split.my_passes(3), ok))
    dist.destroy_process_group()


def test_cfg_pair_split_exchange():
    """The cond / uncond branches of one video on two ranks: pass assignment covers every pass exactly once and the
    one-collective merge reproduces the full prediction bit-exactly on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, 0, [0], [0, 1], True)
    assert res[1] == (1, 1, [1], [2], True)


def test_yaml_schema_matches_wan_call_signature():
    from alg_amd import WanImageToVideoPipeline
    params = set(inspect.signature(WanImageToVideoPipeline.__call__).parameters)
    for name in os.listdir(os.path.join(ROOT, "configs")):
        if name.startswith("wan"):
            with open(os.path.join(ROOT, "configs", name)) as f:
                cfg = yaml.safe_load(f)
            for key in {**cfg.get("generation", {}), **cfg.get("alg", {})}:
                assert key in params, (name, key)


# ---- the product launcher: run.py --jobs ... --gpus N (video v -> rank v mod world, one weight broadcast) -------
class _ToyPipe:
    """A CPU stand-in with the pipelines' call protocol: its output is a deterministic function of the (broadcast)
    weights, the job's generator stream and the job's synthetic inputs -- everything the launcher is responsible for."""
    vae = None

    def __init__(self, w):
        self.w = w

    def to(self, device):
        return self

    def __call__(self, generator=None, prompt_embeds=None, image_latents=None, num_inference_steps=2, **kw):
        from types import SimpleNamespace
        x = torch.randn(4, 8, generator=generator)
        for _ in range(num_inference_steps):
            x = torch.tanh(x @ self.w["a"].float()) + self.w["b"].float()
        x = x + prompt_embeds.float().mean() + image_latents.float().std()
        return SimpleNamespace(frames=x)


def _toy_build(config, args, device):
    sd = None
    if not dist.is_initialized() or dist.get_rank() == 0:   # only rank 0 "reads the checkpoint"
        g = torch.Generator().manual_seed(99)
        sd = {"a": torch.randn(8, 8, generator=g).to(torch.bfloat16), "b": torch.randn(8, generator=g)}
    return _ToyPipe(parallel.broadcast_loaded_state_dict(sd, torch.device("cpu")))


def _run_worker(rank, world, port, argv):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import run
    run.build_pipeline = _toy_build
    run.main(run.make_parser().parse_args(argv))


def test_run_py_data_parallel_jobs_match_single_process_runs(tmp_path):
    """`run.py --jobs` on 2 ranks (gloo): weights exist on rank 0 only and arrive by broadcast, video v runs on rank v mod 2
    with its own seed, and every output equals the same job run alone in one process, bit for bit."""
    import sys
    sys.path.insert(0, ROOT)
    import run
    cfg = tmp_path / "c.yaml"
    cfg.write_text(yaml.safe_dump({"model": {"path": "CogVideoX-toy", "dtype": "bfloat16",
                                             "synthetic_config": {"text_embed_dim": 16, "max_text_seq_length": 4,
                                                                  "sample_height": 4, "sample_width": 4, "in_channels": 8}},
                                   "generation": {"num_inference_steps": 3}, "alg": {}, "video": {"fps": 8}}))
    jobs = [{"output_path": str(tmp_path / ("dp_%d.pt" % v))} for v in range(5)]
    jobs[3]["seed"] = 7
    jf = tmp_path / "jobs.yaml"
    jf.write_text(yaml.safe_dump(jobs))
    argv = ["--config", str(cfg), "--synthetic", "--jobs", str(jf), "--generator_device", "cpu"]
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_run_worker, args=(r, 2, port, argv)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    # the same jobs, one at a time, in this process (world size 1, no process group)
    args = run.make_parser().parse_args(argv)
    config = yaml.safe_load(cfg.read_text())
    pipe = _toy_build(config, args, torch.device("cpu"))
    loaded = run.load_jobs(args)
    assert [j["seed"] for j in loaded] == [42, 43, 44, 7, 46]
    outs = []
    for v, job in enumerate(loaded):
        solo = dict(job, output_path=str(tmp_path / ("solo_%d.pt" % v)))
        run.run_job(pipe, config, args, solo)
        a, b = torch.load(solo["output_path"]), torch.load(job["output_path"])
        assert torch.equal(a, b), v
        outs.append(a)
    assert not torch.equal(outs[0], outs[1])     # different seeds -> different videos


def _cfg_signed_zero_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    split = parallel.CFGPairSplit.from_world()
    full = torch.tensor([[-0.0, 1.0], [0.0, -2.0], [-0.0, -0.0]]).to(torch.bfloat16)
    mine = split.my_passes(3)
    merged = split.merge(torch.stack([full[p] for p in mine]), 3, 1)
    out.put((rank, torch.equal(merged.view(torch.int16), full.view(torch.int16))))
    dist.destroy_process_group()


def test_cfg_pair_merge_moves_bits_not_sums():
    """The pair exchange is an all-gather of rows: signed zeros survive (an all-reduce of a zero-padded buffer gives
    -0 + 0 = +0), and the single-pass rank's pad row is never read."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg_signed_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _loaded_worker(rank, world, port, out, fail):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    g = torch.Generator().manual_seed(9)
    full = {"w": torch.randn(700, 800, generator=g).to(torch.bfloat16), "bias": torch.randn(33, generator=g),
            "empty": torch.zeros(0, 4), "ids": torch.arange(5)}
    sd = (FileNotFoundError("shard missing") if fail else full) if rank == 0 else None
    try:
        got = parallel.broadcast_loaded_state_dict(sd, "cpu", bucket_bytes=1 << 16)
        ok = all(torch.equal(got[k], full[k]) and got[k].dtype == full[k].dtype for k in full)
        # a small tensor owns its storage (it does not pin the bucket it travelled in); the big one may be a view
        small_own = got["bias"].untyped_storage().nbytes() <= 33 * 4 + 16
        out.put((rank, "ok", ok and small_own and list(got) == list(full)))
    except RuntimeError as e:
        out.put((rank, "raised", "shard missing" in str(e)))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_loaded_state_dict_broadcast_status_small_tensors_and_empty_buckets(fail):
    """rank 0 failing to read its shards makes EVERY rank raise (no rank is left in a broadcast); zero-element
    tensors produce no 0-byte collective; small tensors are copied out of their bucket."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loaded_worker, args=(r, 2, port, q, fail)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, "raised" if fail else "ok", True), (1, "raised" if fail else "ok", True)]


def _world8_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    res = {"rank": rank}
    # (1) rank 0 fails to read its shards: EVERY one of the eight ranks raises, nobody is left inside a broadcast
    try:
        parallel.broadcast_loaded_state_dict(FileNotFoundError("shard missing") if rank == 0 else None, "cpu")
        res["load_failure"] = "no error"
    except RuntimeError as e:
        res["load_failure"] = "shard missing" in str(e)
    # (2) the weight broadcast proper, several buckets, and its statistics
    g = torch.Generator().manual_seed(9)
    full = {"w": torch.randn(300, 200, generator=g).to(torch.bfloat16), "b": torch.randn(77, generator=g)}
    got = parallel.broadcast_loaded_state_dict(full if rank == 0 else None, "cpu", bucket_bytes=1 << 14)
    res["bcast_ok"] = all(torch.equal(got[k], full[k]) for k in full)
    res["bcast_stats"] = (parallel.BCAST_STATS["collectives"] >= 2, parallel.BCAST_STATS["bytes"] >= 300 * 200 * 2 + 77 * 4,
                          parallel.BCAST_STATS["seconds"] > 0)
    # (3) 11 jobs on 8 ranks (jobs % world != 0): v -> rank v mod world, every job exactly once, ranks 3.. take one job only
    res["jobs"] = parallel.shard_videos(11, rank, world)
    # (4) four CFG pairs (BASELINE config 5: 2 GPUs x 4 videos): pair groups (0,1) (2,3) (4,5) (6,7), each merging ITS video
    split = parallel.CFGPairSplit.from_world()
    ok = True
    for n_pass in (2, 3):
        gg = torch.Generator().manual_seed(100 * (rank // 2) + n_pass)           # one prediction per PAIR (= per video)
        pred = torch.randn(n_pass, 4, 6, generator=gg).to(torch.bfloat16)
        mine = split.my_passes(n_pass)
        merged = split.merge(torch.cat([pred[p:p + 1] for p in mine]), n_pass, 1)
        ok = ok and torch.equal(merged, pred)
    res["cfg_pairs"] = (split.pair_rank, ok)
    # (5) who is here: eight distinct processes report in (on a GPU node: eight distinct device UUIDs)
    seen = parallel.ranks_seen("cpu")
    res["seen"] = (len(seen["ranks"]), seen["distinct_gpus"], sorted(r["rank"] for r in seen["ranks"]))
    res["tmax"] = parallel.max_over_ranks(float(rank), "cpu")
    parallel.barrier()
    out.put(res)
    dist.destroy_process_group()


def test_world_size_8_pairs_ragged_jobs_and_load_failure():
    """the N = 8 layout of the node the metric is quoted on, on gloo: a failed checkpoint read on rank 0 reaches
    all eight ranks as an exception; the bucketed broadcast delivers identical bits and counts its bytes / seconds; 11 jobs over
    8 ranks are each run exactly once; the four CFG pairs exchange within their own pair only; ranks_seen lists all eight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r["load_failure"] for r in res] == [True] * 8
    assert all(r["bcast_ok"] and r["bcast_stats"] == (True, True, True) for r in res)
    jobs = [v for r in res for v in r["jobs"]]
    assert sorted(jobs) == list(range(11)) and [len(r["jobs"]) for r in res] == [2, 2, 2, 1, 1, 1, 1, 1]
    assert [r["cfg_pairs"] for r in res] == [(i % 2, True) for i in range(8)]
    assert all(r["seen"] == (8, 8, list(range(8))) for r in res)
    assert all(r["tmax"] == 7.0 for r in res)


def test_process_group_timeout_outlasts_a_video():
    assert parallel.DIST_TIMEOUT_S >= 3600      # C4 / C5: 9-13 minutes per video; ranks with fewer jobs wait in the barrier


def _forced_world1_worker(port, out):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert parallel.init_distributed(backend="gloo") == (0, 0, 1) and not dist.is_initialized()   # unforced: still a no-op
    parallel.init_distributed(backend="gloo", force=True)
    g = torch.Generator().manual_seed(5)
    sd = {"a": torch.randn(40, 9, generator=g).to(torch.bfloat16), "b": torch.randint(0, 9, (5,), generator=g),
          "c": torch.randn(300, generator=g).to(torch.float8_e4m3fn)}
    n0 = parallel.BCAST_STATS["collectives"]
    same = parallel.broadcast_loaded_state_dict(sd, "cpu") is sd and parallel.BCAST_STATS["collectives"] == n0
    got = parallel.broadcast_loaded_state_dict(sd, "cpu", bucket_bytes=256, force=True)
    bits = all(torch.equal(got[k].view(torch.uint8), sd[k].view(torch.uint8)) and got[k] is not sd[k] for k in sd)
    seen = parallel.ranks_seen(None, force=True)
    parallel.barrier(force=True)
    out.put((same, bits, parallel.BCAST_STATS["collectives"] - n0, parallel.max_over_ranks(2.5, "cpu", force=True), len(seen["ranks"])))
    dist.destroy_process_group()


def test_forced_one_rank_group_runs_the_collectives():
    """the one-GPU rehearsal of the RCCL path (tests/test_gpu_rccl_world1.py) rests on this switch: force=True takes the
    world-size-1 short-circuits away, the default keeps them"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1_worker, args=(_free_port(), q))
    p.start()
    same, bits, n_coll, mx, n_ranks = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and bits and n_coll >= 3 and mx == 2.5 and n_ranks == 1
