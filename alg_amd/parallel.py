"""Data-parallel plumbing: one process per GPU, independent videos per rank, ONE weight broadcast over
RCCL/xGMI at start-up and no data-path collective afterwards.

The reference has no distributed code at all; this is the build's own addition.  ``torch.distributed`` backend
"nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as dist

# A rank with fewer videos than its neighbours (jobs % world != 0) waits in the final barrier while the others still render:
# a C4 / C5 video is 9-13 minutes, the default NCCL watchdog aborts a waiting rank after 10.  Collectives here are a handful
# per process lifetime, so a generous limit costs nothing (ALG_DIST_TIMEOUT_S overrides it).
DIST_TIMEOUT_S = 6 * 3600
# tensors below this size are copied out of their broadcast bucket: a bias or norm weight a model keeps must not pin 1 GiB
SMALL_TENSOR_BYTES = 1 << 20


# what the one-time weight broadcast cost (bench.py reports it next to an N > 1 line): seconds are host wall time around each
# bucket's collective with the device drained on both sides -- start-up only, never on the data path
BCAST_STATS = {"seconds": 0.0, "bytes": 0, "collectives": 0}

# Every helper below short-circuits at world size 1 (a single-GPU run must not touch a process group).  FORCE_COLLECTIVES (or
# ALG_DIST_FORCE=1, or force=True on a call) takes the short-circuits away: the collectives then run on a ONE-rank group --
# how a one-GPU box executes the RCCL code path an 8-GPU node will run (tests/test_gpu_rccl_world1.py).
FORCE_COLLECTIVES = os.environ.get("ALG_DIST_FORCE") == "1"


def _collective(force=False):
    """True when the helpers must really call torch.distributed: a group exists and has peers, or the caller forces it"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or force or FORCE_COLLECTIVES)


def rccl_version():
    """Version of the RCCL the "nccl" backend binds ("2.26.6"), or None on a build without it (CPU-only torch)."""
    try:
        v = torch.cuda.nccl.version()
    except Exception:   # noqa: BLE001 -- no RCCL in this torch build
        return None
    return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)


def _timed_broadcast(flat, src):
    import time
    cuda = flat.is_cuda
    if cuda:
        torch.cuda.synchronize(flat.device)
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)
    if cuda:
        torch.cuda.synchronize(flat.device)
    BCAST_STATS["seconds"] += time.perf_counter() - t0
    BCAST_STATS["bytes"] += flat.numel() * flat.element_size()
    BCAST_STATS["collectives"] += 1


def ranks_seen(device=None, force=False):
    """One record per rank -- host, device index, name and the GPU's identity (UUID / PCI bus id as the runtime reports them)
    -- all-gathered, so a multi-GPU bench line can PROVE that N ranks sat on N distinct GPUs (`distinct_gpus`)."""
    import socket
    rec = {"rank": dist.get_rank() if dist.is_initialized() else 0, "host": socket.gethostname(), "pid": os.getpid()}
    if device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available():
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        prop = torch.cuda.get_device_properties(idx)
        rec.update(device=idx, name=prop.name, uuid=str(getattr(prop, "uuid", "")),
                   pci="%s:%s:%s" % (getattr(prop, "pci_domain_id", "?"), getattr(prop, "pci_bus_id", "?"),
                                      getattr(prop, "pci_device_id", "?")))
    else:
        rec.update(device=None, name="cpu", uuid="", pci="")
    if _collective(force):
        recs = [None] * dist.get_world_size()
        dist.all_gather_object(recs, rec)
    else:
        recs = [rec]
    ids = {(r["host"], r["uuid"] or r["pci"] or r["pid"]) for r in recs}
    return {"ranks": recs, "distinct_gpus": len(ids)}


def env_world():
    """(rank, local_rank, world) of the torchrun environment.  ALG_DIST_ONE_GPU=1 (test hook: exercising the multi-process
    launch path on a one-GPU box, together with ALG_DIST_BACKEND=gloo) maps every local rank to device 0."""
    local = 0 if os.environ.get("ALG_DIST_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    return int(os.environ.get("RANK", "0")), local, int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None, force=False):
    """Initialise the default process group from the torchrun environment; no-op for world_size 1 (unless forced: a one-rank
    group on the chosen backend -- see FORCE_COLLECTIVES)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force or FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("ALG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        timeout = datetime.timedelta(seconds=int(os.environ.get("ALG_DIST_TIMEOUT_S", DIST_TIMEOUT_S)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def shard_videos(num_videos, rank, world):
    """video v -> rank v mod world (independent videos share nothing but weights)."""
    return [v for v in range(num_videos) if v % world == rank]


def broadcast_state_dict(make_state_dict, shapes, device, src=0, dtype=torch.bfloat16, bucket_bytes=1 << 30, force=False):
    """Rank ``src`` materialises the weights (``make_state_dict()``), every other rank allocates empty tensors of
    ``shapes`` and receives them.  Tensors are coalesced into ~1 GiB flat buckets so the broadcast is a handful of
    large collectives (xGMI is point-to-point: few, large transfers) rather than ~700 small ones."""
    if not _collective(force):
        return make_state_dict()
    rank = dist.get_rank()
    sd = make_state_dict() if rank == src else None
    out = {}
    esize = torch.empty((), dtype=dtype).element_size()
    names = list(shapes.keys())
    i = 0
    while i < len(names):
        group, nbytes = [], 0
        while i < len(names) and (not group or nbytes + _numel(shapes[names[i]]) * esize <= bucket_bytes):
            group.append(names[i])
            nbytes += _numel(shapes[names[i]]) * esize
            i += 1
        flat = torch.empty(nbytes // esize, device=device, dtype=dtype)
        if rank == src:
            off = 0
            for n in group:
                k = _numel(shapes[n])
                flat[off:off + k].copy_(sd[n].to(device=device, dtype=dtype).reshape(-1))
                off += k
        _timed_broadcast(flat, src)
        off = 0
        for n in group:
            k = _numel(shapes[n])
            out[n] = flat[off:off + k].view(shapes[n])
            off += k
    return out


def broadcast_loaded_state_dict(sd, device, src=0, bucket_bytes=1 << 30, force=False):
    """``sd`` is the state dict on rank ``src`` and None elsewhere (only rank ``src`` read the checkpoint from disk):
    names / shapes / dtypes travel as one small object broadcast, the tensors in ~1 GiB flat buckets per dtype (xGMI is
    point-to-point: few, large transfers).  Returns the same dict -- same names, dtypes, bits -- on every rank, tensors on
    ``device``."""
    if not _collective(force):
        return sd
    rank = dist.get_rank()
    # ``sd`` may be an exception on rank ``src`` (the caller caught a failed read): the status travels first, so every rank
    # raises instead of sitting in a broadcast until the watchdog fires
    if rank == src:
        meta = [("error", repr(sd))] if isinstance(sd, BaseException) else \
            [("ok", [(k, tuple(v.shape), v.dtype) for k, v in sd.items()])]
    else:
        meta = [None]
    dist.broadcast_object_list(meta, src=src)
    status, meta = meta[0]
    if status != "ok":
        raise RuntimeError("rank %d failed to load the weights it was to broadcast: %s" % (src, meta))
    out = {}
    by_dtype = {}
    for k, shape, dt in meta:
        by_dtype.setdefault(dt, []).append((k, shape))
    for dt, items in by_dtype.items():
        esize = torch.empty((), dtype=dt).element_size()
        # byte buckets: the collective runs on uint8 so every dtype (fp8 and int64 included) takes the same path
        i = 0
        while i < len(items):
            group, nbytes = [], 0
            while i < len(items) and (not group or nbytes + _numel(items[i][1]) * esize <= bucket_bytes):
                group.append(items[i])
                nbytes += -(-_numel(items[i][1]) * esize // 16) * 16      # keep every tensor 16-byte aligned in the bucket
                i += 1
            if nbytes == 0:                                                # only zero-element tensors: nothing to send
                for k, shape in group:
                    out[k] = torch.empty(shape, dtype=dt, device=device)
                continue
            flat = torch.empty(nbytes, device=device, dtype=torch.uint8)
            if rank == src:
                off = 0
                for k, shape in group:
                    nb = _numel(shape) * esize
                    flat[off:off + nb].copy_(sd[k].to(device).contiguous().reshape(-1).view(torch.uint8))
                    off += -(-nb // 16) * 16
            _timed_broadcast(flat, src)
            off = 0
            for k, shape in group:
                nb = _numel(shape) * esize
                t = flat[off:off + nb].view(dt).view(shape)
                out[k] = t.clone() if nb < SMALL_TENSOR_BYTES and nb < nbytes else t   # small tensors do not pin the bucket
                off += -(-nb // 16) * 16
    return {k: out[k] for k, _, _ in meta}


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def max_over_ranks(value, device, force=False):
    """MAX all-reduce of a python float (the bench's timing rule)."""
    if not _collective(force):
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(force=False):
    if _collective(force):
        dist.barrier()


class CFGPairSplit:
    """The cond / uncond CFG branches of ONE video on two GPUs (BASELINE config 5): ranks 2i and
    2i+1 form a pair, each evaluates its share of the step's DiT sample-forwards, one small all-gather per step merges
    the predictions (the only data-path collective in this build -- the path has a real exchange step here), and both
    ranks then apply the identical CFG combine + scheduler step, so their latents stay bit-identical without a second
    exchange.  Trades throughput-neutral weak scaling for ~2x lower latency per video.

        2 passes [uncond, text]                  -> rank 0: uncond            rank 1: text
        3 passes [uncond_init, uncond, text]     -> rank 0: uncond_init, uncond   rank 1: text
    """

    def __init__(self, group=None, pair_rank=None):
        self.group = group
        if pair_rank is None:
            pair_rank = dist.get_rank() % 2 if dist.is_initialized() else 0
        self.pair_rank = int(pair_rank)

    @classmethod
    def from_world(cls):
        """Build the pair groups of the default process group (every rank must call this: new_group is collective)."""
        world, rank = dist.get_world_size(), dist.get_rank()
        if world % 2:
            raise ValueError("the CFG pair split needs an even number of ranks, got %d" % world)
        mine = None
        for i in range(world // 2):
            g = dist.new_group([2 * i, 2 * i + 1])
            if rank // 2 == i:
                mine = g
        return cls(mine, rank % 2)

    def my_passes(self, n_pass):
        if n_pass < 2:
            return list(range(n_pass)) if self.pair_rank == 0 else []
        return list(range(n_pass - 1)) if self.pair_rank == 0 else [n_pass - 1]

    def merge(self, local_pred, n_pass, batch):
        """local_pred [len(my_passes) * batch, ...] -> full [n_pass * batch, ...] on both ranks.  One all-gather over the
        pair moves exactly the partner's rows (a 3-pass step is split 2 + 1: the single-pass rank pads its send buffer to
        the pair's common size, the pad row is never read); rows are copied, never summed, so the merged prediction is the
        partner's bits (signed zeros included)."""
        mine = self.my_passes(n_pass)
        counts = [len(CFGPairSplit(None, r).my_passes(n_pass)) for r in (0, 1)]
        rows = max(counts) * batch
        tail = tuple(local_pred.shape[1:])
        if local_pred.shape[0] != len(mine) * batch:
            raise ValueError("local prediction has %d rows, expected %d" % (local_pred.shape[0], len(mine) * batch))
        send = local_pred.contiguous()
        if send.shape[0] != rows:
            pad = torch.empty((rows,) + tail, dtype=send.dtype, device=send.device)
            pad[:send.shape[0]] = send
            send = pad
        parts = [torch.empty((rows,) + tail, dtype=send.dtype, device=send.device) for _ in range(2)]
        self.all_gather(parts, send)
        full = torch.empty((n_pass * batch,) + tail, dtype=send.dtype, device=send.device)
        for r in (0, 1):
            for j, p in enumerate(CFGPairSplit(None, r).my_passes(n_pass)):
                full[p * batch:(p + 1) * batch] = parts[r][j * batch:(j + 1) * batch]
        return full

    def all_gather(self, parts, t):
        """parts[r] <- rank r's `t` for both ranks of the pair (RCCL over the pair's direct xGMI link)."""
        dist.all_gather(parts, t, group=self.group)
