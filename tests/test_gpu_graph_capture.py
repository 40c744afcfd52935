"""The C ABI is enqueue-only and allocation-free (include/alg_hip.h "Conventions"): one ALG
sampler step of the C2 workload -- low-pass filter of the condition, the CogVideoX DiT forward over the two-pass CFG batch
(17,776 tokens x 3072, 2 of the 42 blocks), fused CFG + DDIM update in place -- is captured into a hipGraph on a side stream
and replayed; the replay must reproduce the eagerly enqueued step bit for bit, also on new latents.  A `hipMalloc`, a
`hipFree`, a blocking copy or a stream synchronisation inside any entry point fails the capture."""
import pytest
import torch

from alg_amd import CogVideoXTransformer3DModel, CogVideoXTransformerConfig, _lib, lp_utils
from alg_amd.pipeline_cogvideox_image2video_lowpass import get_resize_crop_region_for_grid, rotary_tables

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_c2_sampler_step_captures_into_a_hip_graph_and_replays_bit_identically(device):
    cfg = CogVideoXTransformerConfig(num_layers=2)
    model = CogVideoXTransformer3DModel.from_synthetic(cfg, seed=11, device=device)
    g = torch.Generator().manual_seed(8)
    lat0 = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    lat1 = torch.randn(1, 13, 16, 60, 90, generator=g).to(device, BF)
    cond = torch.zeros(1, 13, 16, 60, 90, dtype=BF, device=device)
    cond[:, 0] = (torch.randn(1, 16, 60, 90, generator=g) * 0.7).to(device, BF)
    ehs = torch.randn(2, 226, 4096, generator=g).to(device, BF)
    ts = torch.full((2,), 979.0, device=device)
    rope = tuple(t.to(device) for t in rotary_tables(64, get_resize_crop_region_for_grid((30, 45), 45, 30), (30, 45), 13))

    def step(lat):
        """filter (lp:49-54) -> forward over [lp | lp] (cog:1060-1090) -> CFG + DDIM in place (cog:1091-1123)"""
        lp = lp_utils.apply_low_pass_filter(cond, "down_up", 0.0, 0, 0.25)
        pred = model.forward_assembled(lat, [lp, lp], ehs, ts, rope)
        _lib.cfg_ddim_step_(pred, lat, 2, 6.0, 0.6, 0.8, 0.31, 0.27)
        return pred

    want = []
    for src in (lat0, lat1):
        lat = src.clone()
        want.append((step(lat).clone(), lat))
    torch.cuda.synchronize()

    lat_g = lat0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on the capture stream: per-stream tables / scratch exist
        step(lat_g)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        pred_g = step(lat_g)
    for src, (pred_e, lat_e) in zip((lat0, lat1, lat0), want + want[:1]):
        lat_g.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(pred_g, pred_e)
        assert torch.equal(lat_g, lat_e)
    assert not torch.equal(want[0][0], want[1][0])


def test_filters_on_pixel_sized_planes_run_in_caller_workspace_under_capture(device):
    """Planes beyond LDS (pixel-space ALG, cog:417-433 decode -> filter -> encode) go through global-memory passes whose fp32
    intermediates live in a caller-owned workspace: capturable, and bit-identical to the eager call."""
    g = torch.Generator(device=device).manual_seed(3)
    x = torch.randn(3, 480, 720, generator=g, device=device)
    want_d = _lib.down_up(x, 120, 180)
    want_g = _lib.gaussian_blur(x, 9, 2.0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _lib.down_up(x, 120, 180), _lib.gaussian_blur(x, 9, 2.0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        got_d = _lib.down_up(x, 120, 180)
        got_g = _lib.gaussian_blur(x, 9, 2.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_d, want_d) and torch.equal(got_g, want_g)


def test_buffers_a_capture_saw_outlive_the_caches_and_cold_tables_stay_out_of_them(device):
    """(1) a tap table first built INSIDE a capture is filled by the replay, not at capture time -- it must not
    reach the eager cache; (2) a table / scratch buffer a capture was handed stays alive when the caches evict or outgrow it:
    the graph holds its raw pointer."""
    g = torch.Generator(device=device).manual_seed(4)
    x = torch.randn(4, 3, 44, 52, generator=g, device=device)              # a plane shape nothing else in the suite uses
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    n_tables, n_pinned = len(_lib._TABLES), len(_lib._PINNED)
    with torch.cuda.graph(graph, stream=side):                             # NO warm-up: the table build is captured
        got = _lib.down_up(x, 11, 13)
    assert len(_lib._TABLES) == n_tables and len(_lib._PINNED) > n_pinned  # pinned for the graph, not cached
    with torch.cuda.stream(side):
        eager = _lib.down_up(x, 11, 13)                                    # builds its own table: no half-built hit
    torch.cuda.synchronize()
    want = _lib.down_up(x, 11, 13)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(eager, want) and torch.equal(got, want)
    # (2) warm-up, capture, then evict everything and outgrow the scratch: the replay must still be right
    big = torch.randn(3, 480, 720, generator=g, device=device)
    with torch.cuda.stream(side):
        _lib.down_up(big, 120, 180)
    torch.cuda.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        got2 = _lib.down_up(big, 120, 180)
        got3 = _lib.down_up(x, 11, 13)
    want2 = _lib.down_up(big, 120, 180)
    _lib.clear_caches()
    with torch.cuda.stream(side):
        _lib.down_up(torch.randn(6, 600, 800, generator=g, device=device), 150, 200)   # a larger scratch on the same stream
        junk = [torch.randn(1 << 20, device=device) for _ in range(8)]                  # reuse whatever the allocator freed
    torch.cuda.synchronize()
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(got2, want2) and torch.equal(got3, want)
    del junk
