"""GPU parity of the low-pass kernels (through the C ABI via alg_amd.lp_utils) against the CPU oracle,
the reference-generated golden vectors, and ATen's own GPU op (the op the reference calls, lp_utils.py:53-54)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import alg_amd
from alg_amd import lp_utils
from oracle import lp_oracle

pytestmark = pytest.mark.gpu


def bf16_ulp(x):
    return np.maximum(np.abs(x), 2.0 ** -120) * 2.0 ** -7


def test_down_up_f32_golden_vectors(device, golden_dir):
    with open(os.path.join(golden_dir, "lp_misc.json")) as f:
        meta = json.load(f)["down_up_meta"]
    vec = np.load(os.path.join(golden_dir, "down_up_vectors.npz"))
    for m in meta:
        x = torch.from_numpy(vec[m["name"] + "_in"]).to(device)
        y = lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, m["factor"])
        assert y is not x and y.shape == x.shape and y.dtype == x.dtype
        err = np.abs(y.cpu().numpy() - vec[m["name"] + "_out"]).max()
        assert err <= 3e-6, (m["name"], err)  # fp32 rounding (fma vs mul+add ordering)


def test_register_blocked_kernels_against_golden_vectors_and_oracle(device, golden_dir, monkeypatch):
    """lowpass_v3.hip takes calls of 128 planes and more; here the plane threshold is dropped to 1 so that the
    reference-generated golden vectors (small plane counts) and the oracle cases run through it directly, not only through
    its bit-identity with the plane-per-workgroup kernels."""
    monkeypatch.setenv("ALG_LOWPASS_PATH", "3")      # lowpass_v3.hip at any plane count
    with open(os.path.join(golden_dir, "lp_misc.json")) as f:
        meta = json.load(f)["down_up_meta"]
    vec = np.load(os.path.join(golden_dir, "down_up_vectors.npz"))
    for m in meta:
        x = torch.from_numpy(vec[m["name"] + "_in"]).to(device)
        y = lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, m["factor"])
        err = np.abs(y.cpu().numpy() - vec[m["name"] + "_out"]).max()
        assert err <= 3e-6, (m["name"], err)
    g = torch.Generator().manual_seed(7)
    for shape, factor in (((1, 16, 13, 60, 90), 0.25), ((1, 20, 21, 60, 104), 0.4), ((1, 16, 1, 90, 160), 0.625),
                          ((1, 16, 3, 32, 32), 0.25), ((1, 2, 12, 18), 0.5)):
        x = torch.randn(shape, generator=g)
        y = lp_utils.apply_low_pass_filter(x.to(device), "down_up", 0.0, 0, factor).cpu().numpy()
        assert np.abs(y - lp_oracle.down_up(x.numpy(), factor, np.float32)).max() <= 3e-6, (shape, factor)
    for shape, k, sigma in (((1, 20, 21, 60, 104), 9, 15.0), ((1, 16, 3, 32, 32), 5, 1.5), ((1, 16, 13, 60, 90), 7, 2.0),
                            ((2, 3, 12, 18), 3, 0.8)):
        x = torch.randn(shape, generator=g)
        y = lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", sigma, k, 1.0).cpu().numpy()
        ref = lp_oracle.gaussian_blur(x.numpy().astype(np.float64), lp_oracle.gaussian_kernel_size(k, shape[-2]), sigma)
        assert np.abs(y - ref).max() <= 1e-5, (shape, k, sigma)


@pytest.mark.parametrize("shape,factor", [((1, 16, 13, 60, 90), 0.25), ((1, 20, 21, 60, 104), 0.4),
                                          ((1, 16, 1, 90, 160), 0.625), ((1, 16, 3, 32, 32), 0.25),
                                          ((2, 3, 7, 5), 0.1), ((1, 2, 13, 17), 0.5), ((3, 1, 1, 2, 60, 90)[1:], 0.9)])
def test_down_up_f32_vs_oracle(device, shape, factor):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g)
    y = lp_utils.apply_low_pass_filter(x.to(device), "down_up", 0.0, 0, factor).cpu().numpy()
    ref = lp_oracle.down_up(x.numpy(), factor, np.float32)
    assert np.abs(y - ref).max() <= 3e-6
    # and against ATen's GPU kernel for the same op (what the reference executes on a GPU)
    xs = x.to(device).reshape(-1, 1, *shape[-2:])
    h1, w1 = lp_oracle.down_up_size(shape[-2], shape[-1], factor)
    a = F.interpolate(xs, size=(h1, w1), mode="bilinear", align_corners=False, antialias=True)
    a = F.interpolate(a, size=shape[-2:], mode="bilinear", align_corners=False, antialias=True)
    assert np.abs(y.reshape(-1) - a.cpu().numpy().reshape(-1)).max() <= 2e-5


@pytest.mark.parametrize("shape,factor", [((1, 16, 13, 60, 90), 0.25), ((1, 20, 4, 60, 104), 0.4),
                                          ((1, 16, 1, 90, 160), 0.625)])
def test_down_up_bf16(device, shape, factor):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).to(torch.bfloat16)
    y = lp_utils.apply_low_pass_filter(x.to(device), "down_up", 0.0, 0, factor)
    assert y.dtype == torch.bfloat16
    got = y.float().cpu().numpy()
    # oracle on the bf16-rounded input, bf16-rounded intermediate and result (two interpolate calls, lp:53-54)
    ref = lp_oracle.down_up(x.float().numpy(), factor, np.float32, storage="bf16")
    exact = lp_oracle.down_up(x.float().numpy(), factor, np.float64)
    # stated bf16 tolerance: 2 bf16 ulps vs the rounded oracle (an intermediate rounding flip moves the result by
    # at most one more ulp), and 1% of the plane scale vs the unrounded fp64 result
    assert (np.abs(got - ref) <= 2 * bf16_ulp(ref) + 1e-3).all()
    assert np.abs(got - exact).max() <= 1e-2 * max(1.0, np.abs(exact).max())
    assert (np.abs(got - ref) > 0).mean() < 0.05  # almost everywhere bit-identical to the rounded oracle
    # ATen's GPU bf16 kernel keeps its tap weights in bf16; report-level check only (loose, documented in DESIGN.md)
    xs = x.to(device).reshape(-1, 1, *shape[-2:])
    h1, w1 = lp_oracle.down_up_size(shape[-2], shape[-1], factor)
    a = F.interpolate(xs, size=(h1, w1), mode="bilinear", align_corners=False, antialias=True)
    a = F.interpolate(a, size=shape[-2:], mode="bilinear", align_corners=False, antialias=True)
    assert np.abs(got.reshape(-1) - a.float().cpu().numpy().reshape(-1)).max() <= 6e-2


def test_down_up_cogvideox_condition_properties(device):
    """C2 conditioning tensor: frame 0 real, frames 1..12 zero padding (cog:402-411) must stay exactly zero;
    constants are preserved (normalised taps); the filter is linear."""
    g = torch.Generator().manual_seed(5)
    cond = torch.zeros(1, 13, 16, 60, 90)
    cond[:, 0] = torch.randn(1, 16, 60, 90, generator=g) * 0.7
    for dt in (torch.float32, torch.bfloat16):
        x = cond.to(device=device, dtype=dt)
        y = lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, 0.25)
        assert torch.count_nonzero(y[:, 1:]) == 0
        assert torch.count_nonzero(y[:, 0]) > 0
    c = torch.full((1, 2, 3, 60, 90), 1.5, device=device)
    assert (lp_utils.apply_low_pass_filter(c, "down_up", 0.0, 0, 0.25) - 1.5).abs().max() <= 1e-6
    a = torch.randn(1, 4, 2, 60, 90, generator=g).to(device)
    b = torch.randn(1, 4, 2, 60, 90, generator=g).to(device)
    f = lambda t: lp_utils.apply_low_pass_filter(t, "down_up", 0.0, 0, 0.25)
    assert (f(a + 2 * b) - (f(a) + 2 * f(b))).abs().max() <= 1e-5


def test_down_up_low_pass_behaviour(device):
    """A plane at the Nyquist frequency is wiped out by the f=0.25 filter, a smooth ramp survives."""
    yy, xx = torch.meshgrid(torch.arange(60.), torch.arange(90.), indexing="ij")
    checker = ((yy + xx) % 2 * 2 - 1).reshape(1, 1, 60, 90).to(device)
    ramp = (xx / 90 + yy / 60).reshape(1, 1, 60, 90).to(device)
    f = lambda t: lp_utils.apply_low_pass_filter(t, "down_up", 0.0, 0, 0.25)
    assert f(checker)[..., 8:-8, 8:-8].abs().max() < 1e-3
    assert (f(ramp) - ramp)[..., 8:-8, 8:-8].abs().max() < 1e-3


@pytest.mark.parametrize("shape,k,sigma", [((1, 20, 21, 60, 104), 9, 15.0), ((1, 20, 3, 60, 104), 9, 0.3846153846),
                                           ((2, 3, 13, 17), 3, 0.7), ((1, 1, 1, 90, 160), 21, 4.0),
                                           ((1, 2, 1, 60, 104), 8, 2.0)])
def test_gaussian_f32_vs_oracle(device, shape, k, sigma):
    g = torch.Generator().manual_seed(13)
    x = torch.randn(shape, generator=g)
    y = lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", sigma, k, 1.0).cpu().numpy()
    kk = lp_oracle.gaussian_kernel_size(k, shape[-2])
    ref = lp_oracle.gaussian_blur(x.numpy().astype(np.float64), kk, sigma)
    assert np.abs(y - ref).max() <= 1e-5


def test_gaussian_relative_kernel_and_bf16(device):
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1, 20, 2, 60, 104, generator=g)
    # default relative size 0.02734375 * 60 -> k = 1: the blur is an identity *computation* (SURVEY a-2)
    y = lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", 15.0, 0.02734375, 1.0)
    assert torch.equal(y.cpu(), x)
    xb = x.to(torch.bfloat16)
    yb = lp_utils.apply_low_pass_filter(xb.to(device), "gaussian_blur", 3.0, 9, 1.0).float().cpu().numpy()
    ref = lp_oracle.gaussian_blur(xb.float().numpy().astype(np.float64), 9, 3.0)
    assert (np.abs(yb - ref) <= bf16_ulp(ref) + 1e-3).all()


def test_filter_edge_cases(device):
    e = torch.zeros(0, 3, 8, 8, device=device)
    assert lp_utils.apply_low_pass_filter(e, "down_up", 0.0, 0, 0.5).shape == e.shape  # empty batch
    x = torch.randn(1, 2, 3, 4, 5, device=device)
    assert lp_utils.apply_low_pass_filter(x, "none", 1.0, 3, 0.5) is x
    assert lp_utils.apply_low_pass_filter(x, "down_up", 1.0, 3, 1.0) is x
    assert lp_utils.apply_low_pass_filter(x, "gaussian_blur", 0, 3, 0.5) is x
    with pytest.raises(RuntimeError):
        lp_utils.apply_low_pass_filter(x.permute(0, 2, 1, 3, 4), "down_up", 0.0, 0, 0.5)
    # pixel-sized planes exceed the LDS-resident kernel and take the global-memory passes (same arithmetic)
    y = lp_utils.apply_low_pass_filter(torch.ones(1, 3, 480, 720, device=device), "down_up", 0.0, 0, 0.25)
    assert (y - 1).abs().max() <= 1e-6
    with pytest.raises(alg_amd.AlgHipError):
        lp_utils.apply_low_pass_filter(torch.zeros(1, 3, 8, 8, device=device, dtype=torch.float16), "down_up", 0., 0, .5)
    # strength-modulated factors of a linear schedule all run
    for s in (1.0, 0.9487179487179487, 0.5128205128205128, 0.02564102564102566):
        f_eff = 1.0 - 0.75 * s
        y = lp_utils.apply_low_pass_filter(x.new_ones(1, 1, 1, 60, 90), "down_up", 0.0, 0, f_eff)
        assert (y - 1).abs().max() <= 1e-6


@pytest.mark.parametrize("shape,factor", [((1, 3, 480, 720), 0.25), ((2, 3, 256, 256), 0.4), ((1, 3, 352, 608), 0.625)])
def test_down_up_pixel_sized_planes(device, shape, factor):
    """The pixel-space ALG branch (cog:628-643) filters the RGB image: planes beyond the LDS budget go through
    lowpass_big.hip -- same oracle, same tolerances as the LDS-resident kernel, and the two paths agree bit-for-bit
    on a plane both can take."""
    g = torch.Generator().manual_seed(23)
    x = torch.randn(shape, generator=g)
    y = lp_utils.apply_low_pass_filter(x.to(device), "down_up", 0.0, 0, factor).cpu().numpy()
    ref = lp_oracle.down_up(x.numpy(), factor, np.float32)
    assert np.abs(y - ref).max() <= 3e-6
    xb = x.to(torch.bfloat16)
    yb = lp_utils.apply_low_pass_filter(xb.to(device), "down_up", 0.0, 0, factor).float().cpu().numpy()
    refb = lp_oracle.down_up(xb.float().numpy(), factor, np.float32, storage="bf16")
    assert (np.abs(yb - refb) <= 2 * bf16_ulp(refb) + 1e-3).all()
    assert (np.abs(yb - refb) > 0).mean() < 0.05


def test_gaussian_pixel_sized_planes(device):
    g = torch.Generator().manual_seed(29)
    x = torch.randn(1, 3, 480, 720, generator=g)
    y = lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", 2.0, 9, 1.0).cpu().numpy()
    ref = lp_oracle.gaussian_blur(x.numpy().astype(np.float64), 9, 2.0)
    assert np.abs(y - ref).max() <= 1e-5
    xb = x.to(torch.bfloat16)
    yb = lp_utils.apply_low_pass_filter(xb.to(device), "gaussian_blur", 2.0, 9, 1.0).float().cpu().numpy()
    refb = lp_oracle.gaussian_blur(xb.float().numpy().astype(np.float64), 9, 2.0)
    assert (np.abs(yb - refb) <= bf16_ulp(refb) + 1e-3).all()


def test_gaussian_more_than_255_taps(device):
    """An integer `lp_blur_kernel_size` larger than 255 (lp:44-46 accepts any size; reflect padding needs k // 2 < min(H, W)):
    the global-memory passes have no tap-count limit."""
    g = torch.Generator().manual_seed(30)
    x = torch.randn(1, 2, 300, 340, generator=g)
    y = lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", 60.0, 301, 1.0).cpu().numpy()
    ref = lp_oracle.gaussian_blur(x.numpy().astype(np.float64), 301, 60.0)
    assert np.abs(y - ref).max() <= 1e-5
    with pytest.raises(RuntimeError, match="Padding size"):
        lp_utils.apply_low_pass_filter(x.to(device), "gaussian_blur", 60.0, 601, 1.0)


def test_global_memory_path_is_bit_identical_to_the_lds_path(device, monkeypatch):
    """ALG_LOWPASS_PATH=4 sends LDS-sized planes through lowpass_big.hip: same bits."""
    g = torch.Generator().manual_seed(31)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(1, 16, 3, 60, 90, generator=g).to(dt).to(device)
        monkeypatch.delenv("ALG_LOWPASS_PATH", raising=False)
        a = lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, 0.25)
        b = lp_utils.apply_low_pass_filter(x, "gaussian_blur", 3.0, 9, 1.0)
        monkeypatch.setenv("ALG_LOWPASS_PATH", "4")
        assert torch.equal(lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, 0.25), a)
        assert torch.equal(lp_utils.apply_low_pass_filter(x, "gaussian_blur", 3.0, 9, 1.0), b)


@pytest.mark.parametrize("shape,dtype,kind,arg", [
    ((1, 13, 16, 60, 90), torch.bfloat16, "down_up", 0.25),       # C2 condition (one video: one plane per workgroup)
    ((8, 13, 16, 60, 90), torch.bfloat16, "down_up", 0.25),       # 8 videos: persistent grid, several planes per workgroup
    ((8, 13, 16, 60, 90), torch.float32, "down_up", 0.25),
    ((1, 20, 21, 60, 104), torch.float32, "down_up", 0.4),        # Wan 480p condition
    ((3, 20, 21, 90, 160), torch.float32, "down_up", 0.4),        # C5 (57.6 KB planes: 512-thread workgroups, 8 prefetch regs)
    ((2, 16, 1, 90, 160), torch.float32, "down_up", 0.625),       # C4 first frame
    ((1, 16, 3, 32, 32), torch.float32, "down_up", 0.25),         # C1
    ((2, 16, 1, 44, 76), torch.bfloat16, "down_up", 0.8125),      # a schedule-modulated factor on a bucketed size
    ((1, 20, 21, 60, 104), torch.float32, "gaussian_blur", (9, 15.0)),
    ((8, 20, 21, 60, 104), torch.float32, "gaussian_blur", (9, 7.5)),
    ((8, 13, 16, 60, 90), torch.bfloat16, "gaussian_blur", (13, 3.0)),
    ((2, 20, 21, 90, 160), torch.float32, "gaussian_blur", (19, 15.0)),   # k > 16: taps read from LDS
    # register-blocked kernels (lowpass_v3.hip): W a multiple of 4, more than two planes per CU
    ((8, 16, 13, 60, 88), torch.bfloat16, "gaussian_blur", (5, 2.0)),
    ((4, 20, 21, 60, 104), torch.float32, "gaussian_blur", (3, 1.0)),
    ((8, 16, 13, 32, 36), torch.float32, "gaussian_blur", (15, 4.0)),
    ((8, 16, 13, 17, 20), torch.bfloat16, "gaussian_blur", (7, 1.5)),    # H not a multiple of the row block
    ((8, 16, 13, 18, 24), torch.float32, "gaussian_blur", (17, 9.0)),    # pad = 8 = H / 2 - 1: mirrored rows overlap the far edge
    ((8, 16, 13, 30, 44), torch.float32, "gaussian_blur", (11, 3.0)),
    ((8, 16, 13, 30, 46), torch.float32, "gaussian_blur", (5, 1.2)),     # W = 2 mod 4: padded rows, half quads
    ((8, 16, 13, 34, 50), torch.bfloat16, "gaussian_blur", (9, 2.5)),
    ((8, 16, 13, 60, 104), torch.float32, "down_up", 0.25),             # integer scale, 9 taps
    ((8, 16, 13, 44, 76), torch.bfloat16, "down_up", 0.8125),           # 5 taps
    ((8, 16, 13, 90, 160), torch.float32, "down_up", 0.625),
    ((8, 16, 13, 30, 46), torch.bfloat16, "down_up", 0.5),              # W = 2 mod 4: chunks straddle rows, half quads
    ((8, 16, 13, 31, 44), torch.float32, "down_up", 0.3),               # odd H, 9 / 11 taps
    ((8, 16, 13, 18, 22), torch.float32, "down_up", 0.2),               # scale 5.5 / 6 -> 11 and 13 taps: W pass fits, H pass table-driven
    ((8, 16, 13, 60, 90), torch.float32, "down_up", 0.34),
])
def test_bandwidth_shaped_kernels_are_bit_identical_to_the_plane_per_workgroup_kernels(device, monkeypatch, shape, dtype,
                                                                                       kind, arg):
    """lowpass_v2.hip (persistent grid, host-built tap tables, register prefetch of the next plane, staged 16-byte stores)
    must reproduce lowpass.hip's kernels bit for bit: same fp32 tap tables, same fma order."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=g).to(dtype).to(device)
    call = (lambda: lp_utils.apply_low_pass_filter(x, "down_up", 0.0, 0, arg)) if kind == "down_up" else \
        (lambda: lp_utils.apply_low_pass_filter(x, "gaussian_blur", arg[1], arg[0], 1.0))
    new = call()
    monkeypatch.setenv("ALG_LOWPASS_PATH", "1")
    old = call()
    monkeypatch.delenv("ALG_LOWPASS_PATH")
    assert new.dtype == dtype and new.shape == x.shape
    assert torch.equal(new, old)
    assert torch.equal(call(), new)                      # deterministic, table cache warm
    assert not torch.equal(new, x)
