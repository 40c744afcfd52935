"""The box-calibration entry points of the C ABI (csrc/calibrate.hip; bench.py's `calibration` and `roofline.extra.box`): the
register-only MFMA loop runs and reports a plausible rate and shader clock, and the clock taps inside the product attention
kernels write plausible clocks WITHOUT changing a bit of the attention output."""
import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF = torch.bfloat16


def test_mfma_calibration_loop_reports_a_plausible_rate_and_clock():
    sink = torch.zeros(4, dtype=torch.float32, device=DEV)
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    clocks = torch.zeros(2 * cus, 4, dtype=torch.int64, device=DEV)
    iters = 2000
    assert _lib.calib_mfma_bf16(sink, iters, 1, 0, clocks) == 2 * cus            # 0 blocks = two workgroups per CU
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(8):
        _lib.calib_mfma_bf16(sink, iters, 2 + i, 2 * cus, clocks)
    b.record()
    b.synchronize()
    tflops = 8 * 2.0 * 32 * 32 * 16 * 32 * iters * 4 * 2 * cus / (a.elapsed_time(b) / 1e3) / 1e12
    assert 800.0 < tflops < 2600.0, tflops                                       # below the 2.5 PF peak, above anything broken
    khz = _lib.wall_clock_khz()
    assert khz == 100_000
    clk = _lib.clock_mhz_from_taps(clocks, khz)
    assert clk["workgroups"] == 2 * cus and 800.0 < clk["min"] <= clk["mean"] <= clk["max"] < 2500.0, clk
    # pipe-busy share at that clock cannot exceed 1
    assert tflops / (cus * 4 * 1024.0 * clk["mean"] * 1e6 / 1e12) < 1.02
    assert float(sink.abs().sum()) == 0.0                                        # the loop's guard never fires


@pytest.mark.parametrize("hd", [64, 128])
def test_attention_clock_taps_leave_the_output_bit_identical(hd):
    S, H, N = (9000, 8, 1) if hd == 64 else (4300, 8, 1)                         # long enough for the statement kernels (d = 128: >= 4,096 keys)
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(hd)
    q = (torch.randn(N, S, D, generator=g, device=DEV) * (0.18 if hd == 64 else 1.0)).to(BF)
    k = torch.randn(N, S, D, generator=g, device=DEV).to(BF)
    s_pad = (S + 127) // 128 * 128
    vt = torch.zeros(N, D, s_pad, dtype=BF, device=DEV)
    vt[:, :, :S] = torch.randn(N, D, S, generator=g, device=DEV).to(BF)

    def run():
        o = torch.empty(N, S, D, dtype=BF, device=DEV)
        if hd == 64:
            qk = torch.cat([q, k], dim=-1).contiguous()
            _lib.flash_attn_d64(qk, qk, vt, o, N, H, S, S * 2 * D, 2 * D, D * s_pad, s_pad, S * D, D, 0.125, k_off=D, q_prescaled=True)
        else:
            _lib.flash_attn_d128(q, k, vt, o, N, H, S, S, S * D, D, S * D, D, D * s_pad, s_pad, S * D, D, 128 ** -0.5)
        torch.cuda.synchronize()
        return o

    plain = run()
    taps = torch.zeros(64, 4, dtype=torch.int64, device=DEV)
    _lib.attn_clock_tap(taps)
    try:
        tapped = run()
    finally:
        _lib.attn_clock_tap(None)
    assert torch.equal(tapped, plain)
    clk = _lib.clock_mhz_from_taps(taps, _lib.wall_clock_khz())
    assert clk is not None and clk["workgroups"] >= 1 and 500.0 < clk["min"] and clk["max"] < 2600.0, clk
    after = taps.clone()
    assert torch.equal(run(), plain) and torch.equal(taps, after)                # switched off: nothing is written any more
    with pytest.raises(_lib.AlgHipError):
        _lib.attn_clock_tap(torch.zeros(8, 3, dtype=torch.int64, device=DEV))
