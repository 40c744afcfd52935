"""CPU tests of the product's host-side logic (no kernel is launched): the lp_utils mirror, the scheduler
tables, RoPE tables, the C-ABI surface, pipeline argument checking and the loop's branch table."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import alg_amd
from alg_amd import lp_utils
from alg_amd.pipeline_cogvideox_image2video_lowpass import (CogVideoXImageToVideoPipeline, get_resize_crop_region_for_grid,
                                                            rotary_tables)
from alg_amd.schedulers import CogVideoXDDIMScheduler
from alg_amd.transformer_cogvideox import CogVideoXTransformerConfig
from alg_amd import weights as W
from oracle import ddim_oracle, dit_oracle, loop_oracle, lp_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(ps, i, total):
    return dict(step_index=i, total_steps=total, lp_strength_schedule_type=ps["kind"],
                schedule_interval_start_time=ps.get("start", 0.0), schedule_interval_end_time=ps.get("end", 0.05),
                schedule_linear_start_weight=ps.get("w0", 1.0), schedule_linear_end_weight=ps.get("w1", 0.0),
                schedule_linear_end_time=ps.get("t1", 0.5), schedule_exp_decay_rate=ps.get("rate", 10.0))


def test_get_lp_strength_bit_exact_vs_reference(golden_dir, capsys):
    with open(os.path.join(golden_dir, "schedule_tables.json")) as f:
        cases = json.load(f)
    for c in cases:
        for i, hx in enumerate(c["strength_hex"]):
            assert float(lp_utils.get_lp_strength(**_args(c["params"], i, c["total_steps"]))).hex() == hx
    capsys.readouterr()
    lp_utils.get_lp_strength(**_args(dict(kind="bogus"), 0, 5))
    assert "Unknown lp_strength_schedule_type" in capsys.readouterr().out
    lp_utils.get_lp_strength(**_args(dict(kind="exponential", rate=-1.0), 1, 5))
    assert "Negative exponential_decay_rate" in capsys.readouterr().out


def test_apply_low_pass_filter_host_contract():
    x = torch.zeros(1, 2, 3, 4, 5)
    assert lp_utils.apply_low_pass_filter(x, "none", 1.0, 3, 0.5) is x
    assert lp_utils.apply_low_pass_filter(x, "down_up", 1.0, 3, 1.0) is x
    assert lp_utils.apply_low_pass_filter(x, "gaussian_blur", 0, 3, 0.5) is x
    assert lp_utils.apply_low_pass_filter(x, "median", 1.0, 3, 0.5) is x  # unknown filter: untouched, like the reference
    with pytest.raises(RuntimeError):  # non-contiguous 5-D input: same failure as the reference's .view (lp:35)
        lp_utils.apply_low_pass_filter(torch.zeros(1, 3, 2, 4, 5).permute(0, 2, 1, 3, 4), "down_up", 0.0, 0, 0.5)
    with pytest.raises(ValueError):
        lp_utils.apply_low_pass_filter(torch.zeros(1, 3, 8, 8), "gaussian_blur", -1.0, 3, 0.5)
    with pytest.raises(RuntimeError):  # reflect pad larger than the plane (quirk a-Q3: 15*1.0 is a float -> 15*H)
        lp_utils.apply_low_pass_filter(torch.zeros(1, 3, 60, 90), "gaussian_blur", 2.0, 15 * 1.0, 0.5)
    # no CPU fallback: a CPU tensor that needs real filtering fails loudly
    with pytest.raises(alg_amd.AlgHipError):
        lp_utils.apply_low_pass_filter(torch.zeros(1, 3, 8, 8), "down_up", 0.0, 0, 0.5)


def test_hunyuan_buckets_vs_reference(golden_dir):
    with open(os.path.join(golden_dir, "lp_misc.json")) as f:
        misc = json.load(f)

    class Img:
        def __init__(self, wh):
            self.size = tuple(wh)

    for b in misc["hunyuan_buckets"]:
        assert lp_utils.get_hunyuan_video_size(b["resolution"], Img(b["image_wh"])) == (b["height"], b["width"])
    assert [list(p) for p in lp_utils._generate_crop_size_list(480, 32)] == misc["crop_size_list_480_32"]
    with pytest.raises(NameError):
        lp_utils.get_hunyuan_video_size("1080p", Img((832, 480)))


def test_scheduler_tables_vs_oracle():
    for steps in (1, 2, 7, 50):
        s, o = CogVideoXDDIMScheduler(), ddim_oracle.DDIMOracle()
        s.set_timesteps(steps)
        o.set_timesteps(steps)
        assert torch.equal(s.timesteps, o.timesteps)
        for t in s.timesteps:
            a = s.step_coefficients(t)
            b = tuple(float(v) for v in o.coefficients(t))
            assert a == b
    s = CogVideoXDDIMScheduler()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert ts[0] == 999 and ts[1] == 979 and ts[-1] == 19 and len(ts) == 50  # 'trailing' spacing
    assert s.alphas_cumprod[-1].item() == 0.0  # zero terminal SNR
    assert s.init_noise_sigma == 1.0 and s.order == 1
    with pytest.raises(ValueError):
        CogVideoXDDIMScheduler().step_coefficients(10)


def test_rotary_tables_vs_oracle():
    cfg = dit_oracle.DiTConfig()
    cos, sin = dit_oracle.rope_tables(cfg, 480, 720, 13)
    crops = get_resize_crop_region_for_grid((30, 45), 45, 30)
    assert crops == ((0, 0), (30, 45))
    c2, s2 = rotary_tables(64, crops, (30, 45), 13)
    assert cos.shape == (17550, 64) and torch.equal(cos, c2) and torch.equal(sin, s2)
    assert torch.equal(cos[:, 0::2], cos[:, 1::2])  # repeat-interleaved pairs
    small = dit_oracle.DiTConfig(sample_height=32, sample_width=32, sample_frames=9)
    c3, _ = dit_oracle.rope_tables(small, 256, 256, 3)
    assert c3.shape == (3 * 16 * 16, 64)


def test_param_count_and_shapes():
    cfg = CogVideoXTransformerConfig()
    n = W.count_parameters(cfg)
    assert abs(n - 5.55e9) / 5.55e9 < 0.02, n  # "5B" sanity (SURVEY a-6)
    assert n == dit_oracle.count_params(dit_oracle.DiTConfig())
    assert W.parameter_shapes(cfg) == dit_oracle.param_shapes(dit_oracle.DiTConfig())
    tokens = 13 * 30 * 45 + 226
    assert tokens == 17776
    assert abs(dit_oracle.flops_per_forward(dit_oracle.DiTConfig(), tokens) - 3.322e14) / 3.322e14 < 0.01


def test_cogvideox_1_5_shapes_and_rope():
    """CogVideoX1.5-5B-I2V: patch_size_t 2 (Linear patch embed over c*p_t*p*p = 256 inputs, proj_out 128 wide), ofs embedding,
    no learned positions, 81 frames @ 768x1360 -> 22 padded latent frames -> 11 x 48 x 85 tokens on the slice rotary grid."""
    kw = dict(patch_size_t=2, ofs_embed_dim=512, use_learned_positional_embeddings=False, sample_height=96,
              sample_width=170, sample_frames=81)
    cfg, ocfg = CogVideoXTransformerConfig(**kw), dit_oracle.DiTConfig(**kw)
    shapes = W.parameter_shapes(cfg)
    assert shapes == dit_oracle.param_shapes(ocfg)
    assert shapes["patch_embed.proj.weight"] == (3072, 256) and shapes["proj_out.weight"] == (128, 3072)
    assert shapes["ofs_embedding.linear_1.weight"] == (512, 512) and "patch_embed.pos_embedding" not in shapes
    cos, sin = dit_oracle.rope_tables(ocfg, 768, 1360, 22)
    assert cos.shape == (11 * 48 * 85, 64)
    from alg_amd.pipeline_cogvideox_image2video_lowpass import rotary_tables
    c2, s2 = rotary_tables(64, None, (48, 85), 11, max_size=(48, 85))
    assert torch.equal(c2, cos) and torch.equal(s2, sin)


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "alg_hip.h")).read()
    declared = set(re.findall(r"\b(alg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(alg_amd._lib.EXPORTS), declared ^ set(alg_amd._lib.EXPORTS)
    lib = alg_amd.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.alg_version() == 110
    assert ctypes.sizeof(alg_amd._lib.GemmArgs) == 6 * 8 + 9 * 8 + 7 * 4 + 4 + 8 + 8 + 4 * 8 + 16  # struct alg_gemm_args (+pad, gate_seg_stride, perm_col0/conv_cin_log2, fp8 scales, conv_wp/conv_hpwp/conv_kw/reserved)


def test_c_abi_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it is testable on CPU."""
    lib = alg_amd.load_library()
    rc = lib.alg_down_up(None, None, 1, 4, 4, 2, 2, 0, 0, None, None, 0, None)
    assert rc == -1 and b"alg_down_up" in lib.alg_last_error()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.alg_gaussian_blur(p, ctypes.c_void_p(p.value + 4), 1, 4, 4, 4, 1.0, 0, None, 0, None) == -1  # even kernel
    assert b"odd" in lib.alg_last_error()
    assert lib.alg_gaussian_blur(p, ctypes.c_void_p(p.value + 4), 1, 4, 4, 9, 1.0, 0, None, 0, None) == -1  # pad >= plane
    assert lib.alg_gaussian_blur(p, ctypes.c_void_p(p.value + 4), 1, 4, 4, 3, 0.0, 0, None, 0, None) == -1  # sigma
    assert lib.alg_cfg_ddim_step(p, 0, p, 0, 4, 16, 1.0, 1.0, 1.0, 1.0, 1.0, None) == -1
    assert lib.alg_down_up(p, p, 1, 4, 4, 2, 2, 0, 0, None, None, 0, None) == -1  # aliasing
    assert lib.alg_down_up(p, ctypes.c_void_p(p.value + 4), 0, 4, 4, 2, 2, 0, 0, None, None, 0, None) == 0  # empty input is fine
    # the library never allocates: sizes of caller-owned buffers are plain host queries
    assert lib.alg_lowpass_tables_bytes(60, 90, 15, 22) > 0 and lib.alg_lowpass_tables_bytes(60, 90, 15, 22) % 16 == 0
    assert lib.alg_down_up_workspace_bytes(208, 60, 90, 15, 22) == 0                       # latent planes live in LDS
    assert lib.alg_down_up_workspace_bytes(3, 480, 720, 120, 180) == 4 * 3 * (480 * 180 + 120 * 180 + 120 * 720)
    assert lib.alg_gaussian_blur_workspace_bytes(420, 60, 104, 9) == 0
    assert lib.alg_gaussian_blur_workspace_bytes(3, 480, 720, 9) == 4 * (3 * 480 * 720 + 256)
    # a pixel-sized plane without its workspace is an argument error, not an allocation
    big = (ctypes.c_float * 1)()
    pb = ctypes.cast(big, ctypes.c_void_p)
    assert lib.alg_down_up(pb, ctypes.c_void_p(pb.value + 64), 3, 480, 720, 120, 180, 0, 0, None, None, 0, None) == -1
    assert b"workspace" in lib.alg_last_error()
    assert lib.alg_lowpass_tables_build(None, 0, 60, 90, 15, 22, None) == -1
    # C2's attention launch (2 samples x 48 heads x 17,776 tokens) has a split-KV tail: 8 XCDs x 8 units x 8 chunks x 256 rows
    ws = lib.alg_flash_attn_d64_workspace_bytes(2, 48, 17776, 1)
    assert ws == 8 * 8 * 8 * 256 * 66 * 4, ws
    assert lib.alg_flash_attn_d64_workspace_bytes(1, 8, 1024, 0) == 0


class _FakeTransformer:
    dtype = torch.bfloat16
    config = CogVideoXTransformerConfig()


def test_check_inputs_errors():
    pipe = CogVideoXImageToVideoPipeline(transformer=_FakeTransformer(), scheduler=CogVideoXDDIMScheduler())
    ok = dict(image=torch.zeros(1, 3, 8, 8), prompt="a", height=480, width=720, negative_prompt=None,
              callback_on_step_end_tensor_inputs=["latents"])
    pipe.check_inputs(**ok)
    for bad, msg in ((dict(image=3), "`image` has to be of type"), (dict(height=481), "divisible by 8"),
                     (dict(callback_on_step_end_tensor_inputs=["x"]), "callback_on_step_end_tensor_inputs"),
                     (dict(prompt=None), "Provide either `prompt`"), (dict(prompt=7), "`prompt` has to be of type"),
                     (dict(prompt_embeds=torch.zeros(1, 2, 3)), "Cannot forward both `prompt`")):
        with pytest.raises(ValueError, match=re.escape(msg)):
            pipe.check_inputs(**{**ok, **bad})
    with pytest.raises(ValueError, match="must have the same shape"):
        pipe.check_inputs(**{**ok, "prompt": None, "prompt_embeds": torch.zeros(1, 2, 3),
                             "negative_prompt_embeds": torch.zeros(1, 3, 3)})
    # HIP-only: running the sampler on a CPU device fails loudly instead of falling back
    pipe.to("cpu")
    with pytest.raises(alg_amd.AlgHipError):
        pipe(image=None, image_latents=torch.zeros(1, 1, 16, 60, 90), prompt_embeds=torch.zeros(1, 226, 4096),
             negative_prompt_embeds=torch.zeros(1, 226, 4096), output_type="latent")


def test_loop_branch_table_oracle():
    """The reference loop's 2-/3-pass structure (102 forwards at C2, 5 at C1) from the oracle loop
    with a stand-in transformer."""
    def run(steps, **kw):
        calls = []

        def tf(x, emb, ts, rope):
            calls.append((x.shape[0], emb.shape[0]))
            return torch.zeros(x.shape[0], x.shape[1], x.shape[2] // 2, *x.shape[3:])

        trace = []
        lat = torch.zeros(1, 3, 4, 8, 8)
        cond = torch.randn(1, 3, 4, 8, 8)
        pe, ne = torch.zeros(1, 5, 16), torch.zeros(1, 5, 16)
        loop_oracle.alg_denoise_loop(tf, ddim_oracle.DDIMOracle(), lat, cond, pe, ne, steps, trace=trace, **kw)
        return trace, calls

    trace, calls = run(50, lp_strength_schedule_type="interval", schedule_interval_end_time=0.04)
    assert sum(n for _, _, n in trace) == 102 and [tp for _, tp, _ in trace[:3]] == [False, False, True]
    assert all(a == b for a, b in calls)
    trace, _ = run(2, lp_strength_schedule_type="interval", schedule_interval_end_time=0.04)
    assert [n for _, _, n in trace] == [3, 2]
    trace, _ = run(40, lp_filter_type="gaussian_blur", lp_blur_kernel_size=3, lp_strength_schedule_type="linear")
    assert sum(n for _, _, n in trace) == 100
    trace, _ = run(50, lp_strength_schedule_type="exponential")
    assert sum(1 for _, tp, _ in trace if not tp) == 12
    trace, _ = run(10, use_low_pass_guidance=False)
    assert all(n == 2 for _, _, n in trace)


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """The drop-in boundary is a C ABI: include/alg_hip.h must compile as C (no C++/torch types) and a C program must
    link against libalg_hip.so and resolve every declared entry point (no compute call: there is no GPU here)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    header = open(os.path.join(ROOT, "include", "alg_hip.h")).read()
    names = sorted(set(re.findall(r"\b(alg_[a-z0-9_]+)\s*\(", header)))
    src = tmp_path / "abi.c"
    src.write_text('#include "alg_hip.h"\n#include <stdio.h>\nint main(void) {\n  const void* fns[] = {%s};\n'
                   '  printf("%%d %%d\\n", (int)(sizeof(fns) / sizeof(fns[0])), alg_version());\n  return fns[0] == 0;\n}\n'
                   % ", ".join("(const void*)%s" % n for n in names))
    lib_dir = os.path.join(ROOT, "alg_amd")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-lalg_hip", "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == len(names) and int(out[1]) > 0


def test_dpm_scheduler_scalars_match_the_oracle():
    """CogVideoXDPMScheduler (the second step signature, cog:1114-1122): host multipliers vs the oracle's, and the
    first / last step fall back to first order."""
    from alg_amd.schedulers import CogVideoXDPMScheduler
    from oracle.ddim_oracle import DPMOracle
    p, o = CogVideoXDPMScheduler(), DPMOracle()
    p.set_timesteps(50)
    o.set_timesteps(50)
    assert torch.equal(p.timesteps, o.timesteps)
    ts = o.timesteps
    for i in (0, 1, 2, 25, 48):
        a = o.multipliers(ts[i], ts[i - 1] if i else None)
        b = p.multipliers(ts[i], ts[i - 1] if i else None)
        assert a[7] == b[7]
        for x, y in zip(a[:7], b[:7]):
            assert (x is None and y is None) or abs(float(x) - y) <= 1e-12 * max(1.0, abs(y))
    assert p.multipliers(ts[0], None)[4] is None                      # no history on the first step
    assert p.multipliers(ts[49], ts[48])[7] < 0                       # last step: prev_timestep < 0 -> first order
    x = torch.randn(1, 2, 3, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    nxt, x0 = o.step(torch.zeros_like(x), None, ts[49], ts[48], x, generator=g)
    assert torch.allclose(nxt, x0)                                    # alpha_prev = 1: the last step returns x0 (no noise)


def test_output_writer_round_trips(tmp_path):
    """run:121-133: npy exact, Motion-JPEG AVI decodes back close to the frames, PNG directory exact, mp4 = an H.264 (all
    I_PCM) track whose YUV 4:2:0 samples are exactly the BT.601 conversion of the frames (Motion-JPEG-in-mp4 as an option)."""
    from alg_amd import video_io
    yy, xx = np.mgrid[0:48, 0:64]
    frames = np.stack([np.stack([(xx * 4 + 8 * t) % 256, (yy * 5) % 256, (xx + yy + t) % 256], -1) for t in range(5)]).astype(np.uint8)
    p = video_io.write_video(str(tmp_path / "v.npy"), frames)
    assert np.array_equal(np.load(p), frames)
    p = video_io.write_video(str(tmp_path / "v.avi"), torch.from_numpy(frames), fps=8)
    back = video_io.read_mjpeg_avi(p)
    assert back.shape == frames.shape
    mse = ((back.astype(np.float32) - frames.astype(np.float32)) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / mse) > 28.0
    d = video_io.write_video(str(tmp_path / "frames"), frames)
    from PIL import Image
    assert sorted(os.listdir(d))[0] == "frame_00000.png" and np.array_equal(np.asarray(Image.open(os.path.join(d, "frame_00003.png"))), frames[3])
    video_io.write_video(str(tmp_path / "v.gif"), frames)
    # mp4 / h264: container + bitstream walked by an independent parser; the stored samples are the exact 4:2:0 conversion
    smooth = np.stack([np.stack([(xx * 3 + 5 * t) % 256, 255 - yy * 4, (xx + 2 * yy) % 256], -1) for t in range(3)]).astype(np.uint8)
    p = video_io.write_video(str(tmp_path / "v.mp4"), smooth, fps=16)
    back, info = video_io.read_mp4(p)
    assert (info["codec"], info["fps"], info["frames"], info["profile"], info["cropped"]) == ("avc1", 16.0, 3, 66, (64, 48))
    for a_, b_ in zip(info["yuv"], video_io.rgb_to_yuv420(smooth)):
        assert np.array_equal(a_, b_)
    assert back.shape == smooth.shape
    mse = ((back.astype(np.float32) - smooth.astype(np.float32)) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / mse) > 30.0            # what 4:2:0 subsampling costs on this pattern
    data = open(p, "rb").read()
    assert data[4:8] == b"ftyp" and b"avcC" in data and b"moov" in data
    mdat = data[data.index(b"mdat") + 4:data.index(b"moov") - 4]
    assert b"\x00\x00\x00" not in mdat[4:] and b"\x00\x00\x01" not in mdat[4:] and b"\x00\x00\x02" not in mdat[4:]
    # a size that is not a whole number of macroblocks: padded by edge replication, cropped again by the SPS
    odd = smooth[:, :36, :50]
    back, info = video_io.read_mp4(video_io.write_video(str(tmp_path / "odd.mp4"), odd, fps=8))
    assert info["cropped"] == (50, 36) and back.shape == odd.shape
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(info["yuv"], video_io.rgb_to_yuv420(odd)))
    # black / white frames produce runs of equal bytes: emulation prevention must keep start codes out of the payload
    flat = np.zeros((2, 32, 32, 3), np.uint8)
    flat[1] = 255
    back, info = video_io.read_mp4(video_io.write_video(str(tmp_path / "flat.mp4"), flat))
    assert np.abs(back.astype(int) - flat.astype(int)).max() <= 1
    back, info = video_io.read_mp4(video_io.write_mp4(str(tmp_path / "j.mp4"), smooth, fps=8, codec="mjpeg"))
    assert info["codec"] == "mp4v" and back.shape == smooth.shape
    with pytest.raises(ValueError, match="uint8"):
        video_io.write_video(str(tmp_path / "w.npy"), frames.astype(np.float32))
