"""Drop-in mirror of the reference's ``lp_utils`` API (boundary b-3 of SURVEY.md section 8):

    apply_low_pass_filter(tensor, filter_type, blur_sigma, blur_kernel_size, resize_factor) -> Tensor
    get_lp_strength(step_index, total_steps, lp_strength_schedule_type, ...) -> float
    get_hunyuan_video_size(i2v_resolution, input_image) -> (height, width)

Same names, argument meaning and error behaviour as /root/reference/lp_utils.py; the filters run as
hand-written gfx950 kernels (alg_amd/csrc/lowpass.hip) through the C ABI.  Host-side logic here is limited
to what the reference also does on the host: the identity exits, the 5-D -> plane view, the kernel-size and
target-size arithmetic (Python rounding), and the scalar schedule.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib


def apply_low_pass_filter(tensor, filter_type, blur_sigma, blur_kernel_size, resize_factor):
    """reference lp_utils.py:8-60.

    * the three no-op cases return the *input object* (lp:23-28);
    * 4-D ``[B,C,H,W]`` / 5-D ``[B,C,K,H,W]`` tensors are filtered per (H, W) plane; like the reference's
      ``.view`` (lp:35) a non-contiguous 5-D tensor raises ``RuntimeError``;
    * ``gaussian_blur``: a ``float`` kernel size is a fraction of the plane height, an ``int`` is absolute,
      even sizes are bumped to the next odd (lp:41-46); sigma <= 0 (other than the ==0 exit) raises like
      torchvision does;
    * ``down_up``: target size ``max(1, int(round(size * factor)))`` with Python's banker's rounding (lp:51-52).
    """
    if filter_type == "none":
        return tensor
    if filter_type == "down_up" and resize_factor == 1.0:
        return tensor
    if filter_type == "gaussian_blur" and blur_sigma == 0:
        return tensor

    if tensor.ndim == 5:
        if not tensor.is_contiguous():
            # same failure mode as tensor.view(B*K, C, H, W) on a permuted tensor (lp:35)
            raise RuntimeError("view size is not compatible with input tensor's size and stride "
                               "(apply_low_pass_filter needs a contiguous 5-D tensor)")
    elif tensor.ndim != 4:
        raise ValueError("not enough values to unpack (expected a 4-D or 5-D tensor, got %d-D)" % tensor.ndim)
    H, W = tensor.shape[-2:]

    if filter_type == "gaussian_blur":
        if isinstance(blur_kernel_size, float):
            ksize = max(int(blur_kernel_size * H), 1)
        else:
            ksize = int(blur_kernel_size)
        if ksize % 2 == 0:
            ksize += 1
        if blur_sigma <= 0:
            raise ValueError("sigma should have positive values. Got %s" % blur_sigma)
        if ksize // 2 >= min(H, W):
            raise RuntimeError("Padding size should be less than the corresponding input dimension "
                               "(kernel %d on a %dx%d plane)" % (ksize, H, W))
        src = tensor if tensor.is_contiguous() else tensor.contiguous()
        return _lib.gaussian_blur(src, ksize, float(blur_sigma))

    if filter_type == "down_up":
        h1 = max(1, int(round(H * resize_factor)))
        w1 = max(1, int(round(W * resize_factor)))
        src = tensor if tensor.is_contiguous() else tensor.contiguous()
        return _lib.down_up(src, h1, w1, round_intermediate=True)

    return tensor  # unknown filter names fall through untouched, as in the reference (no else branch)


def get_lp_strength(
    step_index,
    total_steps,
    lp_strength_schedule_type,
    schedule_interval_start_time,
    schedule_interval_end_time,
    schedule_linear_start_weight,
    schedule_linear_end_weight,
    schedule_linear_end_time,
    schedule_exp_decay_rate,
):
    """reference lp_utils.py:63-111 -- float64 host scalar, bit-exact (it drives ``== 0`` branch tests)."""
    t = step_index / max(total_steps - 1, 1)
    kind = lp_strength_schedule_type
    if kind == "interval":
        inside = schedule_interval_start_time <= t <= schedule_interval_end_time
        return 1.0 if inside else 0.0
    if kind == "linear":
        span = schedule_linear_end_time
        if span <= 0:
            return schedule_linear_start_weight
        if t >= span:
            return schedule_linear_end_weight
        frac = t / span
        return schedule_linear_start_weight * (1 - frac) + schedule_linear_end_weight * frac
    if kind == "exponential":
        rate = schedule_exp_decay_rate
        if rate < 0:
            print(f"Warning: Negative exponential_decay_rate ({rate}) is unusual. Using abs value.")
            rate = abs(rate)
        return math.exp(-rate * t)
    if kind == "none":
        return 1.0
    print(f"Warning: Unknown lp_strength_schedule_type '{kind}'. Using constant strength 1.0.")
    return 1.0


# -- HunyuanVideo resolution buckets (reference lp_utils.py:113-189); host-only ------------------------

def _generate_crop_size_list(base_size=256, patch_size=32, max_ratio=4.0):
    budget = round((base_size / patch_size) ** 2)
    assert max_ratio >= 1.0
    sizes = []
    wp, hp = budget, 1
    while wp > 0:
        if max(wp, hp) / min(wp, hp) <= max_ratio:
            sizes.append((wp * patch_size, hp * patch_size))
        if (hp + 1) * wp <= budget:
            hp += 1
        else:
            wp -= 1
    return sizes


def _get_closest_ratio(height, width, ratios, buckets):
    aspect = float(height) / float(width)
    deltas = ratios - aspect
    if aspect >= 1:
        pool = [(i, d) for i, d in enumerate(deltas) if d <= 0]
    else:
        pool = [(i, d) for i, d in enumerate(deltas) if d > 0]
    best = min(pool, key=lambda item: abs(item[1]))[0]
    return buckets[best], ratios[best]


def get_hunyuan_video_size(i2v_resolution, input_image):
    base = {"720p": 960, "540p": 720, "360p": 480}.get(i2v_resolution)
    if base is None:
        # the reference leaves bucket_hw_base_size unbound here -> UnboundLocalError (a NameError subclass)
        raise UnboundLocalError("unknown i2v_resolution %r (expected '720p', '540p' or '360p')" % (i2v_resolution,))
    width, height = input_image.size
    buckets = _generate_crop_size_list(base, 32)
    ratios = np.array([round(float(h) / float(w), 5) for h, w in buckets])
    (target_h, target_w), _ = _get_closest_ratio(height, width, ratios, buckets)
    return target_h, target_w
