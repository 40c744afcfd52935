"""CPU oracle for the ALG low-pass filters and strength schedule (numpy only).

TEST INFRASTRUCTURE -- never imported by ``alg_amd`` (see oracle/__init__.py).

Restates, function by function:
  * /root/reference/lp_utils.py:8-60    apply_low_pass_filter
  * /root/reference/lp_utils.py:63-111  get_lp_strength
  * /root/reference/lp_utils.py:113-189 HunyuanVideo resolution buckets
and the third-party arithmetic those call sites reach:
  * ATen ``_upsample_bilinear2d_aa`` (lp_utils.py:53-54), restated from its
    published algorithm (separable antialiased triangle filter, W pass then H
    pass); pinned against ATen's own CPU kernel via the imported reference.
  * torchvision ``gaussian_blur`` (lp_utils.py:47), restated from its published
    algorithm; torchvision is absent here -> parity unpinned at that boundary.
"""
from __future__ import annotations

import math

import numpy as np

# ----------------------------------------------------------------------------
# schedule  (lp_utils.py:63-111) -- float64 scalar arithmetic, bit-exact
# ----------------------------------------------------------------------------


def get_lp_strength(
    step_index,
    total_steps,
    lp_strength_schedule_type,
    schedule_interval_start_time=0.0,
    schedule_interval_end_time=0.05,
    schedule_linear_start_weight=1.0,
    schedule_linear_end_weight=0.0,
    schedule_linear_end_time=0.5,
    schedule_exp_decay_rate=10.0,
):
    """lp_utils.py:81 step_norm; :83-92 linear; :94-98 interval; :100-105 exponential;
    :107-111 none / unknown -> 1.0."""
    step_norm = step_index / max(total_steps - 1, 1)
    kind = lp_strength_schedule_type
    if kind == "linear":
        if schedule_linear_end_time <= 0:
            return schedule_linear_start_weight
        if step_norm >= schedule_linear_end_time:
            return schedule_linear_end_weight
        progress = step_norm / schedule_linear_end_time
        return schedule_linear_start_weight * (1 - progress) + schedule_linear_end_weight * progress
    if kind == "interval":
        return 1.0 if schedule_interval_start_time <= step_norm <= schedule_interval_end_time else 0.0
    if kind == "exponential":
        rate = abs(schedule_exp_decay_rate)
        return math.exp(-rate * step_norm)
    return 1.0


def modulated_params(strength, lp_blur_sigma, lp_blur_kernel_size, lp_resize_factor, schedule_blur_kernel_size):
    """pipeline_cogvideox_image2video_lowpass.py:1034-1040 (same in wan:861-867, hy:1145-1151)."""
    sigma = lp_blur_sigma * strength
    ksize = lp_blur_kernel_size * strength if schedule_blur_kernel_size else lp_blur_kernel_size
    factor = 1.0 - (1.0 - lp_resize_factor) * strength
    return sigma, ksize, factor


def two_pass_flag(strength, schedule_type, use_low_pass_guidance=True):
    """pipeline_cogvideox_image2video_lowpass.py:1029-1032."""
    two_pass = strength == 0 or not use_low_pass_guidance
    if schedule_type == "exponential" and strength < 0.1:
        two_pass = True
    return bool(two_pass)


# ----------------------------------------------------------------------------
# down_up  (lp_utils.py:49-54)
# ----------------------------------------------------------------------------


def down_up_size(h0, w0, resize_factor):
    """lp_utils.py:51-52 -- Python round() is banker's rounding (90*0.25=22.5 -> 22)."""
    h1 = max(1, int(round(h0 * resize_factor)))
    w1 = max(1, int(round(w0 * resize_factor)))
    return h1, w1


def aa_taps(in_size, out_size, ftype=np.float64):
    """Per-output-index tap table of ATen's antialiased bilinear 1-D resize
    (align_corners=False), in the arithmetic type ``ftype`` (ATen uses float for
    float/bf16 tensors, double for double).

    Returns (xmin[int64 out], xsize[int64 out], w[out, max_taps] ftype, zero padded).
    """
    ft = ftype
    scale = ft(in_size) / ft(out_size)
    support = ft(scale) if scale >= 1.0 else ft(1.0)  # interp_size/2 * scale, interp_size=2
    invscale = ft(1.0) / scale if scale >= 1.0 else ft(1.0)
    max_taps = int(math.ceil(float(support))) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    xsize = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, max_taps), ft)
    for i in range(out_size):
        center = ft(scale * ft(i + 0.5))
        lo = max(int(ft(center - support + ft(0.5))), 0)
        hi = min(int(ft(center + support + ft(0.5))), in_size)
        n = min(max(hi - lo, 0), max_taps)
        tot = ft(0.0)
        for j in range(n):
            x = ft(ft(j + lo) - center + ft(0.5)) * invscale
            wj = max(ft(0.0), ft(1.0) - abs(x))
            w[i, j] = wj
            tot = ft(tot + wj)
        if tot != 0:
            w[i, :n] = w[i, :n] / tot
        xmin[i], xsize[i] = lo, n
    return xmin, xsize, w


def resize_matrix(in_size, out_size, ftype=np.float64):
    """Dense [out, in] matrix of the 1-D antialiased bilinear resize."""
    xmin, xsize, w = aa_taps(in_size, out_size, ftype)
    m = np.zeros((out_size, in_size), ftype)
    for i in range(out_size):
        m[i, xmin[i]: xmin[i] + xsize[i]] = w[i, : xsize[i]]
    return m


def _resize_last(x, out_size, ftype):
    m = resize_matrix(x.shape[-1], out_size, ftype)
    return np.einsum("...w,ow->...o", x.astype(ftype, copy=False), m)


def resize_aa(x, h1, w1, ftype=np.float64):
    """F.interpolate(x, (h1, w1), mode='bilinear', align_corners=False, antialias=True)
    on the last two axes: W pass first, then H pass (ATen's separable order)."""
    y = _resize_last(x, w1, ftype)
    y = np.swapaxes(_resize_last(np.swapaxes(y, -1, -2), h1, ftype), -1, -2)
    return y


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (numpy has no bf16 dtype)."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32)
    return np.where(np.isnan(a), a, out).reshape(a.shape)


def down_up(x, resize_factor, ftype=np.float64, storage="f32"):
    """lp_utils.py:49-54 on the last two axes of ``x``.

    storage='f32'  : intermediate and result kept in ``ftype`` (the fp32/fp64 CPU path).
    storage='bf16' : the reference's bf16 GPU path -- each F.interpolate call returns a
                     bf16 tensor, so the (h1, w1) intermediate and the result are rounded
                     to bf16 (fp32 accumulation inside each call).
    """
    h0, w0 = x.shape[-2:]
    h1, w1 = down_up_size(h0, w0, resize_factor)
    y = resize_aa(x, h1, w1, ftype)
    if storage == "bf16":
        y = bf16_round(y)
    z = resize_aa(y, h0, w0, ftype)
    if storage == "bf16":
        z = bf16_round(z)
    return z


# ----------------------------------------------------------------------------
# gaussian_blur  (lp_utils.py:40-47; torchvision.transforms.functional.gaussian_blur)
# ----------------------------------------------------------------------------


def gaussian_kernel_size(blur_kernel_size, height):
    """lp_utils.py:41-46: float -> fraction of plane height, int -> absolute; forced odd."""
    if isinstance(blur_kernel_size, float):
        k = max(int(blur_kernel_size * height), 1)
    else:
        k = int(blur_kernel_size)
    if k % 2 == 0:
        k += 1
    return k


def gaussian_kernel1d(ksize, sigma, ftype=np.float64):
    """torchvision _get_gaussian_kernel1d: x = linspace(-(k-1)/2, (k-1)/2, k);
    pdf = exp(-0.5 (x/sigma)^2); pdf / pdf.sum()."""
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize).astype(ftype)
    pdf = np.exp(-0.5 * (x / ftype(sigma)) ** 2).astype(ftype)
    return (pdf / pdf.sum()).astype(ftype)


def gaussian_blur(x, ksize, sigma, ftype=np.float64):
    """Reflect-pad k//2 (no edge repeat) then correlate with g (x) g on the last two axes."""
    if sigma <= 0:
        raise ValueError("sigma should be positive")  # torchvision raises for sigma<=0
    h, w = x.shape[-2:]
    pad = ksize // 2
    if pad >= h or pad >= w:
        raise ValueError("reflect padding needs k//2 < min(H, W)")
    g = gaussian_kernel1d(ksize, sigma, ftype)
    a = x.astype(ftype, copy=False)
    cfg = [(0, 0)] * (a.ndim - 2) + [(pad, pad), (pad, pad)]
    ap = np.pad(a, cfg, mode="reflect")
    # W pass
    t = np.zeros(ap.shape[:-1] + (w,), ftype)
    for j in range(ksize):
        t += g[j] * ap[..., j: j + w]
    # H pass
    y = np.zeros(a.shape, ftype)
    for i in range(ksize):
        y += g[i] * t[..., i: i + h, :]
    return y


# ----------------------------------------------------------------------------
# apply_low_pass_filter  (lp_utils.py:8-60)
# ----------------------------------------------------------------------------


def apply_low_pass_filter(x, filter_type, blur_sigma, blur_kernel_size, resize_factor, ftype=np.float64,
                          storage="f32"):
    """numpy restatement; returns the SAME object on the three identity exits (lp:23-28).
    4-D [B,C,H,W] and 5-D [B,C,K,H,W] are both filtered per (H, W) plane (lp:31-37 is a
    pure view because both filters are per-plane and channel independent)."""
    if filter_type == "none":
        return x
    if filter_type == "down_up" and resize_factor == 1.0:
        return x
    if filter_type == "gaussian_blur" and blur_sigma == 0:
        return x
    if x.ndim not in (4, 5):
        raise ValueError("expected a 4-D or 5-D tensor")
    if filter_type == "gaussian_blur":
        k = gaussian_kernel_size(blur_kernel_size, x.shape[-2])
        y = gaussian_blur(x, k, blur_sigma, ftype)
        return bf16_round(y) if storage == "bf16" else y
    if filter_type == "down_up":
        return down_up(x, resize_factor, ftype, storage)
    return x  # unknown filter types fall through untouched (lp:40-54 has no else branch)


# ----------------------------------------------------------------------------
# HunyuanVideo buckets  (lp_utils.py:113-189)
# ----------------------------------------------------------------------------


def generate_crop_size_list(base_size=256, patch_size=32, max_ratio=4.0):
    """lp_utils.py:113-136."""
    num_patches = round((base_size / patch_size) ** 2)
    assert max_ratio >= 1.0
    out = []
    wp, hp = num_patches, 1
    while wp > 0:
        if max(wp, hp) / min(wp, hp) <= max_ratio:
            out.append((wp * patch_size, hp * patch_size))
        if (hp + 1) * wp <= num_patches:
            hp += 1
        else:
            wp -= 1
    return out


def get_closest_ratio(height, width, ratios, buckets):
    """lp_utils.py:138-161."""
    aspect = float(height) / float(width)
    diff = ratios - aspect
    if aspect >= 1:
        cand = [(i, d) for i, d in enumerate(diff) if d <= 0]
    else:
        cand = [(i, d) for i, d in enumerate(diff) if d > 0]
    idx = min(cand, key=lambda p: abs(p[1]))[0]
    return buckets[idx], ratios[idx]


def get_hunyuan_video_size(i2v_resolution, image_size_wh):
    """lp_utils.py:163-189; ``image_size_wh`` is PIL's ``image.size`` = (width, height)."""
    base = {"720p": 960, "540p": 720, "360p": 480}[i2v_resolution]
    crops = generate_crop_size_list(base, 32)
    ratios = np.array([round(float(h) / float(w), 5) for h, w in crops])
    (th, tw), _ = get_closest_ratio(image_size_wh[1], image_size_wh[0], ratios, crops)
    return th, tw
