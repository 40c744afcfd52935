"""CPU restatement of the CogVideoX ALG denoising loop
(pipeline_cogvideox_image2video_lowpass.py:1000-1140) and of prepare_lp's latent branch
(cog:682-703), with the DiT and the scheduler injected as callables.

TEST INFRASTRUCTURE -- never imported by ``alg_amd``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import lp_oracle


def apply_low_pass_filter_torch(t, filter_type, blur_sigma, blur_kernel_size, resize_factor):
    """lp_utils.py:8-60 on a torch tensor.  fp32/fp64 down_up goes through the very ATen op the
    reference calls (lp:53-54); bf16 (no CPU kernel) and gaussian_blur (torchvision absent) go
    through oracle.lp_oracle."""
    if filter_type == "none":
        return t
    if filter_type == "down_up" and resize_factor == 1.0:
        return t
    if filter_type == "gaussian_blur" and blur_sigma == 0:
        return t
    shape = t.shape
    if t.ndim == 5:
        B, C, K, H, W = shape
        x = t.view(B * K, C, H, W)  # lp:35 -- raises for non-contiguous input, like the reference
    else:
        x = t
    if filter_type == "down_up" and t.dtype in (torch.float32, torch.float64):
        h0, w0 = x.shape[-2:]
        h1, w1 = lp_oracle.down_up_size(h0, w0, resize_factor)
        x = F.interpolate(x, size=(h1, w1), mode="bilinear", align_corners=False, antialias=True)
        x = F.interpolate(x, size=(h0, w0), mode="bilinear", align_corners=False, antialias=True)
    elif filter_type in ("down_up", "gaussian_blur"):
        bf16 = t.dtype == torch.bfloat16
        a = x.float().numpy()
        y = lp_oracle.apply_low_pass_filter(a, filter_type, blur_sigma, blur_kernel_size, resize_factor,
                                            ftype=np.float32 if t.dtype != torch.float64 else np.float64,
                                            storage="bf16" if bf16 else "f32")
        x = torch.from_numpy(np.ascontiguousarray(y)).to(t.dtype)
    return x.view(shape) if t.ndim == 5 else x


def prepare_lp_latent(image_latents, filter_type, sigma, ksize, factor):
    """cog:684-701: [B,F,C,H,W] -> permute -> contiguous -> filter -> permute back -> contiguous."""
    perm = image_latents.permute(0, 2, 1, 3, 4).contiguous()
    out = apply_low_pass_filter_torch(perm, filter_type, sigma, ksize, factor)
    return out.permute(0, 2, 1, 3, 4).contiguous().to(image_latents.dtype)


def alg_denoise_loop(transformer, scheduler, latents, image_latents, prompt_embeds, negative_prompt_embeds,
                     num_inference_steps, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
                     lp_blur_sigma=15.0, lp_blur_kernel_size=0.02734375, lp_resize_factor=0.25,
                     lp_strength_schedule_type="interval", schedule_blur_kernel_size=False,
                     schedule_interval_start_time=0.0, schedule_interval_end_time=0.05,
                     schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0, schedule_linear_end_time=0.5,
                     schedule_exp_decay_rate=10.0, image_rotary_emb=None, trace=None):
    """Returns final latents [B,F,C,H,W].  ``transformer(hidden_states, encoder_hidden_states, timestep,
    image_rotary_emb)`` -> noise prediction.  ``trace`` (list) receives per-step
    (strength, two_pass, n_forward) for branch-table tests."""
    do_cfg = guidance_scale > 1.0
    dtype = prompt_embeds.dtype
    if do_cfg and use_low_pass_guidance:  # cog:948-951
        pe3 = torch.cat([negative_prompt_embeds, negative_prompt_embeds, prompt_embeds], dim=0)
        pe2 = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
    elif do_cfg:  # cog:952-955
        pe3 = pe2 = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
    else:
        pe3 = pe2 = prompt_embeds
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    for i, t in enumerate(scheduler.timesteps):
        two_pass = True
        if do_cfg and use_low_pass_guidance:
            s = lp_oracle.get_lp_strength(i, num_inference_steps, lp_strength_schedule_type,
                                          schedule_interval_start_time, schedule_interval_end_time,
                                          schedule_linear_start_weight, schedule_linear_end_weight,
                                          schedule_linear_end_time, schedule_exp_decay_rate)
            two_pass = lp_oracle.two_pass_flag(s, lp_strength_schedule_type, use_low_pass_guidance)
            sigma, ksize, factor = lp_oracle.modulated_params(s, lp_blur_sigma, lp_blur_kernel_size,
                                                              lp_resize_factor, schedule_blur_kernel_size)
            lp_lat = prepare_lp_latent(image_latents, lp_filter_type, sigma, ksize, factor)
            n = 2 if two_pass else 3
            x = scheduler.scale_model_input(torch.cat([latents] * n), t)
            cond = [lp_lat, lp_lat] if two_pass else [image_latents, lp_lat, lp_lat]  # cog:1068-1070
            x = torch.cat([x, torch.cat(cond, dim=0)], dim=2)
        elif do_cfg:
            s = None
            x = scheduler.scale_model_input(torch.cat([latents] * 2), t)
            x = torch.cat([x, torch.cat([image_latents] * 2, dim=0)], dim=2)
        else:
            s = None
            x = torch.cat([scheduler.scale_model_input(latents, t), image_latents], dim=2)
        ts = t.expand(x.shape[0])
        pred = transformer(x, pe2 if two_pass else pe3, ts, image_rotary_emb).float()  # cog:1082-1091
        if do_cfg and use_low_pass_guidance and not two_pass:
            u0, u, tx = pred.chunk(3)
            pred = u0 + guidance_scale * (tx - u)  # cog:1099-1102
        elif do_cfg:
            u, tx = pred.chunk(2)
            pred = u + guidance_scale * (tx - u)  # cog:1096-1097 / 1109
        latents = scheduler.step(pred, t, latents).to(dtype)  # cog:1112, 1123
        if trace is not None:
            trace.append((s, two_pass, x.shape[0]))
    return latents
