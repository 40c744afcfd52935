"""CPU restatement of the CogVideoX ALG denoising loop
(pipeline_cogvideox_image2video_lowpass.py:1000-1140) and of prepare_lp's latent branch
(cog:682-703) and pixel branch (cog:628-680; Wan: wan:493-540), with the DiT, the scheduler and
the VAE encoder injected as callables.

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


def gaussian_blur_torch(x, ksize, sigma):
    """torchvision.transforms.functional.gaussian_blur(x, [k, k], [s, s]) as the reference executes it on the CPU (lp:40-47):
    the k x k kernel is the outer product of g = exp(-0.5 (i / s)^2) / sum over i = -(k-1)/2 .. (k-1)/2 (built in x's dtype),
    the planes are reflect-padded by k // 2 and run through ONE depthwise F.conv2d (ATen's threaded CPU convolution).  Used by
    bench.py's cpu_baseline leg (a threaded baseline next to the single-threaded numpy statement in lp_oracle.gaussian_blur,
    which it matches to fp32 rounding: tests/test_oracle_golden.py)."""
    lim = (ksize - 1) * 0.5
    i = torch.linspace(-lim, lim, steps=ksize, dtype=x.dtype)
    g = torch.exp(-0.5 * (i / sigma) ** 2)
    g = g / g.sum()
    k2 = torch.outer(g, g)
    shape = x.shape
    planes = x.reshape(-1, 1, shape[-2], shape[-1])
    pad = ksize // 2
    y = F.conv2d(F.pad(planes.transpose(0, 1), (pad, pad, pad, pad), mode="reflect"),
                 k2[None, None].expand(planes.shape[0], 1, ksize, ksize), groups=planes.shape[0])
    return y.transpose(0, 1).reshape(shape)


def prepare_lp_latent(image_latents, filter_type, sigma, ksize, factor):
    """cog:684-701: [B,F,C,H,W] -> permute -> contiguous -> filter -> permute back -> contiguous."""
    perm = image_latents.permute(0, 2, 1, 3, 4).contiguous()
    out = apply_low_pass_filter_torch(perm, filter_type, sigma, ksize, factor)
    return out.permute(0, 2, 1, 3, 4).contiguous().to(image_latents.dtype)


def posterior_sample(moments, generator, noise_dtype=None):
    """diffusers DiagonalGaussianDistribution(moments).sample(generator): mean / logvar = chunk(2, dim=1), logvar clamped to
    [-30, 20], std = exp(0.5 logvar), noise = randn_tensor(mean.shape, generator, dtype=moments.dtype), mean + std * noise.
    ``noise_dtype`` is the dtype the REFERENCE's VAE runs in (cog: the pipeline dtype, bf16 under run.py:65-69; wan: float32,
    run:51-55): torch's CPU generator gives a different stream per dtype (bf16 randn is not fp32 randn rounded), so an fp32
    oracle run that is to see the product's noise draws it in that dtype and widens it."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    noise = torch.randn(mean.shape, generator=generator, dtype=noise_dtype or moments.dtype).to(moments.dtype)
    return mean + std * noise


def cog_encode_image(image, vae_moments, generator, scaling_factor=0.7, invert_scale_latents=False, noise_dtype=None):
    """cog:388-400 (prepare_latents): image [B, 3, H, W] -> unsqueeze(2) -> vae.encode -> latent_dist.sample(generator) ->
    x scaling_factor (or its inverse) -> [B, 1, C, h, w] (the frame the zero padding of cog:402-411 follows)."""
    z = posterior_sample(vae_moments(image.unsqueeze(2)), generator, noise_dtype)
    z = scaling_factor * z if not invert_scale_latents else 1 / scaling_factor * z
    return z.permute(0, 2, 1, 3, 4)


def prepare_lp_pixel_cog(image_tensor, vae_moments, generator, num_frames, filter_type, sigma, ksize, factor, out_dtype,
                         scaling_factor=0.7, invert_scale_latents=False, temporal_ratio=4, patch_size_t=None,
                         noise_dtype=None):
    """cog:628-680, the `lp_filter_in_latent=False` branch, statement by statement: filter the RGB image [B, 3, H, W]
    (cog:632-638) -> unsqueeze(2) (cog:642) -> vae.encode(...).latent_dist.sample(generator) (cog:645: FRESH posterior noise
    on every call, i.e. every step) -> x scaling factor (cog:647-650) -> permute to [B, 1, C, h, w] (cog:652) -> zero frames up
    to (num_frames - 1) // 4 + 1 (cog:655-671) -> CogVideoX-1.5 leading-frame repeat (cog:673-680) -> cast (cog:701).
    ``vae_moments(x[B,3,1,H,W]) -> [B, 2C, 1, h, w]`` is the VAE oracle's encoder."""
    img = apply_low_pass_filter_torch(image_tensor, filter_type, sigma, ksize, factor)
    enc = cog_encode_image(img, vae_moments, generator, scaling_factor, invert_scale_latents, noise_dtype)
    padded = (num_frames - 1) // temporal_ratio + 1
    if padded > enc.shape[1]:
        b, f, c, h, w = enc.shape
        enc = torch.cat([enc, torch.zeros((b, padded - f, c, h, w), dtype=enc.dtype)], dim=1)
    else:
        enc = enc[:, :padded]
    if patch_size_t is not None and enc.size(1) % patch_size_t:
        n = min(patch_size_t - enc.size(1) % patch_size_t, enc.shape[1])
        enc = torch.cat([enc[:, :n], enc], dim=1)
    return enc.to(out_dtype)


def wan_condition(latent, num_frames, temporal_ratio=4):
    """wan:439-449 / wan:527-538: [mask4 | normalised latent16] -- ones for the first pixel frame, repeated 4 times, folded
    in groups of 4 pixel frames into channels."""
    b, _, _, h, w = latent.shape
    mask = torch.ones(b, 1, num_frames, h, w)
    mask[:, :, list(range(1, num_frames))] = 0
    first = torch.repeat_interleave(mask[:, :, 0:1], dim=2, repeats=temporal_ratio)
    mask = torch.concat([first, mask[:, :, 1:, :]], dim=2)
    mask = mask.view(b, -1, temporal_ratio, h, w).transpose(1, 2)
    return torch.concat([mask.to(latent.dtype), latent], dim=1)


def prepare_lp_pixel_wan(image_tensor, vae_moments, generator, num_frames, filter_type, sigma, ksize, factor, out_dtype,
                         latents_mean, latents_std, temporal_ratio=4, noise_dtype=None):
    """wan:493-540, the `lp_filter_in_latent=False` branch: filter the RGB image (wan:495-501) -> [image_lp, zeros x
    (num_frames - 1)] (wan:509-518) -> vae.encode(...).latent_dist.SAMPLE(generator) (wan:526 -- not the mode the unfiltered
    condition uses at wan:430) -> (z - mean) * (1 / std) (wan:519-527) -> mask channels in front (wan:529-540) -> cast."""
    img = apply_low_pass_filter_torch(image_tensor, filter_type, sigma, ksize, factor)
    x = img.unsqueeze(2)
    video = torch.cat([x, x.new_zeros(x.shape[0], x.shape[1], num_frames - 1, x.shape[3], x.shape[4])], dim=2)
    z = posterior_sample(vae_moments(video), generator, noise_dtype)
    mean = torch.tensor(latents_mean).view(1, -1, 1, 1, 1).to(img.dtype)
    inv_std = 1.0 / torch.tensor(latents_std).view(1, -1, 1, 1, 1).to(img.dtype)
    return wan_condition((z - mean) * inv_std, num_frames, temporal_ratio).to(out_dtype)


def alg_denoise_loop(transformer, scheduler, latents, image_latents, prompt_embeds, negative_prompt_embeds,
                     num_inference_steps, guidance_scale=6.0, use_low_pass_guidance=True, lp_filter_type="down_up",
                     lp_blur_sigma=15.0, lp_blur_kernel_size=0.02734375, lp_resize_factor=0.25,
                     lp_strength_schedule_type="interval", schedule_blur_kernel_size=False,
                     schedule_interval_start_time=0.0, schedule_interval_end_time=0.05,
                     schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0, schedule_linear_end_time=0.5,
                     schedule_exp_decay_rate=10.0, image_rotary_emb=None, trace=None, generator=None, prepare_lp=None):
    """Returns final latents [B,F,C,H,W].  ``transformer(hidden_states, encoder_hidden_states, timestep,
    image_rotary_emb)`` -> noise prediction.  ``trace`` (list) receives per-step
    (strength, two_pass, n_forward) for branch-table tests.  ``prepare_lp(filter_type, sigma, ksize, factor)`` replaces the
    latent-space filter by the pixel branch (cog:1043-1057 calls prepare_lp on EVERY ALG step, so a pixel-branch callable
    consumes its generator once per step)."""
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
    old_pred_original_sample = None                     # cog:998, only used by the DPM scheduler branch
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
            if prepare_lp is None:
                lp_lat = prepare_lp_latent(image_latents, lp_filter_type, sigma, ksize, factor)
            else:
                lp_lat = prepare_lp(lp_filter_type, sigma, ksize, factor)
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
        if hasattr(scheduler, "multipliers"):             # CogVideoXDPMScheduler: the second step signature (cog:1114-1122)
            latents, old_pred_original_sample = scheduler.step(
                pred, old_pred_original_sample, t, scheduler.timesteps[i - 1] if i > 0 else None, latents,
                generator=generator)
            latents = latents.to(dtype)
        else:
            latents = scheduler.step(pred, t, latents).to(dtype)  # cog:1112, 1123
        if trace is not None:
            trace.append((s, two_pass, x.shape[0]))
    return latents


# ---------------------------------------------------------------------------------------------------------------------
# Wan 2.1 and HunyuanVideo ALG loops (pipeline_wan_image2video_lowpass.py:843-927,
# pipeline_hunyuan_video_image2video_lowpass.py:1127-1270) -- torch-CPU restatements with the DiT injected.
# bf16 elementwise ops on CPU round after every op exactly as on the GPU; the only CPU/GPU difference is a 0-dim
# *tensor* scalar times a bf16 tensor (CPU casts the scalar to bf16 first), which the scheduler oracles avoid by
# receiving python floats where the pipelines pass python floats (guidance scales).
# ---------------------------------------------------------------------------------------------------------------------
def _schedule(i, n, kw):
    s = lp_oracle.get_lp_strength(i, n, kw.get("lp_strength_schedule_type", "none"),
                                  kw.get("schedule_interval_start_time", 0.0),
                                  kw.get("schedule_interval_end_time", 0.05),
                                  kw.get("schedule_linear_start_weight", 1.0),
                                  kw.get("schedule_linear_end_weight", 0.0), kw.get("schedule_linear_end_time", 0.5),
                                  kw.get("schedule_exp_decay_rate", 10.0))
    sigma, ksize, factor = lp_oracle.modulated_params(s, kw.get("lp_blur_sigma", 15.0),
                                                      kw.get("lp_blur_kernel_size", 0.02734375),
                                                      kw.get("lp_resize_factor", 0.25),
                                                      kw.get("schedule_blur_kernel_size", False))
    return s, sigma, ksize, factor


def wan_denoise_loop(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
                     num_inference_steps, guidance_scale=5.0, use_low_pass_guidance=True, transformer_dtype=torch.bfloat16,
                     patch_t=1, trace=None, prepare_lp=None, **kw):
    """wan:815-927.  latents fp32 [B,16,F,h,w]; condition fp32 [B,20,F,h,w].  ``transformer(x, timestep, ehs,
    ehs_image)`` -> prediction in transformer_dtype.  ``scheduler``: oracle.sched_oracle.UniPCOracle.
    ``prepare_lp(filter_type, sigma, ksize, factor)``: the pixel branch (wan:493-540), called on every ALG step (wan:869-880)."""
    scheduler.set_timesteps(num_inference_steps)
    do_cfg = guidance_scale > 1
    for i, t in enumerate(scheduler.timesteps):
        s = None
        if do_cfg and use_low_pass_guidance:
            s, sigma, ksize, factor = _schedule(i, num_inference_steps, kw)
            if prepare_lp is not None:
                lp = prepare_lp(kw.get("lp_filter_type", "none"), sigma, ksize, factor)
            else:
                lp = apply_low_pass_filter_torch(condition, kw.get("lp_filter_type", "none"), sigma, ksize, factor)
                rem = lp.size(1) % patch_t                        # wan:549-556 (dim 1 is the channel dim here)
                if rem != 0:
                    lp = torch.cat([lp[:, :min(patch_t - rem, lp.shape[1])], lp], dim=1)
            lp = lp.to(condition.dtype)
            if s == 0.0:
                x = torch.cat([torch.cat([latents] * 2), torch.cat([condition, condition], dim=0)], dim=1)
                ehs = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
            else:
                x = torch.cat([torch.cat([latents] * 3), torch.cat([condition, lp, lp], dim=0)], dim=1)
                ehs = torch.cat([negative_prompt_embeds, negative_prompt_embeds, prompt_embeds], dim=0)
        elif do_cfg:
            x = torch.cat([torch.cat([latents] * 2), torch.cat([condition, condition], dim=0)], dim=1)
            ehs = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
        else:
            raise UnboundLocalError("latent_model_input")
        x = x.to(transformer_dtype)
        ie = image_embeds.repeat(x.shape[0], 1, 1) if image_embeds.shape[0] != x.shape[0] else image_embeds
        pred = transformer(x, t.expand(x.shape[0]), ehs, ie)
        if pred.shape[0] == 3:
            u0, u, tx = pred.chunk(3)
            pred = u0 + guidance_scale * (tx - u)
        else:
            u, tx = pred.chunk(2)
            pred = u + guidance_scale * (tx - u)
        latents = scheduler.step(pred, t, latents)
        if trace is not None:
            trace.append((s, 3 if x.shape[0] == 3 * latents.shape[0] else 2, x.shape[0]))
    return latents


def hunyuan_denoise_loop(transformer, scheduler, latents, image_latents, pos, neg, num_inference_steps,
                         true_cfg_scale=1.0, guidance_scale=1.0, use_low_pass_guidance=False, lp_on_noisy_latent=False,
                         image_condition_type="token_replace", guidance_embeds=True, transformer_dtype=torch.bfloat16,
                         patch=2, sigmas=None, trace=None, **kw):
    """hy:1111-1270.  ``pos`` / ``neg``: (embeds, pooled, mask) triples (neg may be None).  ``transformer(x, timestep,
    ehs, mask, pooled, guidance)``.  ``scheduler``: oracle.sched_oracle.FlowMatchEulerOracle."""
    do_true_cfg = true_cfg_scale > 1 and neg is not None
    sig = np.linspace(1.0, 0.0, num_inference_steps + 1)[:-1] if sigmas is None else sigmas
    scheduler.set_timesteps(sigmas=sig)
    n_steps = len(scheduler.timesteps)
    guidance = None
    if guidance_embeds:
        guidance = torch.tensor([guidance_scale] * latents.shape[0], dtype=transformer_dtype) * 1000.0

    def low_passed(i):
        s, sigma, ksize, factor = _schedule(i, n_steps, kw)
        lp = apply_low_pass_filter_torch(image_latents, kw.get("lp_filter_type", "none"), sigma, ksize, factor)
        rem = lp.size(1) % patch                                # hy:780-787 (dim 1 is the channel dim here)
        if rem != 0:
            lp = torch.cat([lp[:, :min(patch - rem, lp.shape[1])], lp], dim=1)
        return s, lp.to(image_latents.dtype)

    for i, t in enumerate(scheduler.timesteps):
        s = None
        if do_true_cfg and use_low_pass_guidance:
            s, lp = low_passed(i)
            if s == 0.0 or lp_on_noisy_latent:
                conds = [image_latents, image_latents]
            else:
                conds = [image_latents, lp, lp]
        elif do_true_cfg:
            conds = [image_latents, image_latents]
        elif not use_low_pass_guidance:
            conds = [image_latents]
        else:
            s, lp = low_passed(i)
            conds = [lp]
        n = len(conds)
        x = torch.cat([torch.cat(conds, dim=0), torch.cat([latents] * n)[:, :, 1:]], dim=2).to(transformer_dtype)
        if n == 1:
            ehs, pooled, mask = pos
        else:
            ehs, pooled, mask = (torch.cat([a] * (n - 1) + [b], dim=0) for a, b in zip(neg, pos))
        pred = transformer(x, t.expand(x.shape[0]).to(transformer_dtype), ehs, mask, pooled, guidance)
        if pred.shape[0] == 3:
            u0, u, tx = pred.chunk(3)
            pred = u0 + true_cfg_scale * (tx - u)
        elif pred.shape[0] == 2:
            u, tx = pred.chunk(2)
            pred = u + true_cfg_scale * (tx - u)
        if image_condition_type == "latent_concat":
            latents = scheduler.step(pred, t, latents)
        else:
            latents = scheduler.step(pred[:, :, 1:], t, latents[:, :, 1:])
            latents = torch.cat([image_latents, latents], dim=2)
        if trace is not None:
            trace.append((s, n, x.shape[0]))
    return latents
