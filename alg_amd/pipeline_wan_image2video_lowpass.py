"""Wan 2.1 image-to-video sampler with Adaptive Low-pass Guidance -- the denoising loop, MI355X-native.

Drop-in for the loop of the reference's ``pipeline_wan_image2video_lowpass.WanImageToVideoPipeline``
(SURVEY.md section 8 row a-5w): same class name, same ``__call__`` keyword arguments and defaults (wan:587-634),
same ``check_inputs`` errors (wan:318-369), same output object with ``.frames``.

Per step (wan:843-927) this sampler launches, all through the C ABI of ``libalg_hip.so``:
    * ``prepare_lp`` (wan:472-560): the HIP low-pass filters over the 20-channel condition ``[mask4 | latent16]``,
      once per *distinct* schedule strength (the reference re-filters every step);
    * ``alg_concat_cast``: the 2-/3-pass CFG batch ``[latents | condition_p]`` in the transformer dtype in ONE
      launch (reference: cat([latents]*n), cat(dim=0) of conditions, cat(dim=1), .to(dtype));
    * the transformer -- an injected object with the diffusers Wan signature (the Wan DiT itself is SURVEY section 8
      row a-6w, "next");
    * ``alg_cfg_combine``: ``u0 + g (text - u)`` in the prediction dtype (wan:919-924);
    * ``UniPCMultistepScheduler.step`` (alg_amd.schedulers): ``alg_lincomb`` + ``alg_unipc_update`` launches.

Once-per-video components outside the hot path (UMT5 text encoder, CLIP image encoder, Wan VAE: alg_amd's HIP
implementations or any duck-typed object with the diffusers call protocol) are injected; without them pass
``prompt_embeds`` / ``negative_prompt_embeds`` / ``image_embeds`` (reference kwargs), the pre-encoded ``image_condition``
(extension kwarg: the ``[B, 20, F, h, w]`` tensor wan:372-468 builds) or just the normalised VAE latents of the condition
video as ``latent_condition`` ``[B, 16, F, h, w]`` (the mask channels are then built here, wan:439-456), and
``output_type="latent"``.  With a VAE attached the pipeline goes image in -> frames out (wan:426-430 encode, :959 decode) and
the pixel-space ALG branch (``lp_filter_in_latent=False``, wan:493-540: filter the RGB image, re-encode the 81-frame
condition video EVERY step with a freshly sampled posterior) is available.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from . import _lib, lp_utils
from .schedulers import UniPCMultistepScheduler


@dataclass
class WanPipelineOutput:
    frames: Any


def assemble_channel_concat(latents, cond_groups, out_dtype):
    """[n_pass * B, C_lat + C_cond, F, H, W] = for pass p, sample b: [latents[b] | cond_groups[p][b]] (wan:877-889)."""
    B, C, F, H, W = latents.shape
    Cc = cond_groups[0].shape[1]
    src0 = [latents[b] for _ in cond_groups for b in range(B)]
    src1 = [g[b] for g in cond_groups for b in range(B)]
    R = F * H * W
    out = _lib.concat_cast(src0, src1, 1, C, Cc, R, C * R, Cc * R, 0, out_dtype)
    return out.view(len(src0), C + Cc, F, H, W)


def build_wan_condition(latent_condition, num_frames, vae_scale_factor_temporal=4, has_last_image=False):
    """wan:439-456: the 20-channel condition ``[mask4 | latent16]`` from the (normalised) VAE latents of the padded
    condition video ``[B, 16, F_lat, h, w]``.  The mask marks the conditioning frames in PIXEL time (first frame, plus the
    last one for first-last-frame checkpoints), the first frame's flag is repeated 4 times, and groups of 4 pixel frames
    fold into the channel dimension.  Once per video; plain tensor plumbing."""
    B, _, f_lat, h, w = latent_condition.shape
    mask = torch.ones(B, 1, num_frames, h, w)
    if has_last_image:
        mask[:, :, 1:num_frames - 1] = 0
    else:
        mask[:, :, 1:] = 0
    first = torch.repeat_interleave(mask[:, :, 0:1], dim=2, repeats=vae_scale_factor_temporal)
    mask = torch.concat([first, mask[:, :, 1:]], dim=2)
    mask = mask.view(B, -1, vae_scale_factor_temporal, h, w).transpose(1, 2)
    return torch.concat([mask.to(latent_condition.device, latent_condition.dtype), latent_condition], dim=1)


def prompt_clean(text):
    """wan:97-111: ftfy.fix_text (when the package is present), double html.unescape, whitespace collapse."""
    import html
    import re
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class WanImageToVideoPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, image_encoder=None, image_processor=None, transformer=None,
                 vae=None, scheduler=None):
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.image_encoder, self.image_processor = image_encoder, image_processor
        self.transformer, self.vae, self.scheduler = transformer, vae, scheduler
        self.vae_scale_factor_temporal = 2 ** sum(vae.temperal_downsample) if vae is not None else 4
        self.vae_scale_factor_spatial = 2 ** len(vae.temperal_downsample) if vae is not None else 8
        self._device = torch.device("cpu")
        self._guidance_scale = None
        self._num_timesteps = None
        self._current_timestep = None
        self._attention_kwargs = None
        self._interrupt = False
        # measurement hook (bench.py: a 2-pass step of the schedule timed without the 3-pass steps in front of it): loop iterations
        # below this index are skipped exactly as interrupted ones are (wan:845-846: `continue`).  0 = the reference's loop.
        self._first_step = 0
        self._lp_cache = {}

    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.bfloat16, transformer=None, scheduler=None, vae=None,
                        text_encoder=None, tokenizer=None, image_encoder=None, image_processor=None, device="cuda",
                        fp8=False, **_):
        """Local-disk loader of a diffusers-format Wan2.1-I2V directory (`run.py:54-66`): `transformer/`, `text_encoder/`
        (UMT5), `tokenizer/`, `image_encoder/` (CLIP ViT-H), `image_processor/`, `scheduler/` (UniPC), `vae/` (AutoencoderKLWan)."""
        import os

        from .image_encoder_clip import CLIPImageProcessor, CLIPVisionModel
        from .schedulers import UniPCMultistepScheduler
        from .text_encoder_t5 import UMT5EncoderModel
        from .transformer_wan import WanTransformer3DModel
        from .weights import load_tokenizer

        has = lambda sub: os.path.isdir(os.path.join(model_path, sub))
        if transformer is None:
            transformer = WanTransformer3DModel.from_pretrained(model_path, device=device, fp8=fp8)
        if text_encoder is None and has("text_encoder"):
            text_encoder = UMT5EncoderModel.from_pretrained(model_path, device=device)
        if tokenizer is None:
            tokenizer = load_tokenizer(model_path, "tokenizer")
        if image_encoder is None and has("image_encoder"):
            image_encoder = CLIPVisionModel.from_pretrained(model_path, device=device)
        if image_processor is None:
            image_processor = (CLIPImageProcessor.from_pretrained(model_path) if has("image_processor")
                               else CLIPImageProcessor())
        if scheduler is None:
            scheduler = UniPCMultistepScheduler.from_pretrained(model_path) if has("scheduler") else UniPCMultistepScheduler()
        if vae is None and has("vae"):
            from .autoencoder_kl_wan import AutoencoderKLWan
            vae = AutoencoderKLWan.from_pretrained(model_path, device=device)
        return cls(tokenizer, text_encoder, image_encoder, image_processor, transformer, vae, scheduler)

    def to(self, device=None, *args, **kwargs):
        if device is not None:
            self._device = torch.device(device)
        return self

    @property
    def _execution_device(self):
        return self._device

    def maybe_free_model_hooks(self):
        pass

    guidance_scale = property(lambda self: self._guidance_scale)
    do_classifier_free_guidance = property(lambda self: self._guidance_scale > 1)
    num_timesteps = property(lambda self: self._num_timesteps)
    current_timestep = property(lambda self: self._current_timestep)
    interrupt = property(lambda self: self._interrupt)
    attention_kwargs = property(lambda self: self._attention_kwargs)

    def check_inputs(self, prompt, negative_prompt, image, height, width, prompt_embeds=None,
                     negative_prompt_embeds=None, image_embeds=None, callback_on_step_end_tensor_inputs=None):
        """wan:318-369."""
        if image is not None and image_embeds is not None:
            raise ValueError(
                f"Cannot forward both `image`: {image} and `image_embeds`: {image_embeds}. Please make sure to"
                " only forward one of the two.")
        if image is None and image_embeds is None:
            raise ValueError(
                "Provide either `image` or `prompt_embeds`. Cannot leave both `image` and `image_embeds` undefined.")
        if image is not None and not isinstance(image, torch.Tensor) and not hasattr(image, "convert"):
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` but is {type(image)}")
        if height % 16 != 0 or width % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
            raise ValueError(
                f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(
                f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                " only forward one of the two.")
        elif negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(
                f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`: "
                f"{negative_prompt_embeds}. Please make sure to only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError(
                "Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        elif negative_prompt is not None and (
                not isinstance(negative_prompt, str) and not isinstance(negative_prompt, list)):
            raise ValueError(f"`negative_prompt` has to be of type `str` or `list` but is {type(negative_prompt)}")

    def encode_image(self, image, device=None):
        """wan:228-234: CLIP image processor, vision tower, penultimate hidden state."""
        if self.image_encoder is None or self.image_processor is None:
            raise _lib.AlgHipError("no image encoder / processor is attached: pass `image_embeds` (CLIP penultimate hidden "
                                   "state, [B, 257, 1280])")
        device = device or self._execution_device
        image = self.image_processor(images=image, return_tensors="pt").to(device)
        image_embeds = self.image_encoder(**image, output_hidden_states=True)
        return image_embeds.hidden_states[-2]

    def _get_t5_prompt_embeds(self, prompt, num_videos_per_prompt=1, max_sequence_length=512, device=None, dtype=None):
        """wan:185-226: clean, tokenise to `max_sequence_length` with an attention mask, run the (U)MT5 encoder with that
        mask, keep each prompt's valid rows and zero the rest."""
        if self.text_encoder is None or self.tokenizer is None:
            raise _lib.AlgHipError("no text encoder / tokenizer is attached: pass prompt_embeds / negative_prompt_embeds "
                                   "(UMT5 embeddings [B, %d, 4096])" % max_sequence_length)
        dtype = dtype or self.text_encoder.dtype
        prompt = [prompt] if isinstance(prompt, str) else prompt
        prompt = [prompt_clean(u) for u in prompt]
        batch_size = len(prompt)
        text_inputs = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                     add_special_tokens=True, return_attention_mask=True, return_tensors="pt")
        text_input_ids, mask = text_inputs.input_ids, text_inputs.attention_mask
        seq_lens = mask.gt(0).sum(dim=1).long()
        prompt_embeds = self.text_encoder(text_input_ids.to(device), mask.to(device)).last_hidden_state
        prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
        prompt_embeds = [u[:v] for u, v in zip(prompt_embeds, seq_lens)]
        prompt_embeds = torch.stack(
            [torch.cat([u, u.new_zeros(max_sequence_length - u.size(0), u.size(1))]) for u in prompt_embeds], dim=0)
        _, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_videos_per_prompt, 1)
        return prompt_embeds.view(batch_size * num_videos_per_prompt, seq_len, -1)

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, num_videos_per_prompt=1,
                      prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length=226, device=None, dtype=None):
        """wan:237-317."""
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._get_t5_prompt_embeds(prompt, num_videos_per_prompt, max_sequence_length, device, dtype)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = batch_size * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif batch_size != len(negative_prompt):
                raise ValueError(
                    f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                    f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                    " the batch size of `prompt`.")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt,
                                                                max_sequence_length, device, dtype)
        return prompt_embeds.to(device), (None if negative_prompt_embeds is None else negative_prompt_embeds.to(device))

    def preprocess_image(self, image, height, width):
        """Minimal VideoProcessor.preprocess (wan:820): tensors are taken as [B, 3, H, W] in [-1, 1]; PIL images are resized
        (Lanczos) and scaled to [-1, 1]; fp32 like the reference."""
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image.unsqueeze(0)
            return t.to(torch.float32)
        import numpy as np
        imgs = image if isinstance(image, list) else [image]
        arr = [np.asarray(im.convert("RGB").resize((width, height), resample=1), dtype=np.float32) / 255.0 for im in imgs]
        t = torch.from_numpy(np.stack(arr)).permute(0, 3, 1, 2).contiguous()
        return 2.0 * t - 1.0

    def _latent_stats(self, device, dtype):
        cfg = self.vae.config
        mean = torch.tensor(cfg.latents_mean).view(1, cfg.z_dim, 1, 1, 1).to(device, dtype)
        inv_std = 1.0 / torch.tensor(cfg.latents_std).view(1, cfg.z_dim, 1, 1, 1).to(device, dtype)
        return mean, inv_std

    def encode_condition(self, image, batch_size, height, width, num_frames, dtype, device, last_image=None,
                         sample_generator=None):
        """wan:402-456 (and wan:505-540 with `sample_generator`): the condition video [image, zeros ...] through the VAE,
        its posterior mode (or, in the pixel-space ALG branch, a sample) normalised by the latent statistics, the frame
        mask folded into 4 channels in front."""
        if self.vae is None:
            raise _lib.AlgHipError("no Wan VAE is attached to this pipeline: pass the pre-encoded `image_condition` "
                                   "[B, 20, F, h, w] or `latent_condition` [B, 16, F, h, w]")
        image = image.unsqueeze(2)
        if last_image is None:
            video = torch.cat([image, image.new_zeros(image.shape[0], image.shape[1], num_frames - 1, height, width)], dim=2)
        else:
            video = torch.cat([image, image.new_zeros(image.shape[0], image.shape[1], num_frames - 2, height, width),
                               last_image.unsqueeze(2)], dim=2)
        video = video.to(device=device, dtype=self.vae.dtype)
        dist = self.vae.encode(video).latent_dist
        if sample_generator is None:
            lat = dist.mode().repeat(batch_size // image.shape[0], 1, 1, 1, 1) if image.shape[0] != batch_size else dist.mode()
        else:
            lat = dist.sample(generator=sample_generator)
        mean, inv_std = self._latent_stats(lat.device, dtype)
        lat = (lat.to(dtype) - mean) * inv_std
        return build_wan_condition(lat, num_frames, self.vae_scale_factor_temporal, has_last_image=last_image is not None)

    def postprocess_video(self, video, output_type="np"):
        """VideoProcessor.postprocess_video (wan:960): [B, C, F, H, W] in [-1, 1] -> 'pt' | 'np' | 'pil'."""
        v = (video * 0.5 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return v.permute(0, 2, 1, 3, 4)
        arr = v.permute(0, 2, 3, 4, 1).cpu().float().numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            from PIL import Image
            return [[Image.fromarray((f * 255).round().astype("uint8")) for f in vid] for vid in arr]
        raise ValueError(f"{output_type} is not supported. Make sure to choose one of ['np', 'pt', 'pil']")

    def prepare_latents(self, image_condition, batch_size, num_channels_latents=16, height=480, width=832,
                        num_frames=81, dtype=None, device=None, generator=None, latents=None):
        """wan:372-468 with the VAE-encoded condition supplied: noise latents + shape checks."""
        f_lat = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        shape = (batch_size, num_channels_latents, f_lat, height // self.vae_scale_factor_spatial,
                 width // self.vae_scale_factor_spatial)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype)
                                     for g in generator]).to(device)
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        want = (batch_size, self.vae_scale_factor_temporal + num_channels_latents) + shape[2:]
        if tuple(image_condition.shape) != want:
            raise ValueError(f"`image_condition` must have shape {want}, got {tuple(image_condition.shape)}")
        return latents.contiguous(), image_condition.to(device=device, dtype=dtype).contiguous()

    def prepare_lp(self, lp_filter_type, lp_blur_sigma, lp_blur_kernel_size, lp_resize_factor, generator, num_frames,
                   use_low_pass_guidance, lp_filter_in_latent, orig_image_latents, orig_image_tensor):
        """wan:472-560, latent branch: the filter runs per (H, W) plane over the whole ``[mask | latent]`` condition;
        the reference's temporal-patch padding looks at dim 1 (the 20 channels) and prepends leading channels when
        that is not a multiple of patch_size[0] -- reproduced as is."""
        if not use_low_pass_guidance:
            return None
        if not lp_filter_in_latent:
            # wan:493-540: filter the RGB image, re-encode the zero-padded condition video, SAMPLE the posterior
            if self.vae is None or orig_image_tensor is None or not torch.is_tensor(orig_image_tensor):
                raise _lib.AlgHipError("lp_filter_in_latent=False re-encodes the filtered image every step: it needs an "
                                       "attached Wan VAE and the `image`")
            image_lp = lp_utils.apply_low_pass_filter(orig_image_tensor, lp_filter_type, lp_blur_sigma, lp_blur_kernel_size,
                                                      lp_resize_factor)
            b, _, height, width = orig_image_tensor.shape
            out = self.encode_condition(image_lp, b, height, width, num_frames, image_lp.dtype, image_lp.device,
                                        sample_generator=generator)
            return out.to(dtype=orig_image_latents.dtype)
        out = lp_utils.apply_low_pass_filter(orig_image_latents, lp_filter_type, lp_blur_sigma, lp_blur_kernel_size,
                                             lp_resize_factor)
        patch = getattr(getattr(self.transformer, "config", None), "patch_size", None)
        if patch is not None:
            rem = out.size(1) % patch[0]
            if rem != 0:
                n_pre = min(patch[0] - rem, out.shape[1])
                out = torch.cat([out[:, :n_pre], out], dim=1)
        return out.to(dtype=orig_image_latents.dtype)

    @torch.no_grad()
    def __call__(
        self,
        image=None,
        prompt: Union[str, List[str]] = None,
        negative_prompt: Union[str, List[str]] = None,
        height: int = 480,
        width: int = 832,
        num_frames: int = 81,
        num_inference_steps: int = 50,
        guidance_scale: float = 5.0,
        num_videos_per_prompt: Optional[int] = 1,
        generator=None,
        latents: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        image_embeds: Optional[torch.Tensor] = None,
        last_image: Optional[torch.Tensor] = None,
        output_type: Optional[str] = "np",
        return_dict: bool = True,
        attention_kwargs: Optional[Dict[str, Any]] = None,
        callback_on_step_end: Optional[Callable] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        max_sequence_length: int = 512,
        use_low_pass_guidance: bool = False,
        lp_filter_type: str = "none",
        lp_filter_in_latent: bool = False,
        lp_blur_sigma: float = 15.0,
        lp_blur_kernel_size: float = 0.02734375,
        lp_resize_factor: float = 0.25,
        lp_strength_schedule_type: str = "none",
        schedule_blur_kernel_size: bool = False,
        schedule_interval_start_time: float = 0.0,
        schedule_interval_end_time: float = 0.05,
        schedule_linear_start_weight: float = 1.0,
        schedule_linear_end_weight: float = 0.0,
        schedule_linear_end_time: float = 0.5,
        schedule_exp_decay_rate: float = 10.0,
        # ---- extensions (not in the reference signature) ----
        image_condition: Optional[torch.Tensor] = None,
        latent_condition: Optional[torch.Tensor] = None,
        step_trace: Optional[list] = None,
        cfg_split=None,
    ):
        self.check_inputs(prompt, negative_prompt, image, height, width, prompt_embeds, negative_prompt_embeds,
                          image_embeds, callback_on_step_end_tensor_inputs)
        if num_frames % self.vae_scale_factor_temporal != 1:  # wan:764-769
            num_frames = num_frames // self.vae_scale_factor_temporal * self.vae_scale_factor_temporal + 1
        num_frames = max(num_frames, 1)
        self._guidance_scale = guidance_scale
        self._attention_kwargs = attention_kwargs
        self._current_timestep = None
        self._interrupt = False
        self._lp_cache = {}
        device = self._execution_device
        if device.type != "cuda":
            raise _lib.AlgHipError("the ALG sampler's hot path is HIP-only: move the pipeline to a GPU "
                                   "(`pipe.to('cuda')`); there is no CPU fallback")
        if image_condition is None and latent_condition is not None:
            image_condition = build_wan_condition(latent_condition.float(), num_frames, self.vae_scale_factor_temporal,
                                                  has_last_image=last_image is not None)
        image_tensor = None
        if image is not None and self.vae is not None and (image_condition is None or not lp_filter_in_latent):
            image_tensor = self.preprocess_image(image, height, width).to(device)                    # wan:820
        if image_condition is None:
            if image_tensor is None:
                raise _lib.AlgHipError("no Wan VAE is attached to this pipeline: pass the pre-encoded `image_condition` "
                                       "[B, 20, F, h, w] or `latent_condition` [B, 16, F, h, w]")
            n_vid = (len(prompt) if isinstance(prompt, list) else 1) if prompt is not None else prompt_embeds.shape[0]
            last_t = None if last_image is None else self.preprocess_image(last_image, height, width).to(device)
            image_condition = self.encode_condition(image_tensor, n_vid * num_videos_per_prompt, height, width, num_frames,
                                                    torch.float32, device, last_image=last_t)
        if not isinstance(self.scheduler, UniPCMultistepScheduler):
            raise TypeError("this sampler drives alg_amd.schedulers.UniPCMultistepScheduler (HIP step)")
        if output_type != "latent" and self.vae is None:   # before the 40-50 step loop, not after it
            raise _lib.AlgHipError("no Wan VAE is attached to this pipeline: use output_type='latent'")

        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = self.do_classifier_free_guidance
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, negative_prompt, do_cfg, num_videos_per_prompt, prompt_embeds, negative_prompt_embeds,
            max_sequence_length, device)
        tdtype = self.transformer.dtype
        prompt_embeds = prompt_embeds.to(tdtype)
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = negative_prompt_embeds.to(tdtype)
        if image_embeds is None:                                                      # wan:805-810
            if last_image is None:
                image_embeds = self.encode_image(image, device)
            else:
                image_embeds = self.encode_image([image, last_image], device)
                _, l, d = image_embeds.shape
                image_embeds = image_embeds.reshape(-1, 2 * l, d)
        image_embeds = image_embeds.to(device).repeat(batch_size, 1, 1).to(tdtype)  # wan:811-812

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        self._num_timesteps = len(timesteps)
        z_dim = image_condition.shape[1] - self.vae_scale_factor_temporal
        latents, condition = self.prepare_latents(image_condition, batch_size * num_videos_per_prompt, z_dim, height,
                                                  width, num_frames, torch.float32, device, generator, latents)

        for i, t in enumerate(timesteps):
            if self._interrupt or i < self._first_step:
                continue
            self._current_timestep = t
            strength = None
            if do_cfg and use_low_pass_guidance:
                strength = lp_utils.get_lp_strength(
                    step_index=i, total_steps=num_inference_steps,
                    lp_strength_schedule_type=lp_strength_schedule_type,
                    schedule_interval_start_time=schedule_interval_start_time,
                    schedule_interval_end_time=schedule_interval_end_time,
                    schedule_linear_start_weight=schedule_linear_start_weight,
                    schedule_linear_end_weight=schedule_linear_end_weight,
                    schedule_linear_end_time=schedule_linear_end_time,
                    schedule_exp_decay_rate=schedule_exp_decay_rate)
                sigma_i = lp_blur_sigma * strength
                ksize_i = lp_blur_kernel_size * strength if schedule_blur_kernel_size else lp_blur_kernel_size
                factor_i = 1.0 - (1.0 - lp_resize_factor) * strength
                key = (lp_filter_type, sigma_i, ksize_i, type(ksize_i), factor_i)
                lp_cond = self._lp_cache.get(key) if lp_filter_in_latent else None
                if lp_cond is None:  # the reference filters every step, also when the result goes unused (wan:866)
                    lp_cond = self.prepare_lp(lp_filter_type, sigma_i, ksize_i, factor_i, generator, num_frames,
                                              use_low_pass_guidance, lp_filter_in_latent, condition, image_tensor)
                    if lp_filter_in_latent:     # the pixel branch draws from the generator every step: never cached
                        self._lp_cache[key] = lp_cond
                if strength == 0.0:  # wan:879 equivalent to vanilla
                    groups, embeds = [condition, condition], [negative_prompt_embeds, prompt_embeds]
                else:
                    groups = [condition, lp_cond, lp_cond]
                    embeds = [negative_prompt_embeds, negative_prompt_embeds, prompt_embeds]
            elif do_cfg:
                groups, embeds = [condition, condition], [negative_prompt_embeds, prompt_embeds]
            else:
                # reference quirk (wan:843-898): without CFG no branch assigns `latent_model_input`
                raise UnboundLocalError("local variable 'latent_model_input' referenced before assignment "
                                        "(the Wan ALG loop needs guidance_scale > 1)")
            latent_model_input = assemble_channel_concat(latents, groups, tdtype)
            n = latent_model_input.shape[0]
            timestep = t.expand(n).to(device)
            ehs = torch.cat(embeds, dim=0)
            ehs_img = image_embeds.repeat(n, 1, 1) if image_embeds.shape[0] != n else image_embeds
            if cfg_split is not None:
                # alg_amd.parallel.CFGPairSplit: this rank's share of the CFG passes, one all-gather merges the predictions
                B_ = latents.shape[0]
                rows = [p_ * B_ + b for p_ in cfg_split.my_passes(len(groups)) for b in range(B_)]
                local = self.transformer(hidden_states=latent_model_input[rows].contiguous(), timestep=timestep[:len(rows)],
                                         encoder_hidden_states=ehs[rows].contiguous(),
                                         encoder_hidden_states_image=ehs_img[rows].contiguous(),
                                         attention_kwargs=attention_kwargs, return_dict=False)[0]
                noise_pred = cfg_split.merge(local.contiguous(), len(groups), B_)
            else:
                noise_pred = self.transformer(
                    hidden_states=latent_model_input, timestep=timestep, encoder_hidden_states=ehs,
                    encoder_hidden_states_image=ehs_img, attention_kwargs=attention_kwargs, return_dict=False)[0]
            # wan:919-924 keys the 3-chunk combine on shape[0] == 3, so the reference's 3-pass step only works for one
            # video per call (a [3B, ...] prediction would be chunked in two and fail in the scheduler)
            n_pass = 3 if noise_pred.shape[0] == 3 else 2
            if n_pass != len(groups):
                raise ValueError("the Wan ALG loop (3-pass CFG keyed on shape[0] == 3, wan:919) supports one video "
                                 f"per call; got a batch of {latents.shape[0]}")
            noise_pred = _lib.cfg_combine(noise_pred.contiguous(), n_pass, guidance_scale)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            if step_trace is not None:
                step_trace.append((strength, len(groups), n))
            if callback_on_step_end is not None:
                pool = {"latents": latents, "prompt_embeds": prompt_embeds,
                        "negative_prompt_embeds": negative_prompt_embeds}
                outs = callback_on_step_end(self, i, t, {k: pool[k] for k in callback_on_step_end_tensor_inputs}) or {}
                latents = outs.pop("latents", latents).contiguous()
                prompt_embeds = outs.pop("prompt_embeds", prompt_embeds)
                negative_prompt_embeds = outs.pop("negative_prompt_embeds", negative_prompt_embeds)
        self._current_timestep = None

        if output_type != "latent":   # wan:945-960
            lat = latents.to(self.vae.dtype)
            mean, inv_std = self._latent_stats(lat.device, lat.dtype)
            lat = lat / inv_std + mean
            video = self.vae.decode(lat.contiguous(), return_dict=False)[0]
            video = self.postprocess_video(video, output_type=output_type)
        else:
            video = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return WanPipelineOutput(frames=video)
