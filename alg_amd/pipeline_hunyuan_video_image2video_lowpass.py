"""HunyuanVideo image-to-video sampler with Adaptive Low-pass Guidance -- the denoising loop, MI355X-native.

Drop-in for the loop of the reference's ``pipeline_hunyuan_video_image2video_lowpass.HunyuanVideoImageToVideoPipeline``
(SURVEY.md section 8 row a-5h): same class name, same ``__call__`` keyword arguments and defaults (hy:796-852),
same ``check_inputs`` errors (hy:494-548), same four loop branches (hy:1127-1229):

    true-CFG + ALG   2-pass [img, img] when strength == 0 or lp_on_noisy_latent, else 3-pass [img, lp, lp]
    true-CFG only    2-pass [img, img]
    neither          1 pass with the clean first frame
    ALG without CFG  1 pass with the LOW-PASSED first frame (no combine)

Per step everything elementwise runs through ``libalg_hip.so``: the HIP low-pass filter of the first-frame latent
(``prepare_lp``, once per distinct strength), ``alg_concat_cast`` for the first-frame token replace + CFG batch + cast
(hy:1146-1160, 1230), ``alg_cfg_combine`` with ``true_cfg_scale`` (hy:1254-1261), ``alg_lincomb`` for the flow-match
Euler step and ``alg_concat_cast`` again to re-prepend the clean first frame (hy:1265-1270).  The transformer is an
injected object with the diffusers HunyuanVideo signature (alg_amd's HIP DiT).  The prompt encoders are outside the hot
path and attached like diffusers registers them: the Llava-Llama-3 encoder (``text_encoder`` + ``tokenizer`` +
``image_processor``, hy:282-420: templated prompt next to the input image) and the CLIP-L text tower
(``text_encoder_2``); without them pass ``prompt_embeds`` / ``pooled_prompt_embeds`` / ``prompt_attention_mask``
(+ negatives).  With the HunyuanVideo VAE attached (``vae``: alg_amd's HIP `AutoencoderKLHunyuanVideo`) the conditioning image
is encoded by ``prepare_latents`` (hy:550-599) and the final latents are decoded to frames (hy:1290-1295); without it pass the
pre-encoded, scaled first frame as ``image_latents`` (extension kwarg) and use ``output_type="latent"``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import _lib, lp_utils
from .pipeline_cogvideox_image2video_lowpass import retrieve_timesteps
from .schedulers import FlowMatchEulerDiscreteScheduler

DEFAULT_PROMPT_TEMPLATE = {   # hy:84-100
    "template": (
        "<|start_header_id|>system<|end_header_id|>\n\n<image>\nDescribe the video by detailing the following aspects according to the reference image: "
        "1. The main content and theme of the video."
        "2. The color, shape, size, texture, quantity, text, and spatial relationships of the objects."
        "3. Actions, events, behaviors temporal relationships, physical movement changes of the objects."
        "4. background environment, light, style and atmosphere."
        "5. camera angles, movements, and transitions used in the video:<|eot_id|>\n\n"
        "<|start_header_id|>user<|end_header_id|>\n\n{}<|eot_id|>"
        "<|start_header_id|>assistant<|end_header_id|>\n\n"
    ),
    "crop_start": 103,
    "image_emb_start": 5,
    "image_emb_end": 581,
    "image_emb_len": 576,
    "double_return_token_id": 271,
}


def _expand_input_ids_with_image_tokens(text_input_ids, prompt_attention_mask, max_sequence_length, image_token_index,
                                        image_emb_len, image_emb_start, image_emb_end, pad_token_id):
    """Token ids / mask / positions of the Llava prompt once every `<image>` placeholder has grown into `image_emb_len` image
    tokens (behaviour of hy:107-146, restated): a token's new index is the running sum of the widths in front of it
    (placeholder: image_emb_len, anything else: 1); the image span is the template's [image_emb_start, image_emb_end); whatever
    is not the pad id is attended to; positions count attended tokens, masked ones sit at position 1."""
    ids = text_input_ids
    rows, _ = ids.shape
    is_image = ids == image_token_index
    width = torch.where(is_image, image_emb_len, 1)
    start = width.cumsum(dim=-1) - width                                   # first slot of every token after the expansion
    total = int(max_sequence_length + is_image.sum(dim=-1).max() * (image_emb_len - 1))
    # text tokens scatter to their slots; placeholders are parked in an extra trailing column that is cut off again
    out = ids.new_full((rows, total + 1), pad_token_id)
    out.scatter_(1, torch.where(is_image, total, start), ids)
    out = out[:, :total]
    out[:, image_emb_start:image_emb_end] = image_token_index
    attend = (out != pad_token_id).to(prompt_attention_mask.dtype)
    position_ids = (attend.cumsum(dim=-1) - 1).masked_fill(attend == 0, 1)
    return {"input_ids": out, "attention_mask": attend, "position_ids": position_ids}


def _drop_window(length, first, window_start, window):
    """Column indices [rows, length - first - window] of `first .. length` with the `window` columns from window_start[row] on
    left out -- the gather form of cat([x[first:ws], x[ws + window:]]) for a per-row ws."""
    j = first + torch.arange(length - first - window, device=window_start.device)[None, :]
    return j + window * (j >= window_start[:, None])


@dataclass
class HunyuanVideoPipelineOutput:
    frames: Any


def assemble_first_frame(latents, cond_groups, out_dtype):
    """[n_pass * B, C, Fc + F - 1, H, W]: for pass p, sample b the frames [cond_groups[p][b] | latents[b, :, 1:]]
    (hy:1146-1160: cat([img_cond, latent_model_input[:, :, 1:]], dim=2).to(dtype))."""
    B, C, F, H, W = latents.shape
    Fc = cond_groups[0].shape[2]
    R = H * W
    src0 = [g[b] for g in cond_groups for b in range(B)]
    src1 = [latents[b] for _ in cond_groups for b in range(B)]
    out = _lib.concat_cast(src0, src1, C, Fc, F - 1, R, Fc * R, F * R, 1, out_dtype)
    return out.view(len(src0), C, Fc + F - 1, H, W)


class HunyuanVideoImageToVideoPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds"]

    def __init__(self, text_encoder=None, tokenizer=None, transformer=None, vae=None, scheduler=None,
                 text_encoder_2=None, tokenizer_2=None, image_processor=None):
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.text_encoder_2, self.tokenizer_2, self.image_processor = text_encoder_2, tokenizer_2, image_processor
        self.transformer, self.vae, self.scheduler = transformer, vae, scheduler
        self.vae_scaling_factor = vae.config.scaling_factor if vae is not None else 0.476986
        self.vae_scale_factor_temporal = vae.temporal_compression_ratio if vae is not None else 4
        self.vae_scale_factor_spatial = vae.spatial_compression_ratio if vae is not None else 8
        self._device = torch.device("cpu")
        self._guidance_scale = None
        self._num_timesteps = None
        self._current_timestep = None
        self._attention_kwargs = None
        self._interrupt = False
        self._lp_cache = {}

    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.bfloat16, transformer=None, scheduler=None, vae=None,
                        text_encoder=None, tokenizer=None, text_encoder_2=None, tokenizer_2=None, image_processor=None,
                        device="cuda", **_):
        """Local-disk loader of a diffusers-format HunyuanVideo-I2V directory (`run.py:68-90`): `transformer/`,
        `text_encoder/` (Llava-Llama-3) + `tokenizer/` + `image_processor/`, `text_encoder_2/` (CLIP-L text tower) +
        `tokenizer_2/`, `vae/`, `scheduler/`."""
        import os

        from .schedulers import FlowMatchEulerDiscreteScheduler
        from .text_encoder_clip import CLIPTextModel
        from .transformer_hunyuan_video import HunyuanVideoTransformer3DModel
        from .weights import load_tokenizer

        has = lambda sub: os.path.isdir(os.path.join(model_path, sub))
        if transformer is None:
            transformer = HunyuanVideoTransformer3DModel.from_pretrained(model_path, device=device)
        if text_encoder is None and has("text_encoder"):
            from .text_encoder_llava import LlavaForConditionalGeneration
            text_encoder = LlavaForConditionalGeneration.from_pretrained(model_path, device=device)
        if tokenizer is None:
            tokenizer = load_tokenizer(model_path, "tokenizer")
        if image_processor is None and has("image_processor"):
            from .image_encoder_clip import CLIPImageProcessor
            image_processor = CLIPImageProcessor.from_pretrained(model_path)
        if text_encoder_2 is None and has("text_encoder_2"):
            text_encoder_2 = CLIPTextModel.from_pretrained(model_path, device=device)
        if tokenizer_2 is None:
            tokenizer_2 = load_tokenizer(model_path, "tokenizer_2")
        if vae is None and has("vae"):
            from .autoencoder_kl_hunyuan_video import AutoencoderKLHunyuanVideo
            vae = AutoencoderKLHunyuanVideo.from_pretrained(model_path, device=device)
        if scheduler is None:
            scheduler = (FlowMatchEulerDiscreteScheduler.from_pretrained(model_path) if has("scheduler")
                         else FlowMatchEulerDiscreteScheduler(shift=7.0))
        return cls(text_encoder, tokenizer, transformer, vae, scheduler, text_encoder_2, tokenizer_2, image_processor)

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
    num_timesteps = property(lambda self: self._num_timesteps)
    current_timestep = property(lambda self: self._current_timestep)
    interrupt = property(lambda self: self._interrupt)
    attention_kwargs = property(lambda self: self._attention_kwargs)

    def check_inputs(self, prompt, prompt_2, height, width, prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, prompt_template=None, true_cfg_scale=1.0,
                     guidance_scale=1.0):
        """hy:494-548."""
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
        elif prompt_2 is not None and prompt_embeds is not None:
            raise ValueError(
                f"Cannot forward both `prompt_2`: {prompt_2} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                " only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError(
                "Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        elif prompt_2 is not None and (not isinstance(prompt_2, str) and not isinstance(prompt_2, list)):
            raise ValueError(f"`prompt_2` has to be of type `str` or `list` but is {type(prompt_2)}")
        if prompt_template is not None:
            if not isinstance(prompt_template, dict):
                raise ValueError(f"`prompt_template` has to be of type `dict` but is {type(prompt_template)}")
            if "template" not in prompt_template:
                raise ValueError(
                    f"`prompt_template` has to contain a key `template` but only found {prompt_template.keys()}")

    def preprocess_image(self, image, height, width):
        """Minimal VideoProcessor.preprocess (hy:1046): tensors are taken as [B, 3, H, W] in [-1, 1]; PIL images are resized
        (Lanczos) and scaled to [-1, 1]."""
        if isinstance(image, torch.Tensor):
            return (image if image.ndim == 4 else image.unsqueeze(0)).to(torch.float32)
        imgs = image if isinstance(image, list) else [image]
        arr = [np.asarray(im.convert("RGB").resize((width, height), resample=1), dtype=np.float32) / 255.0 for im in imgs]
        return 2.0 * torch.from_numpy(np.stack(arr)).permute(0, 3, 1, 2).contiguous() - 1.0

    def postprocess_video(self, video, output_type="np"):
        """VideoProcessor.postprocess_video (hy:1295): [B, C, F, H, W] in [-1, 1] -> 'pt' | 'np' | 'pil'."""
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

    def encode_image(self, image, dtype, device):
        """hy:575-584: every image alone through the VAE as a one-frame video, posterior MODE ("argmax": the generator is
        not consumed), times the scaling factor -> [B, C, 1, h, w]."""
        if self.vae is None:
            raise _lib.AlgHipError("no HunyuanVideo VAE is attached to this pipeline: pass the pre-encoded, scaled first "
                                   "frame as `image_latents` [B, C, 1, h, w]")
        image = image.to(device=device, dtype=self.vae.dtype).unsqueeze(2)
        lat = [self.vae.encode(img.unsqueeze(0)).latent_dist.mode() for img in image]
        return torch.cat(lat, dim=0).to(dtype) * self.vae_scaling_factor

    def prepare_latents(self, image_latents, batch_size, num_channels_latents=32, height=720, width=1280,
                        num_frames=129, dtype=None, device=None, generator=None, latents=None,
                        image_condition_type="latent_concat", i2v_stable=False):
        """hy:550-599; ``image_latents`` is the VAE-encoded, scaled first frame [B, C, 1, h, w] (`encode_image`, or the
        extension kwarg of `__call__`)."""
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        f_lat = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        shape = (batch_size, num_channels_latents, f_lat, height // self.vae_scale_factor_spatial,
                 width // self.vae_scale_factor_spatial)
        image_latents = image_latents.to(device=device, dtype=dtype)
        if tuple(image_latents.shape) != shape[:2] + (1,) + shape[3:]:
            raise ValueError(f"`image_latents` must have shape {shape[:2] + (1,) + shape[3:]}, "
                             f"got {tuple(image_latents.shape)}")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype)
                                     for g in generator]).to(device)
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        if i2v_stable:  # hy:589-592
            image_latents = image_latents.repeat(1, 1, f_lat, 1, 1)
            t = torch.tensor([0.999]).to(device=device)
            latents = latents * t + image_latents * (1 - t)
        if image_condition_type == "token_replace":
            image_latents = image_latents[:, :, :1]
        return latents.contiguous(), image_latents.contiguous()

    def prepare_lp(self, lp_filter_type, lp_blur_sigma, lp_blur_kernel_size, lp_resize_factor, generator, num_frames,
                   use_low_pass_guidance, lp_filter_in_latent, orig_image_latents, orig_image_tensor, last_image=None):
        """hy:650-793, latent branch: per-plane HIP filter of ``[B, C, Fc, h, w]``; the reference's patch padding looks
        at dim 1 (channels) modulo the integer ``patch_size`` -- reproduced as is."""
        if not use_low_pass_guidance:
            return None
        if not lp_filter_in_latent:
            # hy:741-748 reads `self.vae.config.latents_mean` / `.latents_std` / `.z_dim` (the Wan VAE's config entries, the
            # branch was carried over from the Wan pipeline): AutoencoderKLHunyuanVideo has none of them, so the reference
            # stops here with an AttributeError as well
            raise AttributeError("lp_filter_in_latent=False is not runnable for HunyuanVideo: the branch (hy:741-748) "
                                 "normalises with vae.config.latents_mean / latents_std / z_dim, which "
                                 "AutoencoderKLHunyuanVideo's config does not have -- use lp_filter_in_latent=True")
        out = lp_utils.apply_low_pass_filter(orig_image_latents, lp_filter_type, lp_blur_sigma, lp_blur_kernel_size,
                                             lp_resize_factor)
        patch = getattr(getattr(self.transformer, "config", None), "patch_size", None)
        if patch is not None:
            rem = out.size(1) % patch
            if rem != 0:
                n_pre = min(patch - rem, out.shape[1])
                out = torch.cat([out[:, :n_pre], out], dim=1)
        return out.to(dtype=orig_image_latents.dtype)

    def _get_llama_prompt_embeds(self, image, prompt, prompt_template, num_videos_per_prompt=1, device=None, dtype=None,
                                 max_sequence_length=256, num_hidden_layers_to_skip=2, image_embed_interleave=2):
        """The Llava prompt embedding of hy:282-420 (same inputs, same outputs; the index arithmetic is restated in closed form):

          1. the prompt goes into the template and is tokenised to crop_start + max_sequence_length tokens;
          2. the one `<image>` placeholder grows into image_emb_len image tokens, the encoder runs on ids + pixel values, the
             hidden state `num_hidden_layers_to_skip` layers before the last is the embedding;
          3. output row = [every image_embed_interleave-th image token | user text], where "user text" is everything behind the
             template's first crop_start tokens except the four tokens of the assistant header, which end at the LAST
             double-return token of the row (a single prompt so long that the header was truncated away has only three of
             them: the window then ends at the sequence end).  Text indices live in expanded coordinates (+ image_emb_len - 1), mask
             indices in the tokenizer's."""
        if self.text_encoder is None or self.tokenizer is None or self.image_processor is None:
            raise _lib.AlgHipError("no Llava prompt encoder / tokenizer / image processor is attached: pass prompt_embeds and "
                                   "prompt_attention_mask")
        device = device or self._execution_device
        dtype = dtype or self.text_encoder.dtype
        tpl = prompt_template
        texts = [tpl["template"].format(p) for p in ([prompt] if isinstance(prompt, str) else prompt)]
        n_img, img_lo, img_hi = tpl.get("image_emb_len", 576), tpl.get("image_emb_start", 5), tpl.get("image_emb_end", 581)
        dr_id = tpl.get("double_return_token_id", 271)
        crop_start = tpl.get("crop_start", None)
        if crop_start is None:   # measure the template: its own tokens minus <|start_header_id|>, <|end_header_id|>, assistant, <|eot_id|>, {}
            crop_start = self.tokenizer(tpl["template"], padding="max_length", return_tensors="pt", return_length=False,
                                        return_overflowing_tokens=False, return_attention_mask=False)["input_ids"].shape[-1] - 5
        seq_len = max_sequence_length + crop_start
        tok = self.tokenizer(texts, max_length=seq_len, padding="max_length", truncation=True, return_tensors="pt",
                             return_length=False, return_overflowing_tokens=False, return_attention_mask=True)
        ids, mask = tok.input_ids.to(device=device), tok.attention_mask.to(device=device)
        pixels = self.image_processor(image, return_tensors="pt")
        pixels = (pixels["pixel_values"] if isinstance(pixels, dict) else pixels.pixel_values).to(device)
        enc_cfg = self.text_encoder.config
        expanded = _expand_input_ids_with_image_tokens(ids, mask, seq_len, enc_cfg.image_token_index, n_img, img_lo, img_hi,
                                                       enc_cfg.pad_token_id)
        hidden = self.text_encoder(**expanded, pixel_values=pixels,
                                   output_hidden_states=True).hidden_states[-(num_hidden_layers_to_skip + 1)].to(dtype=dtype)
        if crop_start <= 0:
            return hidden, mask
        L = ids.shape[-1]
        is_dr = ids == dr_id
        last_dr = L - 1 - is_dr.flip(-1).to(torch.int64).argmax(dim=-1)             # last double return of every row ...
        if int(is_dr.sum()) == 3:       # ... or the end: a single prompt so long that its assistant header was truncated away
            last_dr = torch.full_like(last_dr, L)                                    # (the reference's test, hy:361-370)
        shift = n_img - 1                                                            # text index -> expanded index
        keep_h = _drop_window(hidden.shape[1], crop_start + shift, last_dr - 4 + shift, 4)
        keep_m = _drop_window(L, crop_start, last_dr - 4, 4)
        text = hidden.gather(1, keep_h[:, :, None].expand(-1, -1, hidden.shape[-1]))
        text_mask = mask.gather(1, keep_m)
        image_part = hidden[:, img_lo:img_hi]
        if 0 < image_embed_interleave < 6:
            image_part = image_part[:, ::image_embed_interleave]
        image_mask = torch.ones(image_part.shape[:2], dtype=mask.dtype, device=hidden.device)
        return torch.cat([image_part, text], dim=1), torch.cat([image_mask, text_mask], dim=1)

    def encode_prompt(self, image, prompt, prompt_2=None, prompt_template=DEFAULT_PROMPT_TEMPLATE, num_videos_per_prompt=1,
                      prompt_embeds=None, pooled_prompt_embeds=None, prompt_attention_mask=None, device=None, dtype=None,
                      max_sequence_length=256, image_embed_interleave=2):
        """hy:453-492."""
        if prompt_embeds is None:
            prompt_embeds, prompt_attention_mask = self._get_llama_prompt_embeds(
                image, prompt, prompt_template, num_videos_per_prompt, device=device, dtype=dtype,
                max_sequence_length=max_sequence_length, image_embed_interleave=image_embed_interleave)
        if pooled_prompt_embeds is None:
            if prompt_2 is None:
                prompt_2 = prompt
            pooled_prompt_embeds = self._get_clip_prompt_embeds(prompt, num_videos_per_prompt, device=device, dtype=dtype,
                                                                max_sequence_length=77)
        return prompt_embeds, pooled_prompt_embeds, prompt_attention_mask

    def _get_clip_prompt_embeds(self, prompt, num_videos_per_prompt=1, device=None, dtype=None, max_sequence_length=77):
        """hy:421-452: tokenizer_2 to 77 tokens, CLIP text tower, `pooler_output`."""
        if self.text_encoder_2 is None or self.tokenizer_2 is None:
            raise _lib.AlgHipError("no CLIP text encoder / tokenizer is attached: pass pooled_prompt_embeds")
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        text_inputs = self.tokenizer_2(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                       return_tensors="pt")
        return self.text_encoder_2(text_inputs.input_ids.to(device), output_hidden_states=False).pooler_output

    @torch.no_grad()
    def __call__(
        self,
        image=None,
        prompt: Union[str, List[str]] = None,
        prompt_2: Union[str, List[str]] = None,
        negative_prompt: Union[str, List[str]] = "bad quality",
        negative_prompt_2: Union[str, List[str]] = None,
        height: int = 720,
        width: int = 1280,
        num_frames: int = 129,
        num_inference_steps: int = 50,
        sigmas: List[float] = None,
        true_cfg_scale: float = 1.0,
        guidance_scale: float = 1.0,
        num_videos_per_prompt: Optional[int] = 1,
        generator=None,
        latents: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        pooled_prompt_embeds: Optional[torch.Tensor] = None,
        prompt_attention_mask: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        negative_pooled_prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_attention_mask: Optional[torch.Tensor] = None,
        output_type: Optional[str] = "pil",
        return_dict: bool = True,
        attention_kwargs: Optional[Dict[str, Any]] = None,
        callback_on_step_end: Optional[Callable] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        prompt_template: Dict[str, Any] = DEFAULT_PROMPT_TEMPLATE,
        max_sequence_length: int = 256,
        image_embed_interleave: Optional[int] = None,
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
        lp_on_noisy_latent=False,
        enable_lp_img_embeds=False,
        i2v_stable=False,
        # ---- extensions (not in the reference signature) ----
        image_latents: Optional[torch.Tensor] = None,
        step_trace: Optional[list] = None,
        clip_prompt: Optional[Union[str, List[str]]] = None,
        negative_clip_prompt: Optional[Union[str, List[str]]] = None,
    ):
        self.check_inputs(prompt, prompt_2, height, width, prompt_embeds, callback_on_step_end_tensor_inputs,
                          prompt_template, true_cfg_scale, guidance_scale)
        tcfg = self.transformer.config
        image_condition_type = tcfg.image_condition_type
        has_neg_prompt = negative_prompt is not None or (
            negative_prompt_embeds is not None
            and (negative_pooled_prompt_embeds is not None or negative_clip_prompt is not None))
        do_true_cfg = true_cfg_scale > 1 and has_neg_prompt
        self._guidance_scale = guidance_scale
        self._attention_kwargs = attention_kwargs
        self._current_timestep = None
        self._interrupt = False
        self._lp_cache = {}
        device = self._execution_device
        if device.type != "cuda":
            raise _lib.AlgHipError("the ALG sampler's hot path is HIP-only: move the pipeline to a GPU "
                                   "(`pipe.to('cuda')`); there is no CPU fallback")
        if output_type not in ("latent", "pt", "np", "pil"):                # before the 50-step loop, not after it
            raise ValueError(f"{output_type} is not supported. Make sure to choose one of ['latent', 'np', 'pt', 'pil']")
        if output_type != "latent" and getattr(self, "vae", None) is None:
            raise _lib.AlgHipError("no HunyuanVideo VAE is attached to this pipeline: use output_type='latent'")
        if image_latents is None:
            if image is None:
                raise ValueError("`image` (or the pre-encoded `image_latents`) is required")
            # hy:1045-1046, 575-584
            image_latents = self.encode_image(self.preprocess_image(image, height, width), torch.float32, device)
        # hy:483-490 / hy:1005-1013: the pooled embedding comes from the CLIP text tower when it is attached
        # (`clip_prompt` / `negative_clip_prompt` are extension kwargs: the reference's check_inputs forbids a prompt next to
        # prompt_embeds, and the Llava tower that would consume the prompt is not built)
        clip_text = clip_prompt if clip_prompt is not None else prompt
        if pooled_prompt_embeds is None and clip_text is not None and self.text_encoder_2 is not None:
            pooled_prompt_embeds = self._get_clip_prompt_embeds(clip_text, num_videos_per_prompt, device=device)
        neg_text = negative_clip_prompt if negative_clip_prompt is not None else negative_prompt
        if (do_true_cfg and negative_pooled_prompt_embeds is None and neg_text is not None
                and self.text_encoder_2 is not None):
            negative_pooled_prompt_embeds = self._get_clip_prompt_embeds(neg_text, num_videos_per_prompt, device=device)
        if image_embed_interleave is None:   # hy:1021-1027
            image_embed_interleave = 2 if image_condition_type == "latent_concat" else 4 if image_condition_type == "token_replace" else 1
        if prompt_embeds is None and prompt is not None and self.text_encoder is not None:
            # hy:1073-1089: the Llava prompt encoder sees the input image next to the templated prompt
            prompt_embeds, pooled_prompt_embeds, prompt_attention_mask = self.encode_prompt(
                image=image, prompt=prompt, prompt_2=prompt_2, prompt_template=prompt_template,
                num_videos_per_prompt=num_videos_per_prompt, pooled_prompt_embeds=pooled_prompt_embeds, device=device,
                max_sequence_length=max_sequence_length, image_embed_interleave=image_embed_interleave)
        if (do_true_cfg and negative_prompt_embeds is None and negative_prompt is not None and self.text_encoder is not None
                and image is not None):
            from PIL import Image   # hy:1091-1108: the negative prompt is encoded next to a black image
            negative_prompt_embeds, negative_pooled_prompt_embeds, negative_prompt_attention_mask = self.encode_prompt(
                image=Image.new("RGB", (width, height), 0), prompt=negative_prompt, prompt_2=negative_prompt_2,
                prompt_template=prompt_template, num_videos_per_prompt=num_videos_per_prompt,
                pooled_prompt_embeds=negative_pooled_prompt_embeds, device=device, max_sequence_length=max_sequence_length,
                image_embed_interleave=image_embed_interleave)
        if prompt_embeds is None or pooled_prompt_embeds is None or prompt_attention_mask is None:
            raise _lib.AlgHipError("no Llava prompt encoder is attached (`text_encoder` / `tokenizer` / `image_processor`): pass "
                                   "prompt_embeds and prompt_attention_mask (and pooled_prompt_embeds unless a CLIP text tower "
                                   "is attached as `text_encoder_2` / `tokenizer_2`)")
        if do_true_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None
                            or negative_prompt_attention_mask is None):
            raise _lib.AlgHipError("true CFG needs negative_prompt_embeds, negative_pooled_prompt_embeds and "
                                   "negative_prompt_attention_mask (no prompt encoder is attached)")
        if not isinstance(self.scheduler, FlowMatchEulerDiscreteScheduler):
            raise TypeError("this sampler drives alg_amd.schedulers.FlowMatchEulerDiscreteScheduler (HIP step)")
        batch_size = prompt_embeds.shape[0]

        if image_condition_type == "latent_concat":
            num_channels_latents = (tcfg.in_channels - 1) // 2
        elif image_condition_type == "token_replace":
            num_channels_latents = tcfg.in_channels
        latents, image_latents = self.prepare_latents(
            image_latents, batch_size * num_videos_per_prompt, num_channels_latents, height, width, num_frames,
            torch.float32, device, generator, latents, image_condition_type, i2v_stable)
        if image_condition_type == "latent_concat":  # hy:1068-1071 (the mask is built and never used there)
            image_latents[:, :, 1:] = 0

        tdtype = self.transformer.dtype
        to_t = lambda x: None if x is None else x.to(device=device, dtype=tdtype)
        prompt_embeds, pooled_prompt_embeds = to_t(prompt_embeds), to_t(pooled_prompt_embeds)
        prompt_attention_mask = to_t(prompt_attention_mask)
        if do_true_cfg:
            negative_prompt_embeds = to_t(negative_prompt_embeds)
            negative_pooled_prompt_embeds = to_t(negative_pooled_prompt_embeds)
            negative_prompt_attention_mask = to_t(negative_prompt_attention_mask)

        sigmas = np.linspace(1.0, 0.0, num_inference_steps + 1)[:-1] if sigmas is None else sigmas
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, sigmas=sigmas)
        self._num_timesteps = len(timesteps)
        guidance = None
        if tcfg.guidance_embeds:  # hy:1114-1118
            guidance = torch.tensor([guidance_scale] * latents.shape[0], dtype=tdtype, device=device) * 1000.0

        def lp_for_step(i):
            strength = lp_utils.get_lp_strength(
                step_index=i, total_steps=num_inference_steps, lp_strength_schedule_type=lp_strength_schedule_type,
                schedule_interval_start_time=schedule_interval_start_time,
                schedule_interval_end_time=schedule_interval_end_time,
                schedule_linear_start_weight=schedule_linear_start_weight,
                schedule_linear_end_weight=schedule_linear_end_weight,
                schedule_linear_end_time=schedule_linear_end_time, schedule_exp_decay_rate=schedule_exp_decay_rate)
            sigma_i = lp_blur_sigma * strength
            ksize_i = lp_blur_kernel_size * strength if schedule_blur_kernel_size else lp_blur_kernel_size
            factor_i = 1.0 - (1.0 - lp_resize_factor) * strength
            if enable_lp_img_embeds:
                assert False, ("Low-pass filter on image embeds is not supported in HunyuanVideo pipeline. "
                               "Please set enable_lp_img_embeds = False")
            key = (lp_filter_type, sigma_i, ksize_i, type(ksize_i), factor_i)
            hit = self._lp_cache.get(key) if lp_filter_in_latent else None
            if hit is None:
                hit = self.prepare_lp(lp_filter_type, sigma_i, ksize_i, factor_i, generator, num_frames,
                                      use_low_pass_guidance, lp_filter_in_latent, image_latents, image)
                self._lp_cache[key] = hit
            return strength, hit

        neg3 = lambda a, b: torch.cat([a, a, b], dim=0)
        neg2 = lambda a, b: torch.cat([a, b], dim=0)
        for i, t in enumerate(timesteps):
            if self._interrupt:
                continue
            self._current_timestep = t
            strength = None
            if do_true_cfg and use_low_pass_guidance:
                strength, lp_lat = lp_for_step(i)
                if strength == 0.0 or lp_on_noisy_latent:
                    groups, cat = [image_latents, image_latents], neg2
                else:
                    groups, cat = [image_latents, lp_lat, lp_lat], neg3
            elif do_true_cfg:
                groups, cat = [image_latents, image_latents], neg2
            elif not use_low_pass_guidance:
                groups, cat = [image_latents], None
            else:  # ALG without true CFG: one pass on the low-passed first frame (hy:1196-1229)
                strength, lp_lat = lp_for_step(i)
                groups, cat = [lp_lat], None
            latent_model_input = assemble_first_frame(latents, groups, tdtype)
            if cat is None:
                ehs, pooled, mask = prompt_embeds, pooled_prompt_embeds, prompt_attention_mask
            else:
                ehs = cat(negative_prompt_embeds, prompt_embeds)
                pooled = cat(negative_pooled_prompt_embeds, pooled_prompt_embeds)
                mask = cat(negative_prompt_attention_mask, prompt_attention_mask)
            n = latent_model_input.shape[0]
            timestep = t.expand(n).to(device=device, dtype=tdtype)  # hy:1230 (the timestep itself is cast to bf16)
            noise_pred = self.transformer(
                hidden_states=latent_model_input, timestep=timestep, encoder_hidden_states=ehs,
                encoder_attention_mask=mask, pooled_projections=pooled, guidance=guidance,
                attention_kwargs=attention_kwargs, return_dict=False)[0]
            # hy:1254-1261 keys the combine on shape[0] (3 -> three chunks, 2 -> two chunks, anything else: none)
            if noise_pred.shape[0] in (2, 3):
                noise_pred = _lib.cfg_combine(noise_pred.contiguous(), noise_pred.shape[0], true_cfg_scale)
            if image_condition_type == "latent_concat":
                latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            elif image_condition_type == "token_replace":
                # hy:1267-1270: step frames 1.., then cat([image_latents, stepped], dim=2).  The update is elementwise,
                # so the whole tensor is stepped and frame 0 is overwritten by the re-prepend launch.
                stepped = self.scheduler.step(noise_pred.contiguous(), t, latents, return_dict=False)[0]
                B, C, F, H, W = stepped.shape
                out_dt = torch.promote_types(image_latents.dtype, stepped.dtype)
                latents = _lib.concat_cast([image_latents[b] for b in range(B)], [stepped[b] for b in range(B)],
                                           C, 1, F - 1, H * W, H * W, F * H * W, 1, out_dt).view(B, C, F, H, W)
            if step_trace is not None:
                step_trace.append((strength, len(groups), n))
            if callback_on_step_end is not None:
                pool = {"latents": latents, "prompt_embeds": prompt_embeds}
                outs = callback_on_step_end(self, i, t, {k: pool[k] for k in callback_on_step_end_tensor_inputs}) or {}
                latents = outs.pop("latents", latents).contiguous()
                prompt_embeds = outs.pop("prompt_embeds", prompt_embeds)
        self._current_timestep = None

        if output_type != "latent":   # hy:1290-1295
            video = self.vae.decode(latents.to(self.vae.dtype) / self.vae_scaling_factor, return_dict=False)[0]
            if image_condition_type == "latent_concat":
                video = video[:, :, 4:, :, :]
            video = self.postprocess_video(video, output_type=output_type)
        else:
            video = latents[:, :, 1:, :, :] if image_condition_type == "latent_concat" else latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return HunyuanVideoPipelineOutput(frames=video)
