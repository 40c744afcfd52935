#!/usr/bin/env python3
"""ALG image-to-video sampling on MI355X -- same CLI and YAML schema as the reference's run.py
(/root/reference/run.py:26-144): --config --image_path --prompt --output_path --model_cache_dir.

YAML: model.{path,dtype[,flow_shift,flow_reverse]}, generation.*, alg.*, video.{fps[,resolution]}; the merged
generation + alg sections are passed verbatim as __call__ keyword arguments, null = "use the default" (run.py:96-106).
The model family is chosen by substring of model.path (run.py:45,64,70).

What differs from the reference: the hot path runs on alg_amd's HIP kernels; weights are read from local disk only
(no hub download here); the text encoder / VAE are once-per-video components outside the hot path -- without them
(`--synthetic`) the run uses seeded synthetic weights, embeddings and image latents and writes the final latents.
"""
import argparse
import logging
import sys

import torch
import yaml

from alg_amd import (AutoencoderKLCogVideoX, CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel,
                     CogVideoXTransformerConfig, FlowMatchEulerDiscreteScheduler, HunyuanVideoImageToVideoPipeline,
                     HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig, UniPCMultistepScheduler,
                     WanImageToVideoPipeline, WanTransformer3DModel, WanTransformerConfig)
from alg_amd.lp_utils import get_hunyuan_video_size  # noqa: F401  (kept importable here, as in the reference)

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s", stream=sys.stdout)
logger = logging.getLogger(__name__)


def main(args):
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    model_path = config["model"]["path"]
    model_dtype = getattr(torch, config["model"]["dtype"])
    device = "cuda" if torch.cuda.is_available() else "cpu"
    logger.info(f"Using device: {device}")

    if "CogVideoX" in model_path:
        if args.synthetic:
            transformer = CogVideoXTransformer3DModel.from_synthetic(CogVideoXTransformerConfig(), device=device)
            vae = AutoencoderKLCogVideoX.from_synthetic(device=device)   # decoder only: cog:427-433 runs on HIP
            pipe = CogVideoXImageToVideoPipeline(transformer=transformer, scheduler=CogVideoXDDIMScheduler(), vae=vae)
        else:
            pipe = CogVideoXImageToVideoPipeline.from_pretrained(model_path, torch_dtype=model_dtype,
                                                                 cache_dir=args.model_cache_dir)
    elif "Wan" in model_path:
        # run.py:63: UniPC with flow_shift 3.0 for 480p, 5.0 otherwise (the reference compares height with the STRING '480',
        # which never matches an int from YAML, so it always lands on 5.0 -- reproduced)
        flow_shift = 3.0 if config["generation"]["height"] == "480" else 5.0
        if args.synthetic:
            transformer = WanTransformer3DModel.from_synthetic(WanTransformerConfig(), device=device, fp8=args.fp8)
            pipe = WanImageToVideoPipeline(transformer=transformer, scheduler=UniPCMultistepScheduler(flow_shift=flow_shift))
        else:   # run.py:54-66: encoders from the checkpoint directory, UniPC rebuilt with the run's flow_shift
            pipe = WanImageToVideoPipeline.from_pretrained(model_path, device=device, fp8=args.fp8,
                                                           scheduler=UniPCMultistepScheduler(flow_shift=flow_shift))
    elif "HunyuanVideo" in model_path:
        transformer = (HunyuanVideoTransformer3DModel.from_synthetic(HunyuanVideoTransformerConfig(), device=device)
                       if args.synthetic else None)
        # run.py:82-86: from_config(flow_shift=model.flow_shift, invert_sigmas=model.flow_reverse); `flow_shift` is not a
        # parameter of FlowMatchEulerDiscreteScheduler, the checkpoint's own shift (7.0 for HunyuanVideo-I2V) stays in force
        scheduler = FlowMatchEulerDiscreteScheduler(shift=7.0, flow_shift=config["model"].get("flow_shift"),
                                                    invert_sigmas=bool(config["model"].get("flow_reverse", False)))
        pipe = (HunyuanVideoImageToVideoPipeline(transformer=transformer, scheduler=scheduler) if args.synthetic
                else HunyuanVideoImageToVideoPipeline.from_pretrained(model_path, device=device, scheduler=scheduler))
    else:
        raise ValueError(f"unknown model family in model.path: {model_path}")
    pipe.to(device)
    logger.info("Pipeline loaded successfully.")

    generator = torch.Generator(device="cpu").manual_seed(42)  # CPU stream so runs are comparable with the oracle
    pipe_kwargs = {"generator": generator}
    params_from_config = {**config.get("generation", {}), **config.get("alg", {})}
    for key, value in params_from_config.items():
        if value is not None:
            pipe_kwargs[key] = value

    if args.synthetic:
        g = torch.Generator().manual_seed(42)
        if "HunyuanVideo" in model_path:
            nominal = type("Img", (), {"size": (1280, 720)})()           # stands in for the 16:9 input image
            h, w_ = get_hunyuan_video_size(config["video"]["resolution"], nominal)   # run.py:112-113
            pipe_kwargs["height"], pipe_kwargs["width"] = h, w_
            n_tok, n_valid = 256, 48
            mask = torch.cat([torch.ones(1, n_valid), torch.zeros(1, n_tok - n_valid)], dim=1)
            pipe_kwargs["prompt_embeds"] = torch.randn(1, n_tok, 4096, generator=g).to(model_dtype)
            pipe_kwargs["pooled_prompt_embeds"] = torch.randn(1, 768, generator=g).to(model_dtype)
            pipe_kwargs["prompt_attention_mask"] = mask
            pipe_kwargs["negative_prompt_embeds"] = torch.randn(1, n_tok, 4096, generator=g).to(model_dtype)
            pipe_kwargs["negative_pooled_prompt_embeds"] = torch.randn(1, 768, generator=g).to(model_dtype)
            pipe_kwargs["negative_prompt_attention_mask"] = mask.clone()
            pipe_kwargs["negative_prompt"] = None
            pipe_kwargs["image_latents"] = torch.randn(1, 16, 1, h // 8, w_ // 8, generator=g) * 0.7
        elif "Wan" in model_path:
            gen = config.get("generation", {})
            h, w_, nf = gen.get("height", 480), gen.get("width", 832), gen.get("num_frames", 81)
            f_lat = (nf - 1) // 4 + 1
            pipe_kwargs["prompt_embeds"] = torch.randn(1, 512, 4096, generator=g).to(model_dtype)
            pipe_kwargs["negative_prompt_embeds"] = torch.randn(1, 512, 4096, generator=g).to(model_dtype)
            pipe_kwargs["image_embeds"] = torch.randn(1, 257, 1280, generator=g).to(model_dtype)
            cond = torch.randn(1, 20, f_lat, h // 8, w_ // 8, generator=g) * 0.7
            cond[:, :4] = 0.0
            cond[:, :4, 0] = 1.0                      # first-frame mask channels (wan:444-456)
            pipe_kwargs["image_condition"] = cond
        else:
            pipe_kwargs["prompt_embeds"] = torch.randn(1, 226, 4096, generator=g).to(model_dtype)
            pipe_kwargs["negative_prompt_embeds"] = torch.randn(1, 226, 4096, generator=g).to(model_dtype)
            pipe_kwargs["image_latents"] = (torch.randn(1, 1, 16, 60, 90, generator=g) * 0.7).to(model_dtype)
        pipe_kwargs["output_type"] = "latent" if getattr(pipe, "vae", None) is None else "pil"
    else:
        from PIL import Image
        pipe_kwargs["image"] = Image.open(args.image_path).convert("RGB")
        pipe_kwargs["prompt"] = args.prompt

    logger.info("Starting video generation...")
    log_subset = {k: v for k, v in pipe_kwargs.items() if not torch.is_tensor(v) and k not in ("image", "generator")}
    logger.info(f"Pipeline arguments: {log_subset}")
    video_output = pipe(**pipe_kwargs)
    frames = video_output.frames
    if pipe_kwargs.get("output_type") == "latent":
        torch.save(frames.cpu(), args.output_path)
        logger.info(f"Saved final latents {tuple(frames.shape)} to: {args.output_path}")
        return
    video_frames = frames[0]
    logger.info(f"Video generation complete. Received {len(video_frames)} frames.")
    import numpy as np
    from alg_amd import video_io
    arr = np.stack([np.asarray(f) for f in video_frames])  # [T, H, W, C] uint8 (run.py:121-125)
    out_path = args.output_path
    if out_path.lower().endswith(".mp4"):
        # run.py:127-133 writes h264 through torchvision / PyAV; no such encoder exists here -> Motion-JPEG AVI next to it
        out_path = out_path[:-4] + ".avi"
        logger.info("no h264 encoder in this environment: writing Motion-JPEG AVI instead of mp4")
    video_io.write_video(out_path, arr, fps=config["video"]["fps"])
    logger.info(f"Saved {arr.shape} frames (fps {config['video']['fps']}) to: {out_path}")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Arguments")
    parser.add_argument("--config", type=str, default="./configs/cogvideox_alg.yaml")
    parser.add_argument("--image_path", type=str, default="./assets/a red double decker bus driving down a street.jpg")
    parser.add_argument("--prompt", type=str, default="a red double decker bus driving down a street")
    parser.add_argument("--output_path", type=str, default="output.mp4")
    parser.add_argument("--model_cache_dir", type=str, default=None)
    parser.add_argument("--fp8", action="store_true",
                        help="extension (BASELINE config 5, Wan): e4m3 block linears on the fp8 MFMA")
    parser.add_argument("--synthetic", action="store_true",
                        help="extension: seeded synthetic weights/inputs (no checkpoint, text encoder or VAE needed)")
    main(parser.parse_args())
