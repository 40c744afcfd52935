#!/usr/bin/env python3
"""ALG image-to-video sampling on MI355X -- same CLI and YAML schema as the reference's run.py
(/root/reference/run.py:26-144): --config --image_path --prompt --output_path --model_cache_dir.

YAML: model.{path,dtype[,flow_shift,flow_reverse]}, generation.*, alg.*, video.{fps[,resolution]}; the merged
generation + alg sections are passed verbatim as __call__ keyword arguments, null = "use the default" (run.py:96-106).
The model family is chosen by substring of model.path (run.py:45,64,70).

What differs from the reference: the hot path runs on alg_amd's HIP kernels; weights are read from local disk only
(no hub download here); the text encoder / VAE are once-per-video components outside the hot path -- without them
(`--synthetic`) the run uses seeded synthetic weights, embeddings and image latents and writes the final latents.

Data-parallel extension (the reference is single-process):
    python run.py --config C --jobs jobs.yaml --gpus 8
`jobs.yaml` is a list of {image_path, prompt, output_path[, seed]} (one video each; seed defaults to 42 + index).  One
process per GPU (launched here through torch.distributed.run when not already under it); the weights are read or generated
ONCE on rank 0 and reach the other ranks through one bucketed RCCL broadcast over xGMI; video v runs on rank v mod world;
there is no per-step collective, so every video is bit-identical to the same job run alone on one GPU.
"""
import argparse
import logging
import os
import socket
import subprocess
import sys

import torch
import yaml

from alg_amd import (AutoencoderKLCogVideoX, AutoencoderKLHunyuanVideo, AutoencoderKLWan, CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel,
                     CogVideoXTransformerConfig, FlowMatchEulerDiscreteScheduler, HunyuanVideoImageToVideoPipeline,
                     HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig, UniPCMultistepScheduler,
                     WanImageToVideoPipeline, WanTransformer3DModel, WanTransformerConfig, parallel)
from alg_amd.lp_utils import get_hunyuan_video_size  # noqa: F401  (kept importable here, as in the reference)

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s", stream=sys.stdout)
logger = logging.getLogger(__name__)


def _synthetic_transformer(model_path, config, device, fp8=False):
    """Seeded shape-faithful synthetic weights.  `model.synthetic_config` (extension key, tests) overrides config fields.
    In a multi-rank run rank 0 generates them and the other ranks receive them over RCCL (one broadcast)."""
    over = dict(config["model"].get("synthetic_config") or {})
    multi = torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    first = (not multi) or torch.distributed.get_rank() == 0
    if "CogVideoX" in model_path:
        from alg_amd.weights import synthetic_state_dict
        cfg = CogVideoXTransformerConfig(**over)
        sd = synthetic_state_dict(cfg, seed=1234, std=0.02, device=device) if first else None
        return CogVideoXTransformer3DModel(cfg, parallel.broadcast_loaded_state_dict(sd, device), device=device)
    if "Wan" in model_path:
        from alg_amd.transformer_wan import synthetic_state_dict
        cfg = WanTransformerConfig(**over)
        sd = synthetic_state_dict(cfg, seed=1234, device=device) if first else None
        return WanTransformer3DModel(cfg, parallel.broadcast_loaded_state_dict(sd, device), device=device, fp8=fp8)
    from alg_amd.transformer_hunyuan_video import synthetic_state_dict
    cfg = HunyuanVideoTransformerConfig(**over)
    sd = synthetic_state_dict(cfg, seed=1234, device=device) if first else None
    return HunyuanVideoTransformer3DModel(cfg, parallel.broadcast_loaded_state_dict(sd, device), device=device)


def build_pipeline(config, args, device):
    """run.py:44-87: the pipeline of the model family named by model.path, its scheduler rebuilt the way the reference does."""
    model_path = config["model"]["path"]
    model_dtype = getattr(torch, config["model"]["dtype"])
    if "CogVideoX" in model_path:
        if args.synthetic:
            transformer = _synthetic_transformer(model_path, config, device)
            full = not config["model"].get("synthetic_config")
            vae = AutoencoderKLCogVideoX.from_synthetic(device=device) if full else None   # decoder: cog:427-433 on HIP
            pipe = CogVideoXImageToVideoPipeline(transformer=transformer, scheduler=CogVideoXDDIMScheduler(), vae=vae)
        else:
            pipe = CogVideoXImageToVideoPipeline.from_pretrained(model_path, torch_dtype=model_dtype,
                                                                 cache_dir=args.model_cache_dir)
    elif "Wan" in model_path:
        # run.py:63: UniPC with flow_shift 3.0 for 480p, 5.0 otherwise (the reference compares height with the STRING '480',
        # which never matches an int from YAML, so it always lands on 5.0 -- reproduced)
        flow_shift = 3.0 if config["generation"]["height"] == "480" else 5.0
        if args.synthetic:
            transformer = _synthetic_transformer(model_path, config, device, fp8=args.fp8)
            full = not config["model"].get("synthetic_config")
            vae = AutoencoderKLWan.from_synthetic(device=device) if full else None      # decode after the loop: wan:959 on HIP
            pipe = WanImageToVideoPipeline(transformer=transformer, vae=vae,
                                           scheduler=UniPCMultistepScheduler(flow_shift=flow_shift))
        else:   # run.py:54-66: encoders from the checkpoint directory, UniPC rebuilt from its config with the run's flow_shift
            pipe = WanImageToVideoPipeline.from_pretrained(model_path, device=device, fp8=args.fp8)
            pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler.config, flow_shift=flow_shift)
    elif "HunyuanVideo" in model_path:
        # run.py:82-86: from_config(pipe.scheduler.config, flow_shift=model.flow_shift, invert_sigmas=model.flow_reverse);
        # `flow_shift` is not a parameter of FlowMatchEulerDiscreteScheduler, the checkpoint's own shift (7.0 for
        # HunyuanVideo-I2V) stays in force
        over = dict(flow_shift=config["model"].get("flow_shift"), invert_sigmas=bool(config["model"].get("flow_reverse", False)))
        if args.synthetic:
            transformer = _synthetic_transformer(model_path, config, device)
            full = not config["model"].get("synthetic_config")
            vae = AutoencoderKLHunyuanVideo.from_synthetic(device=device) if full else None   # decode: hy:1291-1292 on HIP
            pipe = HunyuanVideoImageToVideoPipeline(transformer=transformer, vae=vae,
                                                    scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0, **over))
        else:
            pipe = HunyuanVideoImageToVideoPipeline.from_pretrained(model_path, device=device)
            pipe.scheduler = FlowMatchEulerDiscreteScheduler.from_config(pipe.scheduler.config, **over)
    else:
        raise ValueError(f"unknown model family in model.path: {model_path}")
    return pipe.to(device)


def synthetic_inputs(config, model_path, model_dtype, seed):
    """Seeded stand-ins for the once-per-video encoders' outputs (no checkpoint / tokenizer / image needed)."""
    g = torch.Generator().manual_seed(seed)
    kw = {}
    over = config["model"].get("synthetic_config") or {}
    gen = config.get("generation", {})
    if "HunyuanVideo" in model_path:
        if gen.get("height") and gen.get("width"):
            h, w_ = gen["height"], gen["width"]
        else:
            nominal = type("Img", (), {"size": (1280, 720)})()           # stands in for the 16:9 input image
            h, w_ = get_hunyuan_video_size(config["video"]["resolution"], nominal)   # run.py:112-113
        kw["height"], kw["width"] = h, w_
        n_tok, n_valid = 256, 48
        td, pd = over.get("text_embed_dim", 4096), over.get("pooled_projection_dim", 768)
        mask = torch.cat([torch.ones(1, n_valid), torch.zeros(1, n_tok - n_valid)], dim=1)
        kw["prompt_embeds"] = torch.randn(1, n_tok, td, generator=g).to(model_dtype)
        kw["pooled_prompt_embeds"] = torch.randn(1, pd, generator=g).to(model_dtype)
        kw["prompt_attention_mask"] = mask
        kw["negative_prompt_embeds"] = torch.randn(1, n_tok, td, generator=g).to(model_dtype)
        kw["negative_pooled_prompt_embeds"] = torch.randn(1, pd, generator=g).to(model_dtype)
        kw["negative_prompt_attention_mask"] = mask.clone()
        kw["negative_prompt"] = None
        kw["image_latents"] = torch.randn(1, 16, 1, h // 8, w_ // 8, generator=g) * 0.7
    elif "Wan" in model_path:
        h, w_, nf = gen.get("height", 480), gen.get("width", 832), gen.get("num_frames", 81)
        f_lat = (nf - 1) // 4 + 1
        td, idim = over.get("text_dim", 4096), over.get("image_dim", 1280)
        kw["prompt_embeds"] = torch.randn(1, 512, td, generator=g).to(model_dtype)
        kw["negative_prompt_embeds"] = torch.randn(1, 512, td, generator=g).to(model_dtype)
        kw["image_embeds"] = torch.randn(1, 257, idim, generator=g).to(model_dtype)
        cond = torch.randn(1, 20, f_lat, h // 8, w_ // 8, generator=g) * 0.7
        cond[:, :4] = 0.0
        cond[:, :4, 0] = 1.0                      # first-frame mask channels (wan:444-456)
        kw["image_condition"] = cond
    else:
        td, tl = over.get("text_embed_dim", 4096), over.get("max_text_seq_length", 226)
        h, w_ = gen.get("height") or 8 * over.get("sample_height", 60), gen.get("width") or 8 * over.get("sample_width", 90)
        c = over.get("in_channels", 32) // 2
        kw["prompt_embeds"] = torch.randn(1, tl, td, generator=g).to(model_dtype)
        kw["negative_prompt_embeds"] = torch.randn(1, tl, td, generator=g).to(model_dtype)
        kw["image_latents"] = (torch.randn(1, 1, c, h // 8, w_ // 8, generator=g) * 0.7).to(model_dtype)
        kw["height"], kw["width"] = h, w_
    return kw


def run_job(pipe, config, args, job):
    """One video: run.py:92-134 for one (image, prompt, seed, output_path)."""
    model_path = config["model"]["path"]
    model_dtype = getattr(torch, config["model"]["dtype"])
    seed = int(job.get("seed", 42))
    # the reference seeds a generator on the run's device (run.py:94: philox on a GPU).  `--generator_device cpu`
    # (extension) draws the initial noise from the CPU stream instead, which is what the CPU oracle can reproduce.
    gdev = args.generator_device or ("cuda" if torch.cuda.is_available() else "cpu")
    generator = torch.Generator(device=gdev).manual_seed(seed)
    pipe_kwargs = {"generator": generator}
    params_from_config = {**config.get("generation", {}), **config.get("alg", {})}
    for key, value in params_from_config.items():
        if value is not None:
            pipe_kwargs[key] = value

    if args.synthetic:
        pipe_kwargs.update(synthetic_inputs(config, model_path, model_dtype, seed))
        if getattr(pipe, "vae", None) is None:
            pipe_kwargs["output_type"] = "latent"
        elif "CogVideoX" in model_path:
            pipe_kwargs["output_type"] = "pil"
    else:
        from PIL import Image
        input_image = Image.open(job["image_path"]).convert("RGB")
        pipe_kwargs["image"] = input_image
        pipe_kwargs["prompt"] = job["prompt"]
        if "HunyuanVideo" in model_path:   # run.py:112-113: the aspect-ratio bucket of the input image
            pipe_kwargs["height"], pipe_kwargs["width"] = get_hunyuan_video_size(config["video"]["resolution"], input_image)

    logger.info("Starting video generation...")
    log_subset = {k: v for k, v in pipe_kwargs.items() if not torch.is_tensor(v) and k not in ("image", "generator")}
    logger.info(f"Pipeline arguments: {log_subset}")
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t_call = time.perf_counter()
    video_output = pipe(**pipe_kwargs)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t_call = time.perf_counter() - t_call
    frames = video_output.frames
    out_path = job["output_path"]
    timing = {"config": args.config, "call_seconds": t_call, "num_inference_steps": pipe_kwargs.get("num_inference_steps"),
              "output_type": pipe_kwargs.get("output_type")}
    if pipe_kwargs.get("output_type") == "latent":
        torch.save(frames.cpu(), out_path)
        logger.info(f"Saved final latents {tuple(frames.shape)} to: {out_path}")
        _write_timing(args, timing)
        return out_path
    video_frames = frames[0]
    logger.info(f"Video generation complete. Received {len(video_frames)} frames.")
    import numpy as np
    from alg_amd import video_io
    arr = np.stack([np.asarray(f) for f in video_frames])  # [T, H, W, C] (run.py:121-125)
    if arr.dtype != np.uint8:   # output_type "np" (the Wan / HunyuanVideo default): floats in [0, 1] -> (x * 255).clamp.to(uint8)
        arr = np.clip(arr.astype(np.float32) * 255.0, 0, 255).astype(np.uint8)
    t_write = time.perf_counter()
    video_io.write_video(out_path, arr, fps=config["video"]["fps"])   # .mp4 -> ISO-BMFF container (run.py:127-133)
    t_write = time.perf_counter() - t_write
    logger.info(f"Saved {arr.shape} frames (fps {config['video']['fps']}) to: {out_path}")
    timing.update(frames=int(arr.shape[0]), height=int(arr.shape[1]), width=int(arr.shape[2]), write_seconds=t_write,
                  frames_per_s_call=arr.shape[0] / t_call, frames_per_s_call_plus_write=arr.shape[0] / (t_call + t_write))
    logger.info("Timing: __call__ (loop + VAE decode) %.2f s, writer %.2f s -> %.3f frames/s" % (
        t_call, t_write, timing["frames_per_s_call_plus_write"]))
    _write_timing(args, timing)
    return out_path


def _write_timing(args, timing):
    """extension (--timing_json): what one job took -- the figure a user of the reference would read off run.py:116-133"""
    if getattr(args, "timing_json", None):
        import json
        with open(args.timing_json, "a") as f:
            f.write(json.dumps(timing) + "\n")


def load_jobs(args):
    if not args.jobs:
        return [dict(image_path=args.image_path, prompt=args.prompt, output_path=args.output_path, seed=42)]
    with open(args.jobs) as f:
        jobs = yaml.safe_load(f)
    if not isinstance(jobs, list) or not jobs:
        raise ValueError("--jobs must be a YAML/JSON list of {image_path, prompt, output_path[, seed]}")
    out = []
    for v, j in enumerate(jobs):
        j = dict(j)
        j.setdefault("seed", 42 + v)
        j.setdefault("image_path", args.image_path)
        j.setdefault("prompt", args.prompt)
        if "output_path" not in j:
            root, ext = os.path.splitext(args.output_path)
            j["output_path"] = "%s_%03d%s" % (root, v, ext)
        out.append(j)
    return out


def self_launch(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(args):
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    rank, local_rank, world = parallel.init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    logger.info(f"Using device: {device}" + (f" (rank {rank} of {world})" if world > 1 else ""))

    jobs = load_jobs(args)
    pipe = build_pipeline(config, args, device)      # multi-rank: weights read / generated on rank 0, one RCCL broadcast
    logger.info("Pipeline loaded successfully.")
    mine = parallel.shard_videos(len(jobs), rank, world)   # video v -> rank v mod world; no data-path collective
    for v in mine:
        run_job(pipe, config, args, jobs[v])
    parallel.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def make_parser():
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
    parser.add_argument("--jobs", type=str, default=None,
                        help="extension: YAML/JSON list of {image_path, prompt, output_path[, seed]}, sharded over the GPUs")
    parser.add_argument("--gpus", type=int, default=1,
                        help="extension: data-parallel over this many GPUs of the node (one process each, weights broadcast once)")
    parser.add_argument("--timing_json", type=str, default=None,
                        help="extension: append one JSON line per job with the seconds of pipe.__call__ (loop + VAE decode) and of the writer")
    parser.add_argument("--generator_device", type=str, default=None, choices=["cpu", "cuda"],
                        help="extension: where the seed-42 generator lives (default: the run's device, as run.py:94)")
    return parser


if __name__ == "__main__":
    main(make_parser().parse_args())
