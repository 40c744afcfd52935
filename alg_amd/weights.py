"""Weight sources for the CogVideoX DiT: a local diffusers-format checkpoint, or seeded synthetic weights at
the same shapes (no network here, so benches and tests run on synthetic weights and say so)."""
from __future__ import annotations

import glob
import json
import math
import os

import torch


def parameter_shapes(cfg):
    """diffusers state-dict name -> shape for CogVideoXTransformer3DModel (1.0: Conv2d patch embed; 1.5 (`patch_size_t`):
    Linear patch embed over (c, t, py, px), p_t-fold proj_out, optional ofs embedding)."""
    D, H = cfg.inner_dim, cfg.attention_head_dim
    p = cfg.patch_size
    lat_f = (cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1
    n_patch = (cfg.sample_height // p) * (cfg.sample_width // p) * lat_f
    E = cfg.time_embed_dim
    F4 = cfg.ff_inner_mult * D
    p_t = cfg.patch_size_t
    shapes = {
        "patch_embed.proj.weight": (D, cfg.in_channels, p, p) if p_t is None else (D, cfg.in_channels * p * p * p_t),
        "patch_embed.proj.bias": (D,),
        "patch_embed.text_proj.weight": (D, cfg.text_embed_dim),
        "patch_embed.text_proj.bias": (D,),
        "time_embedding.linear_1.weight": (E, D),
        "time_embedding.linear_1.bias": (E,),
        "time_embedding.linear_2.weight": (E, E),
        "time_embedding.linear_2.bias": (E,),
        "norm_final.weight": (D,),
        "norm_final.bias": (D,),
        "norm_out.linear.weight": (2 * D, E),
        "norm_out.linear.bias": (2 * D,),
        "norm_out.norm.weight": (D,),
        "norm_out.norm.bias": (D,),
        "proj_out.weight": (p * p * (p_t or 1) * cfg.out_channels, D),
        "proj_out.bias": (p * p * (p_t or 1) * cfg.out_channels,),
    }
    if cfg.ofs_embed_dim is not None:
        O = cfg.ofs_embed_dim
        shapes.update({"ofs_embedding.linear_1.weight": (O, O), "ofs_embedding.linear_1.bias": (O,),
                       "ofs_embedding.linear_2.weight": (O, O), "ofs_embedding.linear_2.bias": (O,)})
    if cfg.use_learned_positional_embeddings:
        shapes["patch_embed.pos_embedding"] = (1, cfg.max_text_seq_length + n_patch, D)
    for i in range(cfg.num_layers):
        b = "transformer_blocks.%d." % i
        for nm in ("norm1", "norm2"):
            shapes[b + nm + ".linear.weight"] = (6 * D, E)
            shapes[b + nm + ".linear.bias"] = (6 * D,)
            shapes[b + nm + ".norm.weight"] = (D,)
            shapes[b + nm + ".norm.bias"] = (D,)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            shapes[b + "attn1." + nm + ".weight"] = (D, D)
            shapes[b + "attn1." + nm + ".bias"] = (D,)
        for nm in ("norm_q", "norm_k"):
            shapes[b + "attn1." + nm + ".weight"] = (H,)
            shapes[b + "attn1." + nm + ".bias"] = (H,)
        shapes[b + "ff.net.0.proj.weight"] = (F4, D)
        shapes[b + "ff.net.0.proj.bias"] = (F4,)
        shapes[b + "ff.net.2.weight"] = (D, F4)
        shapes[b + "ff.net.2.bias"] = (D,)
    return shapes


def count_parameters(cfg):
    return sum(math.prod(s) for s in parameter_shapes(cfg).values())


def synthetic_state_dict(cfg, seed=1234, std=0.02, device="cuda", randomize_affine=False):
    """N(0, std^2) matrices / positional table, zero biases, unit norm gains, bf16, generated tensor by tensor
    on ``device`` from one seeded generator (deterministic for a given device type)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    sd = {}
    for name, shape in parameter_shapes(cfg).items():
        is_matrix = name.endswith("pos_embedding") or (name.endswith(".weight") and len(shape) >= 2)
        if is_matrix:
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32).mul_(std)
        elif name.endswith(".weight"):
            t = torch.ones(shape, device=dev)
            if randomize_affine:
                t = t + 0.1 * torch.randn(shape, generator=g, device=dev)
        else:
            t = torch.zeros(shape, device=dev)
            if randomize_affine:
                t = 0.05 * torch.randn(shape, generator=g, device=dev)
        sd[name] = t.to(torch.bfloat16)
    return sd


def read_shards(root, device=None):
    """Every `*.safetensors` shard of one component directory -> one state dict.  In a multi-rank run (torch.distributed
    initialised, world > 1) only rank 0 touches the disk: the tensors reach the other ranks through ONE bucketed RCCL
    broadcast over xGMI (`parallel.broadcast_loaded_state_dict`) -- eight ranks do not read the same 11-33 GB eight times.
    `ALG_BROADCAST_WEIGHTS=0` restores per-rank disk reads."""
    import torch.distributed as dist
    from safetensors.torch import load_file

    shards = sorted(glob.glob(os.path.join(root, "*.safetensors")))
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and \
        os.environ.get("ALG_BROADCAST_WEIGHTS", "1") != "0"
    if not multi:
        if not shards:
            raise FileNotFoundError("no *.safetensors under %s" % root)
        sd = {}
        for shard in shards:
            sd.update(load_file(shard))
        return sd
    from . import parallel
    sd = None
    if dist.get_rank() == 0:
        try:
            sd = {}
            for shard in shards:
                sd.update(load_file(shard))
            if not shards:
                raise FileNotFoundError("no *.safetensors under %s" % root)
        except Exception as e:      # a missing / corrupt shard: tell the waiting ranks instead of leaving them in a broadcast
            sd = e
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and
                                                                      dist.get_backend() == "nccl") else torch.device("cpu")
    return parallel.broadcast_loaded_state_dict(sd, device)


def load_diffusers_transformer(path, subfolder="transformer"):
    """Read config.json + safetensors shards of a diffusers CogVideoXTransformer3DModel from local disk."""
    from .transformer_cogvideox import CogVideoXTransformerConfig

    root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
    cfg_path = os.path.join(root, "config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(
            "%s not found: weights must be on local disk (this build has no network access; use "
            "CogVideoXTransformer3DModel.from_synthetic for shape-faithful synthetic weights)" % cfg_path)
    with open(cfg_path) as f:
        raw = json.load(f)
    fields = CogVideoXTransformerConfig.__dataclass_fields__
    cfg = CogVideoXTransformerConfig(**{k: v for k, v in raw.items() if k in fields})
    sd = read_shards(root)
    missing = [k for k in parameter_shapes(cfg) if k not in sd]
    if missing:
        raise KeyError("checkpoint is missing %d tensors, e.g. %s" % (len(missing), missing[:3]))
    return cfg, sd


# ---- generic local-checkpoint loading (diffusers / transformers directory layout) ----------------------------------------
def load_component(path, subfolder=None, config_name="config.json"):
    """(raw config dict, state dict, directory) of one component of a diffusers-format checkpoint on LOCAL disk:
    `<path>/<subfolder>/config.json` + every `*.safetensors` shard next to it (the layout `from_pretrained` of the
    reference's pipelines reads: cog / wan / hy `run.py:38-90`).  No hub access, no pickle (`*.bin`) loading."""
    root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
    cfg_path = os.path.join(root, config_name)
    if not os.path.exists(cfg_path):
        raise FileNotFoundError("%s not found: checkpoints must be on local disk (no network access here)" % cfg_path)
    with open(cfg_path) as f:
        raw = json.load(f)
    shards = sorted(glob.glob(os.path.join(root, "*.safetensors")))
    if not shards:
        raise FileNotFoundError("no *.safetensors in %s (pickled *.bin checkpoints are not read)" % root)
    sd = read_shards(root)
    return raw, sd, root


def config_from_dict(config_cls, raw, nested=None):
    """Dataclass config from a checkpoint's config.json: unknown keys are dropped, a nested sub-config (`vision_config` /
    `text_config` of a joint CLIP config) is unwrapped, JSON lists become tuples where the dataclass default is a tuple."""
    if nested and isinstance(raw.get(nested), dict):
        raw = raw[nested]
    fields = config_cls.__dataclass_fields__
    kw = {}
    for k, v in raw.items():
        if k in fields and v is not None:
            kw[k] = tuple(v) if isinstance(v, list) and isinstance(fields[k].default, tuple) else v
    return config_cls(**kw)


def component_from_pretrained(cls, config_cls, path, subfolder, device="cuda", nested=None, **ctor):
    raw, sd, _ = load_component(path, subfolder)
    obj = cls(config_from_dict(config_cls, raw, nested), device=device, **ctor)
    obj.load_state_dict(sd)
    return obj


def scheduler_from_pretrained(cls, path, subfolder="scheduler"):
    """`scheduler/scheduler_config.json` -> cls(**the keys its __init__ knows)."""
    import inspect

    root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
    cfg_path = os.path.join(root, "scheduler_config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError("%s not found" % cfg_path)
    with open(cfg_path) as f:
        raw = json.load(f)
    known = set(inspect.signature(cls.__init__).parameters) - {"self"}
    # a key the file leaves out takes the PUBLISHED class default (diffusers), which for the CogVideoX schedulers differs
    # from this build's constructor defaults (= the values CogVideoX-5b-I2V ships): a partial config must resolve as it
    # would under diffusers, and then fail loudly if it lands on a variant that is not built
    kw = dict(getattr(cls, "_published_defaults", {}))
    kw.update({k: v for k, v in raw.items() if k in known})
    return cls(**kw)


def load_tokenizer(path, subfolder="tokenizer"):
    """The checkpoint's tokenizer through `transformers` (vocabulary files are checkpoint data; the tokenisation algorithm
    itself is outside the hot path and not rebuilt).  None when the directory or the package is absent: the pipelines then
    ask for `prompt_embeds`."""
    d = os.path.join(path, subfolder)
    if not os.path.isdir(d):
        return None
    try:
        from transformers import AutoTokenizer
    except ImportError:
        return None
    return AutoTokenizer.from_pretrained(d, local_files_only=True)
