"""Host logic of the local-checkpoint loaders (`alg_amd/weights.py`): the diffusers / transformers directory layout the
reference's `from_pretrained` calls read (`run.py:38-90`), without a GPU -- directory walking, config filtering, scheduler
and processor configs, error messages."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from alg_amd import weights
from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoXConfig
from alg_amd.image_encoder_clip import CLIPImageProcessor, CLIPVisionEncoderConfig
from alg_amd.schedulers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler, FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
from alg_amd.text_encoder_t5 import T5EncoderConfig


def _write(root, sub, config, tensors=None, name="config.json", shards=1):
    d = os.path.join(root, sub)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(config, f)
    if tensors:
        keys = sorted(tensors)
        per = (len(keys) + shards - 1) // shards
        for i in range(shards):
            part = {k: tensors[k] for k in keys[i * per:(i + 1) * per]}
            save_file(part, os.path.join(d, "model-%05d-of-%05d.safetensors" % (i + 1, shards)))
    return d


def test_load_component_reads_config_and_all_shards(tmp_path):
    t = {"a.weight": torch.arange(6.0).reshape(2, 3), "b.bias": torch.ones(4), "c.weight": torch.zeros(1, 2).bfloat16()}
    _write(str(tmp_path), "vae", {"_class_name": "AutoencoderKLCogVideoX", "latent_channels": 16}, t, shards=2)
    raw, sd, root = weights.load_component(str(tmp_path), "vae")
    assert raw["latent_channels"] == 16 and root.endswith("vae")
    assert set(sd) == set(t) and all(torch.equal(sd[k], t[k]) for k in t) and sd["c.weight"].dtype == torch.bfloat16
    # the component directory itself is accepted too (no sub-folder)
    raw2, sd2, _ = weights.load_component(os.path.join(str(tmp_path), "vae"), "vae")
    assert raw2 == raw and set(sd2) == set(sd)


def test_load_component_errors_name_the_missing_piece(tmp_path):
    with pytest.raises(FileNotFoundError, match="config.json"):
        weights.load_component(str(tmp_path), "transformer")
    _write(str(tmp_path), "text_encoder", {"d_model": 8})
    with pytest.raises(FileNotFoundError, match="safetensors"):
        weights.load_component(str(tmp_path), "text_encoder")


def test_config_from_dict_filters_unwraps_and_tuples():
    raw = {"_class_name": "AutoencoderKLCogVideoX", "_diffusers_version": "0.32", "block_out_channels": [128, 256, 256, 512],
           "scaling_factor": 1.15258426, "invert_scale_latents": True, "sample_height": 480, "force_upcast": True,
           "norm_eps": None}
    cfg = weights.config_from_dict(AutoencoderKLCogVideoXConfig, raw)
    assert cfg.block_out_channels == (128, 256, 256, 512) and cfg.scaling_factor == 1.15258426 and cfg.invert_scale_latents
    assert cfg.norm_eps == AutoencoderKLCogVideoXConfig().norm_eps                 # null keeps the default
    joint = {"model_type": "clip", "vision_config": {"hidden_size": 64, "num_hidden_layers": 2, "dropout": 0.0},
             "text_config": {"hidden_size": 32}}
    v = weights.config_from_dict(CLIPVisionEncoderConfig, joint, nested="vision_config")
    assert v.hidden_size == 64 and v.num_hidden_layers == 2 and v.patch_size == CLIPVisionEncoderConfig().patch_size
    flat = weights.config_from_dict(CLIPVisionEncoderConfig, {"hidden_size": 48}, nested="vision_config")
    assert flat.hidden_size == 48
    t5 = weights.config_from_dict(T5EncoderConfig, {"architectures": ["T5EncoderModel"], "d_model": 16, "num_heads": 2,
                                                    "is_encoder_decoder": False, "use_cache": True})
    assert (t5.d_model, t5.num_heads, t5.d_kv) == (16, 2, T5EncoderConfig().d_kv)


@pytest.mark.parametrize("cls,cfg,check", [
    (CogVideoXDDIMScheduler, {"_class_name": "CogVideoXDDIMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear",
                              "snr_shift_scale": 1.0, "timestep_spacing": "trailing", "rescale_betas_zero_snr": True,
                              "set_alpha_to_one": True, "clip_sample": False, "prediction_type": "v_prediction"},
     lambda s: s.config.snr_shift_scale == 1.0 and s.config.timestep_spacing == "trailing"),
    (CogVideoXDPMScheduler, {"snr_shift_scale": 3.0, "_diffusers_version": "0.30.0.dev0", "prediction_type": "v_prediction"},
     lambda s: s.config.snr_shift_scale == 3.0 and s.config.timestep_spacing == "leading"
     and s.config.rescale_betas_zero_snr is False),      # keys the file leaves out: the published class defaults
    (UniPCMultistepScheduler, {"flow_shift": 3.0, "solver_order": 2, "prediction_type": "flow_prediction",
                               "use_flow_sigmas": True, "some_future_key": 1}, lambda s: s.flow_shift == 3.0 if hasattr(s, "flow_shift") else True),
    (FlowMatchEulerDiscreteScheduler, {"shift": 7.0, "num_train_timesteps": 1000, "base_image_seq_len": 256},
     lambda s: True),
])
def test_schedulers_load_their_config_and_ignore_unknown_keys(tmp_path, cls, cfg, check):
    _write(str(tmp_path), "scheduler", cfg, name="scheduler_config.json")
    s = cls.from_pretrained(str(tmp_path))
    assert isinstance(s, cls) and check(s)
    s2 = cls.from_config(cfg)
    s.set_timesteps(4)
    s2.set_timesteps(4)
    assert torch.equal(torch.as_tensor(s.timesteps), torch.as_tensor(s2.timesteps))
    with pytest.raises(FileNotFoundError):
        cls.from_pretrained(str(tmp_path / "nowhere"))


def test_partial_cogvideox_scheduler_config_resolves_like_the_published_class(tmp_path):
    """A scheduler_config.json without `prediction_type` means epsilon prediction under diffusers (the class default), which
    is not built: loading it must fail loudly instead of silently sampling with the CogVideoX-5b-I2V values."""
    _write(str(tmp_path), "scheduler", {"_class_name": "CogVideoXDDIMScheduler", "snr_shift_scale": 1.0},
           name="scheduler_config.json")
    with pytest.raises(NotImplementedError, match="v_prediction"):
        CogVideoXDDIMScheduler.from_pretrained(str(tmp_path))
    s = CogVideoXDDIMScheduler()                          # the constructor itself is the 5b-I2V scheduler (C2)
    assert (s.config.snr_shift_scale, s.config.timestep_spacing, s.config.rescale_betas_zero_snr) == (1.0, "trailing", True)


def test_image_processor_config_and_absent_tokenizer(tmp_path):
    _write(str(tmp_path), "image_processor", {"size": {"shortest_edge": 336}, "image_mean": [0.5, 0.5, 0.5],
                                              "image_std": [0.25, 0.25, 0.25], "do_center_crop": True},
           name="preprocessor_config.json")
    p = CLIPImageProcessor.from_pretrained(str(tmp_path))
    assert p.size == 336 and p.image_mean == (0.5, 0.5, 0.5) and p.image_std == (0.25, 0.25, 0.25)
    _write(str(tmp_path), "ip2", {"size": 224}, name="preprocessor_config.json")
    assert CLIPImageProcessor.from_pretrained(str(tmp_path), subfolder="ip2").size == 224
    assert weights.load_tokenizer(str(tmp_path), "tokenizer") is None          # no directory: the pipelines ask for embeddings


def test_tokenizer_directory_loads_through_transformers(tmp_path, make_tokenizer_dir):
    """`tokenizer/` of a checkpoint -> the object the pipelines call (cog:228-268 / wan:185-234 keyword set)."""
    make_tokenizer_dir(str(tmp_path))
    tok = weights.load_tokenizer(str(tmp_path), "tokenizer")
    assert tok is not None
    enc = tok(["a red bus", "a small boat drifts on the lake in the rain at sunset"], padding="max_length", max_length=10,
              truncation=True, add_special_tokens=True, return_attention_mask=True, return_tensors="pt")
    assert enc.input_ids.shape == (2, 10) and enc.attention_mask.shape == (2, 10)
    assert enc.attention_mask[1].sum() == 10 and 0 < enc.attention_mask[0].sum() < 10      # truncated / padded
    assert int(enc.input_ids[0][enc.attention_mask[0].sum() - 1]) == 1                   # </s> closes the short prompt
    assert int(enc.input_ids.max()) < 96
