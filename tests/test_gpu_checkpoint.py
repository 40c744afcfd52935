"""`from_pretrained` of the pipelines and their components over a diffusers-format checkpoint DIRECTORY (what
`run.py:38-90` of the reference does): tiny synthetic checkpoints are written with safetensors + config.json in the
published layout, loaded back, and must run bit-identically to the same weights handed over in memory."""
import dataclasses
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from alg_amd import CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline, CogVideoXTransformer3DModel
from alg_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX, AutoencoderKLCogVideoXConfig
from alg_amd.image_encoder_clip import CLIPImageProcessor, CLIPVisionEncoderConfig, CLIPVisionModel
from alg_amd.pipeline_wan_image2video_lowpass import WanImageToVideoPipeline
from alg_amd.schedulers import UniPCMultistepScheduler
from alg_amd.text_encoder_clip import CLIPTextEncoderConfig, CLIPTextModel
from alg_amd.text_encoder_t5 import T5EncoderConfig, T5EncoderModel, UMT5EncoderModel
from alg_amd.transformer_cogvideox import CogVideoXTransformerConfig
from alg_amd.transformer_wan import WanTransformer3DModel, WanTransformerConfig
from oracle import clip_oracle, clip_text_oracle, dit_oracle, t5_oracle, vae_oracle, wan_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _save(root, sub, config, sd, shards=1, extra=None):
    d = os.path.join(root, sub)
    os.makedirs(d, exist_ok=True)
    cfg = dict(config, _class_name="Anything", _diffusers_version="0.0", **(extra or {}))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    for i in range(shards):
        save_file({k: sd[k].to(BF).contiguous() for k in keys[i * per:(i + 1) * per]},
                  os.path.join(d, "diffusion_pytorch_model-%05d-of-%05d.safetensors" % (i + 1, shards)))


class _Tok:
    """Stands in for the checkpoint's tokenizer (vocabulary files are checkpoint data): deterministic ids + mask."""

    def __call__(self, texts, padding=None, max_length=10, truncation=True, add_special_tokens=True, return_tensors="pt",
                 **_):
        texts = [texts] if isinstance(texts, str) else list(texts)
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        mask = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, t in enumerate(texts):
            n = min(max_length, len(t.split()) + 1)
            ids[i, :n] = torch.tensor([(hash(w) % 89) + 3 for w in t.split()][:n - 1] + [1])
            mask[i, :n] = 1
        return type("Enc", (), {"input_ids": ids, "attention_mask": mask})()


def _write_cogvideox_checkpoint(root, make_tokenizer_dir):
    """A tiny CogVideoX-I2V checkpoint directory in the published layout; returns the in-memory weights too."""
    make_tokenizer_dir(root)                                         # tokenizer/: a real sentencepiece T5 tokenizer
    small = dict(num_attention_heads=8, attention_head_dim=64, in_channels=32, out_channels=16, num_layers=1,
                 time_embed_dim=64, text_embed_dim=128, max_text_seq_length=10, sample_width=6, sample_height=4,
                 sample_frames=9, patch_size=2)
    w_tr = {k: v.to(BF) for k, v in dit_oracle.init_weights(dit_oracle.DiTConfig(**small), seed=4, std=0.05,
                                                             randomize_affine=True).items()}
    vkw = dict(layers_per_block=1)
    w_vae = vae_oracle.synthetic_state_dict(vae_oracle.VAEConfig(**vkw), seed=6, encoder=True)
    tkw = dict(vocab_size=96, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    w_t5 = t5_oracle.synthetic_state_dict(t5_oracle.T5Config(**tkw), seed=7)
    _save(root, "transformer", dataclasses.asdict(CogVideoXTransformerConfig(**small)), w_tr, shards=2)
    _save(root, "vae", dict(dataclasses.asdict(AutoencoderKLCogVideoXConfig(**vkw)), sample_height=480), w_vae)
    _save(root, "text_encoder", dict(tkw, feed_forward_proj="gated-gelu", architectures=["T5EncoderModel"]), w_t5, shards=3)
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "CogVideoXDDIMScheduler", "snr_shift_scale": 1.0, "timestep_spacing": "trailing",
                   "rescale_betas_zero_snr": True, "beta_schedule": "scaled_linear", "prediction_type": "v_prediction"}, f)
    return small, w_tr, vkw, w_vae, tkw, w_t5


def test_cogvideox_pipeline_from_a_checkpoint_directory(tmp_path, make_tokenizer_dir):
    root = str(tmp_path)
    small, w_tr, vkw, w_vae, tkw, w_t5 = _write_cogvideox_checkpoint(root, make_tokenizer_dir)

    pipe = CogVideoXImageToVideoPipeline.from_pretrained(root, torch_dtype=BF, device=DEV).to(DEV)
    assert isinstance(pipe.vae, AutoencoderKLCogVideoX) and isinstance(pipe.text_encoder, T5EncoderModel)
    assert type(pipe.tokenizer).__name__.startswith("T5Tokenizer")
    assert pipe.scheduler.config.snr_shift_scale == 1.0 and pipe.vae.config.layers_per_block == 1
    direct = CogVideoXImageToVideoPipeline(
        pipe.tokenizer, T5EncoderModel(T5EncoderConfig(**tkw), device=DEV).load_state_dict({k: v.to(BF) for k, v in w_t5.items()}),
        AutoencoderKLCogVideoX(AutoencoderKLCogVideoXConfig(**vkw), device=DEV).load_state_dict(w_vae),
        CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**small), w_tr, device=DEV),
        CogVideoXDDIMScheduler(snr_shift_scale=1.0)).to(DEV)
    g = torch.Generator().manual_seed(1)
    image = torch.rand(1, 3, 32, 48, generator=g) * 2 - 1            # the whole reference call: prompt + image in, frames out
    kw = dict(prompt="a small boat drifts", negative_prompt="blurry", height=32, width=48, num_frames=9,
              num_inference_steps=2, max_sequence_length=10, use_low_pass_guidance=True, lp_filter_type="down_up",
              lp_resize_factor=0.5, lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
              schedule_interval_end_time=0.6, lp_filter_in_latent=True, output_type="uint8")
    a = pipe(image=image, generator=torch.Generator().manual_seed(2), **kw).frames
    b = direct(image=image, generator=torch.Generator().manual_seed(2), **kw).frames
    assert a.shape == (1, 9, 32, 48, 3) and a.dtype == torch.uint8
    assert torch.equal(a, b)


def test_encoders_from_checkpoint_directories(tmp_path):
    root = str(tmp_path)
    ids = torch.randint(3, 90, (2, 12), generator=torch.Generator().manual_seed(0))
    mask = torch.ones(2, 12, dtype=torch.long)
    mask[1, 8:] = 0
    ukw = dict(vocab_size=96, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    w_u = t5_oracle.synthetic_state_dict(t5_oracle.T5Config(per_layer_bias=True, **ukw), seed=3)
    _save(root, "text_encoder", dict(ukw, model_type="umt5"), w_u, shards=2)
    got = UMT5EncoderModel.from_pretrained(root, device=DEV)(ids.to(DEV), mask.to(DEV)).last_hidden_state
    want = UMT5EncoderModel(T5EncoderConfig(**ukw), device=DEV).load_state_dict({k: v.to(BF) for k, v in w_u.items()})(
        ids.to(DEV), mask.to(DEV)).last_hidden_state
    assert torch.equal(got, want)

    vkw = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    w_v = clip_oracle.synthetic_state_dict(clip_oracle.CLIPVisionConfig(**vkw), seed=4)
    # the 4.x `vision_model.` prefix and a joint CLIP config with a nested vision_config
    _save(root, "image_encoder", {"model_type": "clip", "vision_config": vkw, "text_config": {"hidden_size": 8}},
          {"vision_model." + k: v for k, v in w_v.items()})
    px = torch.randn(1, 3, 28, 28, generator=torch.Generator().manual_seed(5)).to(DEV, BF)
    got = CLIPVisionModel.from_pretrained(root, device=DEV)(px, output_hidden_states=True).hidden_states[-2]
    want = CLIPVisionModel(CLIPVisionEncoderConfig(**vkw), device=DEV).load_state_dict(
        {k: v.to(BF) for k, v in w_v.items()})(px, output_hidden_states=True).hidden_states[-2]
    assert torch.equal(got, want)

    ckw = dict(vocab_size=96, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=12, eos_token_id=2)
    w_c = clip_text_oracle.synthetic_state_dict(clip_text_oracle.CLIPTextConfig(**ckw), seed=6)
    _save(root, "text_encoder_2", dict(ckw, hidden_act="quick_gelu", projection_dim=64), w_c)
    tid = ids.clone()
    tid[:, -1] = 2
    got = CLIPTextModel.from_pretrained(root, device=DEV)(tid.to(DEV)).pooler_output
    want = CLIPTextModel(CLIPTextEncoderConfig(**ckw), device=DEV).load_state_dict({k: v.to(BF) for k, v in w_c.items()})(
        tid.to(DEV)).pooler_output
    assert torch.equal(got, want)


def test_wan_pipeline_from_a_checkpoint_directory(tmp_path):
    root = str(tmp_path)
    kw = dict(num_attention_heads=4, ffn_dim=1024, num_layers=1, text_dim=128, image_dim=128, added_kv_proj_dim=512)
    w = wan_oracle.init_weights(wan_oracle.WanConfig(**kw), seed=3)
    _save(root, "transformer", dataclasses.asdict(WanTransformerConfig(**kw)), w, shards=2)
    ukw = dict(vocab_size=96, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2)
    _save(root, "text_encoder", ukw, t5_oracle.synthetic_state_dict(t5_oracle.T5Config(per_layer_bias=True, **ukw), seed=3))
    vkw = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=28, patch_size=14)
    _save(root, "image_encoder", vkw, clip_oracle.synthetic_state_dict(clip_oracle.CLIPVisionConfig(**vkw), seed=4))
    os.makedirs(os.path.join(root, "image_processor"))
    with open(os.path.join(root, "image_processor", "preprocessor_config.json"), "w") as f:
        json.dump({"size": {"shortest_edge": 28}}, f)
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "UniPCMultistepScheduler", "flow_shift": 3.0, "prediction_type": "flow_prediction",
                   "use_flow_sigmas": True, "solver_order": 2}, f)
    pipe = WanImageToVideoPipeline.from_pretrained(root, device=DEV, tokenizer=_Tok()).to(DEV)
    assert isinstance(pipe.transformer, WanTransformer3DModel) and isinstance(pipe.text_encoder, UMT5EncoderModel)
    assert isinstance(pipe.image_encoder, CLIPVisionModel) and isinstance(pipe.image_processor, CLIPImageProcessor)
    assert isinstance(pipe.scheduler, UniPCMultistepScheduler) and pipe.image_processor.size == 28 and pipe.vae is None
    g = torch.Generator().manual_seed(9)
    cond = torch.randn(1, 20, 3, 16, 24, generator=g) * 0.5
    out = pipe(image=torch.rand(3, 40, 56, generator=g), prompt="a kite over the dunes", negative_prompt="static",
               image_condition=cond.to(DEV), height=128, width=192, num_frames=9, num_inference_steps=2, guidance_scale=5.0,
               max_sequence_length=10, output_type="latent", generator=torch.Generator().manual_seed(3)).frames
    assert out.shape == (1, 16, 3, 16, 24) and bool(torch.isfinite(out.float()).all())
    # the same call with a PIL image (what run.py hands over): the CLIP processor resizes / crops it on the host
    import numpy as np
    from PIL import Image
    pil = Image.fromarray((np.random.default_rng(2).random((40, 56, 3)) * 255).astype("uint8"))
    out2 = pipe(image=pil, prompt=["a kite over the dunes"], negative_prompt=["static"], image_condition=cond.to(DEV),
                height=128, width=192, num_frames=9, num_inference_steps=2, guidance_scale=5.0, max_sequence_length=10,
                output_type="latent", generator=torch.Generator().manual_seed(3), use_low_pass_guidance=True,
                lp_filter_type="down_up", lp_resize_factor=0.5, lp_filter_in_latent=True,
                lp_strength_schedule_type="interval", schedule_interval_start_time=0.0,
                schedule_interval_end_time=0.6).frames
    assert out2.shape == out.shape and bool(torch.isfinite(out2.float()).all())


def test_hunyuan_pipeline_from_a_checkpoint_directory(tmp_path):
    from alg_amd.pipeline_hunyuan_video_image2video_lowpass import HunyuanVideoImageToVideoPipeline
    from alg_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from alg_amd.transformer_hunyuan_video import HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig
    from oracle import hy_oracle
    root = str(tmp_path)
    kw = dict(num_attention_heads=4, num_layers=1, num_single_layers=1, num_refiner_layers=1, text_embed_dim=64,
              pooled_projection_dim=128)
    _save(root, "transformer", dataclasses.asdict(HunyuanVideoTransformerConfig(**kw)),
          hy_oracle.init_weights(hy_oracle.HyConfig(**kw), seed=3), shards=2)
    ckw = dict(vocab_size=96, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
               max_position_embeddings=77, eos_token_id=1)
    _save(root, "text_encoder_2", ckw, clip_text_oracle.synthetic_state_dict(clip_text_oracle.CLIPTextConfig(**ckw), seed=6))
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "shift": 7.0, "num_train_timesteps": 1000}, f)
    pipe = HunyuanVideoImageToVideoPipeline.from_pretrained(root, device=DEV, tokenizer_2=_Tok()).to(DEV)
    assert isinstance(pipe.transformer, HunyuanVideoTransformer3DModel) and isinstance(pipe.text_encoder_2, CLIPTextModel)
    assert isinstance(pipe.scheduler, FlowMatchEulerDiscreteScheduler) and pipe.text_encoder is None and pipe.vae is None
    g = torch.Generator().manual_seed(11)
    mask = torch.ones(1, 6)
    # the Llava embeddings are inputs (encoder not built); the pooled CLIP embedding comes from the loaded text tower
    out = pipe(prompt_embeds=torch.randn(1, 6, 64, generator=g).to(BF).to(DEV), prompt_attention_mask=mask.to(DEV),
               clip_prompt="a paper boat", negative_prompt=None, image_latents=(torch.randn(1, 16, 1, 16, 16, generator=g) * 0.7).to(DEV),
               height=128, width=128, num_frames=5, num_inference_steps=2, guidance_scale=6.0, true_cfg_scale=1.0,
               output_type="latent", generator=torch.Generator().manual_seed(3)).frames
    assert out.shape == (1, 16, 2, 16, 16) and bool(torch.isfinite(out.float()).all())


def test_run_py_command_line_over_a_checkpoint_directory(tmp_path, make_tokenizer_dir):
    """The reference's own invocation (`run.py --config --image_path --prompt --output_path`, run:26-144) with the YAML's
    model.path pointing at a checkpoint directory on disk: image + prompt in, a video file out."""
    import argparse

    import numpy as np
    import yaml
    from PIL import Image

    import run
    from alg_amd import video_io
    root = os.path.join(str(tmp_path), "CogVideoX-tiny-I2V")
    os.makedirs(root)
    _write_cogvideox_checkpoint(root, make_tokenizer_dir)
    img = os.path.join(str(tmp_path), "in.png")
    Image.fromarray((np.random.default_rng(0).random((40, 60, 3)) * 255).astype("uint8")).save(img)
    cfg = os.path.join(str(tmp_path), "alg.yaml")
    with open(cfg, "w") as f:
        yaml.safe_dump({"model": {"path": root, "dtype": "bfloat16"},
                        "generation": {"height": 32, "width": 48, "num_frames": 9, "num_inference_steps": 2,
                                       "guidance_scale": 6.0, "max_sequence_length": 10},
                        "alg": {"use_low_pass_guidance": True, "lp_filter_type": "down_up", "lp_filter_in_latent": True,
                                "lp_resize_factor": 0.5, "lp_strength_schedule_type": "interval",
                                "schedule_interval_start_time": 0.0, "schedule_interval_end_time": 0.6,
                                "lp_blur_sigma": None},
                        "video": {"fps": 8}}, f)
    out = os.path.join(str(tmp_path), "out.mp4")
    run.main(run.make_parser().parse_args(["--config", cfg, "--image_path", img, "--prompt", "a small boat drifts on the lake",
                                           "--output_path", out, "--generator_device", "cpu"]))
    frames, info = video_io.read_mp4(out)                    # run:127-133: an h264 track in an mp4 container
    assert info["codec"] == "avc1" and info["fps"] == 8.0
    assert frames.shape == (9, 32, 48, 3) and frames.dtype == np.uint8 and frames.std() > 0


@pytest.mark.parametrize("case", ["pixel_gaussian", "two_videos_per_prompt", "no_cfg", "plain_cfg_callback", "prompt_list"])
def test_reference_style_calls_on_the_loaded_pipeline(tmp_path, make_tokenizer_dir, case):
    """The keyword combinations a user of the reference passes (cog:727-774), on a pipeline loaded from disk with a PIL
    image: each must run and give finite frames of the documented shape."""
    import numpy as np
    from PIL import Image
    root = str(tmp_path)
    _write_cogvideox_checkpoint(root, make_tokenizer_dir)
    pipe = CogVideoXImageToVideoPipeline.from_pretrained(root, torch_dtype=BF, device=DEV).to(DEV)
    img = Image.fromarray((np.random.default_rng(1).random((50, 70, 3)) * 255).astype("uint8"))
    kw = dict(image=img, prompt="a small boat drifts", height=32, width=48, num_frames=9, num_inference_steps=3,
              max_sequence_length=10, guidance_scale=6.0, generator=torch.Generator().manual_seed(5), output_type="pt",
              use_low_pass_guidance=True, lp_filter_type="down_up", lp_resize_factor=0.5, lp_filter_in_latent=True,
              lp_strength_schedule_type="interval", schedule_interval_start_time=0.0, schedule_interval_end_time=0.5)
    n = 1
    if case == "pixel_gaussian":        # cog:586-703 pixel branch: blur the image, VAE-encode it again every ALG step
        kw.update(lp_filter_in_latent=False, lp_filter_type="gaussian_blur", lp_blur_sigma=2.0, lp_blur_kernel_size=5,
                  lp_strength_schedule_type="linear", schedule_linear_start_weight=1.0, schedule_linear_end_weight=0.0,
                  schedule_linear_end_time=0.7, schedule_blur_kernel_size=False)
    elif case == "two_videos_per_prompt":
        kw.update(num_videos_per_prompt=2)          # cog:903 overwrites it with 1: one video comes back, as in the reference
    elif case == "no_cfg":
        with pytest.raises(NameError, match="two_pass"):   # cog:1011-1012, 1084: ALG on + guidance_scale <= 1 is unbound there too
            pipe(**dict(kw, guidance_scale=1.0))
        kw.update(guidance_scale=1.0, use_low_pass_guidance=False, generator=torch.Generator().manual_seed(5))
    elif case == "plain_cfg_callback":
        seen = []

        def cb(p, i, t, d):
            seen.append((i, int(t), tuple(d["latents"].shape)))
            return d
        kw.update(use_low_pass_guidance=False, negative_prompt="blurry", callback_on_step_end=cb,
                  callback_on_step_end_tensor_inputs=["latents"])
    elif case == "prompt_list":
        kw.update(prompt=["a small boat drifts", "the kite flies over the dunes"], negative_prompt=["blurry", "static"],
                  image=[img, img.transpose(Image.FLIP_LEFT_RIGHT)])
        n = 2
    out = pipe(**kw).frames
    assert out.shape == (n, 9, 3, 32, 48) and bool(torch.isfinite(out.float()).all())
    assert 0.0 <= float(out.min()) and float(out.max()) <= 1.0
    if case == "plain_cfg_callback":
        assert [i for i, _, _ in seen] == [0, 1, 2] and seen[0][2] == (1, 3, 16, 4, 6)
