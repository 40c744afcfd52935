"""CPU: the CLIP vision oracle (wan:228-234) against golden vectors produced by the real
transformers CLIPVisionModel (tests/golden/make_clip_golden.py), plus host-side pieces of the product class."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_vectors.npz")


def test_oracle_matches_transformers_outputs():
    vec = np.load(GOLD)
    cfg, sd, px = clip_oracle.golden_inputs()
    assert np.array_equal(vec["pixel_values"], px.numpy())
    states = clip_oracle.encode(cfg, sd, px)
    assert len(states) == int(vec["n_hidden_states"]) == cfg.num_hidden_layers + 1
    assert np.abs(states[-2].numpy() - vec["penultimate"]).max() <= 2e-5
    assert np.abs(states[-1].numpy() - vec["last_hidden_state"]).max() <= 2e-5


def test_product_class_tables_and_processor():
    from alg_amd import _lib
    from alg_amd.image_encoder_clip import CLIPImageProcessor, CLIPVisionEncoderConfig, CLIPVisionModel
    m = CLIPVisionModel(device="cpu")
    assert (m.tokens, m.head_dim, m.kpad) == (257, 80, 640)
    assert m.param_shapes() == clip_oracle.param_shapes(clip_oracle.CLIPVisionConfig())
    n = sum(torch.Size(s).numel() for s in m.param_shapes().values())
    assert 0.62e9 < n < 0.64e9                       # ViT-H/14 vision tower
    cfg, sd, px = clip_oracle.golden_inputs()
    small = CLIPVisionModel(CLIPVisionEncoderConfig(**clip_oracle.GOLDEN["cfg"]), device="cpu")
    small.load_state_dict({"vision_model." + k: v for k, v in sd.items()})            # 4.x prefix accepted
    assert small.w["encoder.layers.0.qkv"].shape == (960, 320) and small.w["patch"].shape == (320, 640)
    assert torch.equal(small.w["cls"].float(), (sd["embeddings.class_embedding"].bfloat16()
                                                + sd["embeddings.position_embedding.weight"][0].bfloat16()).float())
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        small(pixel_values=px)
    from PIL import Image
    img = Image.fromarray((np.random.RandomState(0).rand(300, 500, 3) * 255).astype("uint8"))
    feat = CLIPImageProcessor()(images=img, return_tensors="pt")
    assert feat["pixel_values"].shape == (1, 3, 224, 224) and set(feat.keys()) == {"pixel_values"}
    assert feat.to("cpu")["pixel_values"].dtype == torch.float32
    assert abs(float(feat["pixel_values"].mean())) < 0.5


def test_clip_text_oracle_matches_transformers_outputs():
    from oracle import clip_text_oracle as cto
    vec = np.load(os.path.join(os.path.dirname(GOLD), "clip_text_vectors.npz"))
    cfg, sd, ids = cto.golden_inputs()
    assert np.array_equal(vec["input_ids"], ids.numpy())
    x, pooled = cto.encode(cfg, sd, ids)
    assert np.abs(x.numpy() - vec["last_hidden_state"]).max() <= 2e-5
    assert np.abs(pooled.numpy() - vec["pooler_output"]).max() <= 2e-5


def test_clip_text_product_tables_and_pooling_rule():
    from alg_amd import _lib
    from alg_amd.text_encoder_clip import CLIPTextEncoderConfig, CLIPTextModel
    from oracle import clip_text_oracle as cto
    m = CLIPTextModel(device="cpu")
    assert m.param_shapes() == cto.param_shapes(cto.CLIPTextConfig())
    n = sum(torch.Size(s).numel() for s in m.param_shapes().values())
    assert 0.12e9 < n < 0.13e9                      # CLIP-L/14 text tower
    ids = torch.tensor([[5, 9, 49407, 0, 0], [7, 49407, 3, 3, 3]])
    assert m.eos_positions(ids).tolist() == [2, 1]              # legacy rule: argmax of the ids
    m2 = CLIPTextModel(CLIPTextEncoderConfig(**cto.GOLDEN["cfg"]), device="cpu")
    assert m2.eos_positions(torch.tensor([[97, 5, 98, 0], [97, 98, 98, 0]])).tolist() == [2, 1]   # first eos token
    cfg, sd, _ = cto.golden_inputs()
    m2.load_state_dict({"text_model." + k: v for k, v in sd.items()})
    assert m2.w["encoder.layers.1.qkv"].shape == (384, 128)
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        m2(torch.zeros(1, 8, dtype=torch.long))
