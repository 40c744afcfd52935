#!/usr/bin/env python3
"""Generates tests/golden/clip_vectors.npz with the REAL third-party implementation the Wan pipeline depends on --
`transformers.CLIPVisionModel` (wan:228-234), as installed in this container (transformers 5.15.0; the reference pins
4.48.1) -- in fp32 on CPU, on the seeded weights and input of oracle/clip_oracle.py.  Only the input and outputs are stored.

    python tests/golden/make_clip_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clip_oracle  # noqa: E402


def main():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg, sd, px = clip_oracle.golden_inputs()
    model = CLIPVisionModel(CLIPVisionConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                             num_hidden_layers=cfg.num_hidden_layers,
                                             num_attention_heads=cfg.num_attention_heads, image_size=cfg.image_size,
                                             patch_size=cfg.patch_size, hidden_act=cfg.hidden_act,
                                             layer_norm_eps=cfg.layer_norm_eps)).eval()
    keys = set(model.state_dict().keys())
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in keys) else ""
    missing, unexpected = model.load_state_dict({prefix + k: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    with torch.no_grad():
        out = model(pixel_values=px, output_hidden_states=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_vectors.npz"),
                        transformers_version=np.array(transformers.__version__), pixel_values=px.numpy(),
                        penultimate=out.hidden_states[-2].numpy().astype(np.float32),
                        last_hidden_state=out.last_hidden_state.numpy().astype(np.float32),
                        n_hidden_states=np.array(len(out.hidden_states)))
    print("hidden_states", len(out.hidden_states), tuple(out.hidden_states[-2].shape), "std %.4f" % out.hidden_states[-2].std())


if __name__ == "__main__":
    main()
