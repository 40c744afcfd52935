#!/usr/bin/env python3
"""Generates tests/golden/clip_text_vectors.npz with the REAL third-party implementation HunyuanVideo's pipeline depends on
-- `transformers.CLIPTextModel` (hy:421-452) as installed in this container (transformers 5.15.0; the reference pins
4.48.1) -- in fp32 on CPU, on the seeded weights and ids of oracle/clip_text_oracle.py.  Only inputs and outputs are stored.

    python tests/golden/make_clip_text_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clip_text_oracle as co  # noqa: E402


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg, sd, ids = co.golden_inputs()
    model = CLIPTextModel(CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                         intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                                         num_attention_heads=cfg.num_attention_heads,
                                         max_position_embeddings=cfg.max_position_embeddings, hidden_act="quick_gelu",
                                         layer_norm_eps=cfg.layer_norm_eps, eos_token_id=cfg.eos_token_id, bos_token_id=97,
                                         pad_token_id=0)).eval()
    keys = set(model.state_dict().keys())
    prefix = "text_model." if any(k.startswith("text_model.") for k in keys) else ""
    missing, unexpected = model.load_state_dict({prefix + k: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    with torch.no_grad():
        out = model(ids, output_hidden_states=False)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_text_vectors.npz"),
                        transformers_version=np.array(transformers.__version__), input_ids=ids.numpy(),
                        last_hidden_state=out.last_hidden_state.numpy().astype(np.float32),
                        pooler_output=out.pooler_output.numpy().astype(np.float32))
    print("pooler", tuple(out.pooler_output.shape), "std %.4f" % out.pooler_output.std())


if __name__ == "__main__":
    main()
