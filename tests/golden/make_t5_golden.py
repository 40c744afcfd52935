#!/usr/bin/env python3
"""Generates tests/golden/t5_vectors.npz by running the REAL third-party implementation the reference depends on --
`transformers.T5EncoderModel` (cog:228-268) and `transformers.UMT5EncoderModel` (wan:185-234), as installed in this
container (transformers 5.15.0; the reference pins 4.48.1) -- in fp32 on CPU, on the seeded weights and inputs of
oracle/t5_oracle.py (`GOLDEN_CASES`).  Only inputs and outputs are stored; the weights are regenerated from the seed.

    python tests/golden/make_t5_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import t5_oracle  # noqa: E402


def main():
    import transformers
    from transformers import T5Config, T5EncoderModel, UMT5Config, UMT5EncoderModel
    out = {"transformers_version": np.array(transformers.__version__)}
    for name in t5_oracle.GOLDEN_CASES:
        cfg, sd, ids, mask = t5_oracle.golden_inputs(name)
        kw = dict(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                  num_heads=cfg.num_heads, relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                  relative_attention_max_distance=cfg.relative_attention_max_distance, feed_forward_proj="gated-gelu",
                  layer_norm_epsilon=cfg.layer_norm_epsilon, dropout_rate=0.0)
        model = (UMT5EncoderModel(UMT5Config(**kw)) if cfg.per_layer_bias else T5EncoderModel(T5Config(**kw))).eval()
        full = dict(sd)
        full["encoder.embed_tokens.weight"] = sd["shared.weight"]
        missing, unexpected = model.load_state_dict(full, strict=False)
        assert not unexpected and all("embed_tokens" in m or m == "shared.weight" for m in missing), (missing, unexpected)
        with torch.no_grad():
            y = model(ids, attention_mask=mask).last_hidden_state if mask is not None else model(ids)[0]
        out[name + "_ids"] = ids.numpy()
        out[name + "_mask"] = (mask if mask is not None else torch.ones_like(ids)).numpy()
        out[name + "_out"] = y.numpy().astype(np.float32)
        print(name, tuple(y.shape), "std %.4f" % y.std().item())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "t5_vectors.npz"), **out)


if __name__ == "__main__":
    main()
