#!/usr/bin/env python3
"""Generates tests/golden/llava_vectors.npz with the REAL third-party implementation the HunyuanVideo pipeline depends on --
`transformers.LlavaForConditionalGeneration` (hy:282-420), as installed in this container (transformers 5.15.0; the
reference pins 4.48.1) -- in fp32 on CPU, on the seeded weights and inputs of oracle/llava_oracle.py.  Only inputs and
outputs are stored.

    python tests/golden/make_llava_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import llava_oracle  # noqa: E402


def main():
    import transformers
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    cfg, sd, ids, mask, pos, px = llava_oracle.golden_inputs()
    v = cfg.vision
    vc = CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                          num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                          hidden_act=v.hidden_act, layer_norm_eps=v.layer_norm_eps)
    tc = LlamaConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                     vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps,
                     max_position_embeddings=256, pad_token_id=cfg.pad_token_id, attention_bias=False, mlp_bias=False)
    model = LlavaForConditionalGeneration(LlavaConfig(vision_config=vc, text_config=tc, image_token_index=cfg.image_token_index,
                                                      vision_feature_layer=cfg.vision_feature_layer,
                                                      vision_feature_select_strategy="default",
                                                      projector_hidden_act="gelu")).eval().float()
    keys = list(model.state_dict().keys())
    # transformers 5.x: model.language_model.* / model.vision_tower.* / model.multi_modal_projector.*;
    # 4.48: language_model.model.* / vision_tower.vision_model.* / multi_modal_projector.*
    def target(k):
        if k.startswith("language_model."):
            rest = k[len("language_model."):]
            for cand in ("model.language_model." + rest, "language_model.model." + rest):
                if cand in keys:
                    return cand
        if k.startswith("vision_tower."):
            rest = k[len("vision_tower."):]
            for cand in ("model.vision_tower." + rest, "model.vision_tower.vision_model." + rest,
                         "vision_tower.vision_model." + rest):
                if cand in keys:
                    return cand
        for cand in ("model." + k, k):
            if cand in keys:
                return cand
        raise KeyError(k)
    mapped = {target(k): t for k, t in sd.items()}
    missing, unexpected = model.load_state_dict(mapped, strict=False)
    assert not unexpected, unexpected
    assert all(("lm_head" in m) or ("position_ids" in m) or ("inv_freq" in m) for m in missing), missing
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask, position_ids=pos, pixel_values=px, output_hidden_states=True)
        text_only = model(input_ids=torch.where(ids == cfg.image_token_index, torch.full_like(ids, 7), ids), attention_mask=mask,
                          position_ids=pos, output_hidden_states=True)
    hs = out.hidden_states
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "llava_vectors.npz"),
                        transformers_version=np.array(transformers.__version__), input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        position_ids=pos.numpy(), pixel_values=px.numpy(), n_hidden_states=np.array(len(hs)),
                        embeds=hs[0].numpy().astype(np.float32), skip2=hs[-3].numpy().astype(np.float32),
                        last=hs[-1].numpy().astype(np.float32), text_only_skip2=text_only.hidden_states[-3].numpy().astype(np.float32))
    print("hidden_states", len(hs), tuple(hs[-3].shape), "std %.4f" % hs[-3].std())


if __name__ == "__main__":
    main()
