"""CPU checks of the Llava prompt-encoder oracle against transformers' own outputs, and of the host-side prompt plumbing of
the HunyuanVideo pipeline (hy:100-146 image-token expansion, hy:282-420 template cropping) with stand-in encoders."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from alg_amd.pipeline_hunyuan_video_image2video_lowpass import (HunyuanVideoImageToVideoPipeline,
                                                               _expand_input_ids_with_image_tokens)
from oracle import llava_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llava_vectors.npz")


def test_llava_oracle_matches_transformers_golden_vectors():
    """tests/golden/llava_vectors.npz = transformers.LlavaForConditionalGeneration (fp32, CPU) on the seeded tiny model."""
    cfg, sd, ids, mask, pos, px = O.golden_inputs()
    g = np.load(GOLDEN)
    assert np.array_equal(g["input_ids"], ids.numpy()) and np.array_equal(g["position_ids"], pos.numpy())
    assert int(g["n_hidden_states"]) == cfg.num_hidden_layers + 1
    states = O.forward(cfg, sd, ids, mask, pos, px)
    valid = mask.bool()
    for name, idx in (("embeds", 0), ("skip2", -3), ("last", -1)):
        ref = torch.from_numpy(g[name])
        assert (states[idx][valid] - ref[valid]).abs().max().item() <= 2e-5, name
    ids2 = torch.where(ids == cfg.image_token_index, torch.full_like(ids, 7), ids)
    t = O.forward(cfg, sd, ids2, mask, pos, None)[-3]
    assert (t[valid] - torch.from_numpy(g["text_only_skip2"])[valid]).abs().max().item() <= 2e-5
    # the image really conditions the text rows behind it
    assert (states[-3][0, 10] - t[0, 10]).abs().max().item() > 1e-2


def test_expand_input_ids_with_image_tokens():
    ids = torch.tensor([[11, 12, 99, 13, 14, 0, 0], [21, 22, 99, 23, 24, 25, 0]])
    mask = (ids != 0).long()
    out = _expand_input_ids_with_image_tokens(ids, mask, 7, 99, 4, 2, 6, 0)
    assert out["input_ids"].tolist() == [[11, 12, 99, 99, 99, 99, 13, 14, 0, 0], [21, 22, 99, 99, 99, 99, 23, 24, 25, 0]]
    assert out["attention_mask"].tolist() == [[1] * 8 + [0, 0], [1] * 9 + [0]]
    assert out["position_ids"].tolist() == [[0, 1, 2, 3, 4, 5, 6, 7, 1, 1], [0, 1, 2, 3, 4, 5, 6, 7, 8, 1]]


def test_llama_prompt_embeds_cropping_and_interleave():
    """hy:282-420 with stand-ins: the encoder returns each token's EXPANDED position as its embedding, so the result shows
    exactly which positions survive: every 2nd image token first, then the user text without template and assistant header."""
    image_len, start, end, crop = 4, 1, 5, 3
    DR = 271

    class Tok:
        def __call__(self, prompt, max_length=None, **kw):
            # <bos> <image> "\n\n" "\n\n" | user text | <eot> <start_header> assistant <end_header> "\n\n" | padding:
            # crop_start = 3 template tokens go, the 4 header tokens before the LAST double return go, the rest stays
            rows = []
            for p in prompt:
                n_text = 2 if "short" in p else 4
                row = [1, 99, DR, DR] + [50 + i for i in range(n_text)] + [6, 7, 8, 9, DR]
                row = row + [0] * (max_length - len(row))
                rows.append(row[:max_length])
            ids = torch.tensor(rows)
            return SimpleNamespace(input_ids=ids, attention_mask=(ids != 0).long())

    class Enc:
        dtype = torch.float32
        config = SimpleNamespace(image_token_index=99, pad_token_id=0)

        def __call__(self, input_ids, attention_mask, position_ids, pixel_values, output_hidden_states):
            L = input_ids.shape[1]
            h = torch.arange(L, dtype=torch.float32)[None, :, None].expand(input_ids.shape[0], L, 2).clone()
            return SimpleNamespace(hidden_states=[h * 0, h * 0, h, h * 0, h * 0])

    proc = lambda image, return_tensors: SimpleNamespace(pixel_values=torch.zeros(1, 3, 4, 4))
    pipe = HunyuanVideoImageToVideoPipeline(text_encoder=Enc(), tokenizer=Tok(), image_processor=proc)
    tpl = {"template": "sys {}", "crop_start": crop, "image_emb_start": start, "image_emb_end": end, "image_emb_len": image_len,
           "double_return_token_id": DR}
    emb, mask = pipe._get_llama_prompt_embeds(None, ["long prompt", "short prompt"], tpl, max_sequence_length=10,
                                              device=torch.device("cpu"), image_embed_interleave=2)
    # expanded coordinates: text position p >= 2 sits at p + 3 (the one placeholder became 4 image tokens at 1..4).
    # row 0 (4 text tokens): kept text positions 3..7 (2nd double return + the text) -> 6..10, header 8..11 cut, 12 -> 15
    assert emb.shape == (2, 8, 2)
    assert emb[0, :, 0].tolist() == [1.0, 3.0, 6.0, 7.0, 8.0, 9.0, 10.0, 15.0]        # every 2nd image token first
    assert mask[0].tolist() == [1, 1, 1, 1, 1, 1, 1, 1]
    # row 1 (2 text tokens): kept 3..5 -> 6..8, header 6..9 cut, last double return 10 and two pads -> 13..15
    assert emb[1, :, 0].tolist() == [1.0, 3.0, 6.0, 7.0, 8.0, 13.0, 14.0, 15.0]
    assert mask[1].tolist() == [1, 1, 1, 1, 1, 1, 0, 0]


def test_llama_prompt_embeds_truncated_prompt_and_two_placeholder_expansion():
    """A single prompt so long that the assistant header is truncated away keeps only three double returns: the cut-out
    window then ends at the sequence end (hy:361-370).  And the id expansion in its general form: two placeholders."""
    image_len, start, end, crop = 4, 1, 5, 3
    DR = 271

    class Tok:
        def __call__(self, prompt, max_length=None, **kw):
            row = [1, 99, DR, DR, DR] + [50 + i for i in range(40)] + [6, 7, 8, 9, DR]   # three double returns up front, as
            ids = torch.tensor([row[:max_length]])                                      # in the published template
            return SimpleNamespace(input_ids=ids, attention_mask=(ids != 0).long())

    class Enc:
        dtype = torch.float32
        config = SimpleNamespace(image_token_index=99, pad_token_id=0)

        def __call__(self, input_ids, attention_mask, position_ids, pixel_values, output_hidden_states):
            L = input_ids.shape[1]
            h = torch.arange(L, dtype=torch.float32)[None, :, None].expand(1, L, 2).clone()
            return SimpleNamespace(hidden_states=[h * 0, h * 0, h, h * 0, h * 0])

    proc = lambda image, return_tensors: SimpleNamespace(pixel_values=torch.zeros(1, 3, 4, 4))
    pipe = HunyuanVideoImageToVideoPipeline(text_encoder=Enc(), tokenizer=Tok(), image_processor=proc)
    tpl = {"template": "sys {}", "crop_start": crop, "image_emb_start": start, "image_emb_end": end, "image_emb_len": image_len,
           "double_return_token_id": DR}
    emb, mask = pipe._get_llama_prompt_embeds(None, "a very long prompt", tpl, max_sequence_length=10,
                                              device=torch.device("cpu"), image_embed_interleave=2)
    # 13 tokens survive the truncation (positions 0-12, expanded 0-15); text positions 3..8 stay (expanded 6..11), the window
    # [L - 4, L) = 9..12 (expanded 12..15) goes
    assert emb[0, :, 0].tolist() == [1.0, 3.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0]
    assert mask[0].tolist() == [1] * 8
    ids = torch.tensor([[11, 99, 12, 99, 13, 0]])
    out = _expand_input_ids_with_image_tokens(ids, (ids != 0).long(), 6, 99, 3, 1, 4, 0)
    assert out["input_ids"].tolist() == [[11, 99, 99, 99, 12, 0, 0, 0, 13, 0]]      # the 2nd span is the template's business
