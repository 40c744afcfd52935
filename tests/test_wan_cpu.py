"""CPU checks for the Wan DiT row (a-6w): the oracle's structure, the product's parameter table and boundary."""
import math

import pytest
import torch

from alg_amd import _lib
from alg_amd.transformer_wan import WanTransformer3DModel, WanTransformerConfig, parameter_shapes
from oracle import wan_oracle


def test_parameter_tables_agree_and_match_the_published_size():
    cfg, ocfg = WanTransformerConfig(), wan_oracle.WanConfig()
    a, b = parameter_shapes(cfg), wan_oracle.param_shapes(ocfg)
    assert a == b
    n = sum(math.prod(s) for s, _ in a.values())
    assert 14.0e9 < n < 16.5e9, n          # Wan2.1-I2V-14B: ~14.3 B in the blocks + embedders
    assert cfg.dim == 5120 and a["blocks.0.ffn.net.0.proj.weight"][0] == (13824, 5120)
    # what diffusers keeps in fp32 (_keep_in_fp32_modules)
    assert a["blocks.3.scale_shift_table"][1] == torch.float32 and a["blocks.3.norm2.weight"][1] == torch.float32
    assert a["condition_embedder.time_embedder.linear_2.weight"][1] == torch.float32
    assert a["blocks.3.attn1.to_q.weight"][1] == torch.bfloat16


def test_oracle_forward_is_deterministic_and_batch_consistent():
    kw = dict(num_attention_heads=2, ffn_dim=256, num_layers=1, text_dim=32, image_dim=32, added_kv_proj_dim=256)
    cfg = wan_oracle.WanConfig(**kw)
    sd = wan_oracle.init_weights(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 36, 2, 4, 6, generator=g)
    txt, img = torch.randn(2, 512, 32, generator=g), torch.randn(2, 257, 32, generator=g)
    t = torch.tensor([900.0, 900.0])
    y = wan_oracle.wan_forward(cfg, sd, x, t, txt, img)
    assert y.shape == (2, 16, 2, 4, 6) and torch.isfinite(y).all()
    y0 = wan_oracle.wan_forward(cfg, sd, x[:1], t[:1], txt[:1], img[:1])
    assert torch.allclose(y0[0], y[0], atol=1e-5)
    # image tokens matter (the I2V cross-attention branch is live), and so does the timestep
    y_img = wan_oracle.wan_forward(cfg, sd, x, t, txt, img + torch.randn(img.shape, generator=g))  # LN: not a rescale
    y_t = wan_oracle.wan_forward(cfg, sd, x, t * 0.5, txt, img)
    assert (y_img - y).abs().max() > 1e-4 and (y_t - y).abs().max() > 1e-4


def test_rope_tables_follow_the_axis_split():
    cfg = wan_oracle.WanConfig()
    cos, sin = wan_oracle.rope_tables(cfg, 3, 4, 6)
    assert cos.shape == (3 * 2 * 3, 64)
    # token (f, h, w): first 22 frequencies depend on f only, next 21 on h, last 21 on w
    tok = lambda f, h, w: (f * 2 + h) * 3 + w
    assert torch.equal(cos[tok(1, 0, 0), :22], cos[tok(1, 1, 2), :22])
    assert torch.equal(cos[tok(0, 1, 0), 22:43], cos[tok(2, 1, 2), 22:43])
    assert torch.equal(sin[tok(0, 0, 2), 43:], sin[tok(2, 1, 2), 43:])
    assert torch.allclose(cos[0], torch.ones(64, dtype=torch.float64)) and torch.allclose(sin[0], torch.zeros(64, dtype=torch.float64))


def test_product_refuses_cpu():
    with pytest.raises(_lib.AlgHipError):
        WanTransformer3DModel(WanTransformerConfig(num_layers=0), {}, device="cpu")


def test_oracle_flf2v_position_embedding_follows_the_published_module():
    """WanImageEmbedding with pos_embed_seq_len (first-last-frame checkpoints; the reference hands `[image, last_image]` to the CLIP
    encoder, /root/reference/pipeline_wan_image2video_lowpass.py:805-812): [2 N, 257, I] is VIEWED as [N, 514, I] -- sample n owns
    rows 2 n (first frame) and 2 n + 1 (last frame) -- and the learned table is added before norm1.  The parameter tables of product
    and oracle agree on the extra tensor; the oracle equals a by-hand restatement of the embedder on its own."""
    kw = dict(num_attention_heads=2, ffn_dim=256, num_layers=1, text_dim=32, image_dim=32, added_kv_proj_dim=256, pos_embed_seq_len=514)
    cfg, pcfg = wan_oracle.WanConfig(**kw), WanTransformerConfig(**kw)
    assert parameter_shapes(pcfg) == wan_oracle.param_shapes(cfg)
    assert parameter_shapes(pcfg)["condition_embedder.image_embedder.pos_embed"] == ((1, 514, 32), torch.bfloat16)
    sd = wan_oracle.init_weights(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 36, 2, 4, 6, generator=g)
    txt, img = torch.randn(2, 512, 32, generator=g), torch.randn(4, 257, 32, generator=g)
    t = torch.tensor([900.0, 900.0])
    y = wan_oracle.wan_forward(cfg, sd, x, t, txt, img)
    assert y.shape == (2, 16, 2, 4, 6) and torch.isfinite(y).all()
    # sample 1 alone = its own (first, last) pair: rows 2, 3 of the image batch
    y1 = wan_oracle.wan_forward(cfg, sd, x[1:], t[1:], txt[1:], img[2:4])
    assert torch.allclose(y1[0], y[1], atol=1e-5)
    # swapping first and last frame changes the result (the table is position dependent), dropping the table as well
    y_swap = wan_oracle.wan_forward(cfg, sd, x, t, txt, img[[1, 0, 3, 2]])
    sd0 = dict(sd)
    sd0["condition_embedder.image_embedder.pos_embed"] = torch.zeros(1, 514, 32, dtype=torch.bfloat16)
    y_nopos = wan_oracle.wan_forward(cfg, sd0, x, t, txt, img)
    assert (y_swap - y).abs().max() > 1e-4 and (y_nopos - y).abs().max() > 1e-4
    # without the table the concatenation order is all that is left of "first / last": the view is [first | last] per sample
    cfg_plain = wan_oracle.WanConfig(**{k: v for k, v in kw.items() if k != "pos_embed_seq_len"})
    sd_plain = {k: v for k, v in sd0.items() if not k.endswith("pos_embed")}
    y_cat = wan_oracle.wan_forward(cfg_plain, sd_plain, x, t, txt, img.reshape(2, 514, 32))
    assert torch.allclose(y_cat, y_nopos, atol=1e-6)
