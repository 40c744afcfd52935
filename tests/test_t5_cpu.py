"""CPU: the T5 / UMT5 encoder oracle against the golden vectors produced by the real
transformers implementation (tests/golden/make_t5_golden.py), plus host-side pieces of the product class."""
import os

import numpy as np
import pytest
import torch

from oracle import t5_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t5_vectors.npz")


@pytest.mark.parametrize("name", sorted(t5_oracle.GOLDEN_CASES))
def test_oracle_matches_transformers_outputs(name):
    vec = np.load(GOLD)
    cfg, sd, ids, mask = t5_oracle.golden_inputs(name)
    assert np.array_equal(vec[name + "_ids"], ids.numpy())
    got = t5_oracle.encode(cfg, sd, ids, mask).numpy()
    want = vec[name + "_out"]
    keep = vec[name + "_mask"].astype(bool)            # padded query rows are don't-care for the pipelines (wan:222-226)
    assert np.abs(got - want)[keep].max() <= 2e-5, np.abs(got - want)[keep].max()


def test_relative_position_buckets_known_values():
    rp = torch.tensor([-200, -128, -20, -8, -7, -1, 0, 1, 7, 8, 20, 127, 500])
    b = t5_oracle.relative_position_bucket(rp).tolist()
    assert b[6] == 0 and b[5] == 1 and b[7] == 17 and b[4] == 7 and b[8] == 23      # exact range: |d| < 8
    assert b[0] == 15 and b[1] == 15 and b[12] == 31                               # clamped at max distance
    assert b[3] == 8 and b[9] == 24 and 8 < b[2] < 15


def test_product_class_tables_and_host_logic():
    from alg_amd import _lib
    from alg_amd.text_encoder_t5 import T5EncoderModel, T5EncoderConfig, UMT5EncoderModel
    cfg = T5EncoderConfig()
    assert (cfg.d_model, cfg.d_ff, cfg.num_layers, cfg.num_heads, cfg.d_kv) == (4096, 10240, 24, 64, 64)
    small = T5EncoderConfig(vocab_size=100, d_model=512, d_ff=1024, num_layers=2, num_heads=8)
    m = T5EncoderModel(small, device="cpu")
    assert m.param_shapes() == t5_oracle.param_shapes(t5_oracle.T5Config(vocab_size=100, d_model=512, d_ff=1024,
                                                                         num_layers=2, num_heads=8))
    um = UMT5EncoderModel(small, device="cpu")
    assert "encoder.block.1.layer.0.SelfAttention.relative_attention_bias.weight" in um.param_shapes()
    assert "encoder.block.1.layer.0.SelfAttention.relative_attention_bias.weight" not in m.param_shapes()
    lut = m.bucket_lut(24)
    ctx = torch.arange(24)
    want = t5_oracle.relative_position_bucket(ctx[None, :] - ctx[:, None])
    assert torch.equal(lut[(ctx[None, :] - ctx[:, None]) + 23].long(), want)
    sd = t5_oracle.synthetic_state_dict(t5_oracle.T5Config(vocab_size=100, d_model=512, d_ff=1024, num_layers=2,
                                                           num_heads=8), seed=1)
    m.load_state_dict(sd)
    assert m.w["encoder.block.0.qkv"].shape == (3 * 512, 512)
    assert torch.equal(m.w["encoder.block.0.qkv"][512:1024].float(), sd["encoder.block.0.layer.0.SelfAttention.k.weight"])
    with pytest.raises(_lib.AlgHipError, match="HIP-only"):
        m(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(KeyError):
        m.load_state_dict({})
