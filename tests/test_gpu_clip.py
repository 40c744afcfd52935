"""GPU parity of the CLIP vision tower (wan:228-234): head_dim-80 attention vs the eager graph, the
whole HIP encoder vs the outputs of the real transformers CLIPVisionModel (tests/golden/clip_vectors.npz), and the Wan
pipeline's `encode_image` wiring."""
import os

import numpy as np
import pytest
import torch

from alg_amd import _lib
from alg_amd.image_encoder_clip import CLIPImageProcessor, CLIPVisionEncoderConfig, CLIPVisionModel
from oracle import clip_oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_vectors.npz")
BF = torch.bfloat16


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,H,L", [(2, 4, 17), (1, 16, 257), (1, 2, 448)])
def test_attention_head_dim_80_vs_eager_graph(B, H, L):
    dev = _dev()
    g = torch.Generator().manual_seed(L)
    inner = H * 80
    qkv = torch.randn(B * L, 3 * inner, generator=g).to(BF).to(dev)
    out = torch.empty(B * L, inner, dtype=BF, device=dev)
    scale = 80 ** -0.5
    _lib.attn_bias(qkv, out, None, None, None, B, H, L, scale=scale, head_dim=80)
    q, k, v = [t.view(B, L, H, 80).transpose(1, 2).float() for t in qkv.split(inner, dim=1)]
    s = ((q @ k.transpose(-1, -2)).to(BF) * scale)                       # CLIP eager: matmul, then * scaling (bf16 ops)
    p = torch.softmax(s.float(), dim=-1).to(BF)
    want = (p.float() @ v).to(BF).transpose(1, 2).reshape(B * L, inner)
    diff = (out.float() - want.float()).abs()
    assert diff.max().item() <= 2.0 ** -5 * max(1.0, want.abs().max().item()) and diff.mean().item() < 2e-3
    with pytest.raises(_lib.AlgHipError, match="head_dim"):
        _lib.attn_bias(qkv, out, None, None, None, B, H, L, head_dim=96)


def test_encoder_matches_transformers_golden_vectors():
    vec = np.load(GOLD)
    cfg, sd, px = clip_oracle.golden_inputs()
    model = CLIPVisionModel(CLIPVisionEncoderConfig(**clip_oracle.GOLDEN["cfg"]), device=_dev()).load_state_dict(sd)
    out = model(pixel_values=px.to(_dev()), output_hidden_states=True)
    assert len(out.hidden_states) == int(vec["n_hidden_states"])
    for got, want in ((out.hidden_states[-2], vec["penultimate"]), (out.last_hidden_state, vec["last_hidden_state"])):
        got = got.float().cpu().numpy()
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert got.shape == want.shape and rel < 2e-2, rel       # bf16 weights / activations vs transformers fp32
    assert model(pixel_values=px.to(_dev())).hidden_states is None


def test_wan_pipeline_encode_image_uses_the_penultimate_state():
    from PIL import Image
    from alg_amd import UniPCMultistepScheduler, WanImageToVideoPipeline
    cfg = CLIPVisionEncoderConfig(**clip_oracle.GOLDEN["cfg"])
    enc = CLIPVisionModel.from_synthetic(cfg, seed=2, device=_dev())
    pipe = WanImageToVideoPipeline(image_encoder=enc, image_processor=CLIPImageProcessor(size=56),
                                   transformer=type("T", (), {"dtype": BF, "config": type("C", (), {"patch_size": (1, 2, 2)})()})(),
                                   scheduler=UniPCMultistepScheduler())
    img = Image.fromarray((np.random.RandomState(1).rand(90, 160, 3) * 255).astype("uint8"))
    emb = pipe.encode_image(img, _dev())
    assert emb.shape == (1, 17, 320) and emb.dtype == BF
    px = CLIPImageProcessor(size=56)(images=img)["pixel_values"].to(_dev())
    assert torch.equal(emb, enc(pixel_values=px, output_hidden_states=True).hidden_states[-2])
    both = pipe.encode_image([img, img], _dev())
    assert both.shape == (2, 17, 320) and torch.equal(both[0], both[1])
