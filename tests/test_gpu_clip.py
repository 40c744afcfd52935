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


def test_causal_attention_and_quick_gelu_vs_eager():
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, H, L = 2, 3, 77
    inner = H * 64
    qkv = torch.randn(B * L, 3 * inner, generator=g).to(BF).to(dev)
    out = torch.empty(B * L, inner, dtype=BF, device=dev)
    _lib.attn_bias(qkv, out, None, None, None, B, H, L, scale=0.125, head_dim=64, causal=True)
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2).float() for t in qkv.split(inner, dim=1)]
    s = ((q @ k.transpose(-1, -2)).to(BF) * 0.125).float()
    s = s.masked_fill(torch.ones(L, L, device=dev).triu(1).bool(), float("-inf"))
    p = torch.softmax(s, dim=-1).to(BF)
    want = (p.float() @ v).to(BF).transpose(1, 2).reshape(B * L, inner)
    diff = (out.float() - want.float()).abs()
    assert diff.max().item() <= 2.0 ** -5 * max(1.0, want.abs().max().item()) and diff.mean().item() < 2e-3
    x = (torch.randn(4096, generator=g) * 3).to(BF).to(dev)
    want = x * torch.sigmoid(1.702 * x)                        # transformers QuickGELUActivation on a bf16 tensor
    got = _lib.quick_gelu_(x.clone())
    assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -6 * want.abs().max().item()


def test_clip_text_encoder_matches_transformers_golden_vectors():
    from alg_amd.text_encoder_clip import CLIPTextEncoderConfig, CLIPTextModel
    from oracle import clip_text_oracle as cto
    vec = np.load(os.path.join(os.path.dirname(GOLD), "clip_text_vectors.npz"))
    cfg, sd, ids = cto.golden_inputs()
    model = CLIPTextModel(CLIPTextEncoderConfig(**cto.GOLDEN["cfg"]), device=_dev()).load_state_dict(sd)
    out = model(ids.to(_dev()), output_hidden_states=False)
    for got, want in ((out.pooler_output, vec["pooler_output"]), (out.last_hidden_state, vec["last_hidden_state"])):
        got = got.float().cpu().numpy()
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert got.shape == want.shape and rel < 2e-2, rel
    full = CLIPTextModel.from_synthetic(CLIPTextEncoderConfig(num_hidden_layers=2), seed=1, device=_dev())   # CLIP-L widths
    ids = torch.randint(1, 49000, (2, 77))
    ids[:, 20] = 49407
    p = full(ids.to(_dev())).pooler_output
    assert p.shape == (2, 768) and bool(torch.isfinite(p.float()).all())


def test_hunyuan_pipeline_pooled_embedding_from_the_clip_text_tower():
    """hy:421-452: with `text_encoder_2` / `tokenizer_2` attached the pooled prompt embedding comes from the HIP CLIP text
    tower (`clip_prompt` extension kwarg: the Llava tower that would consume `prompt` is not built)."""
    from alg_amd import FlowMatchEulerDiscreteScheduler, HunyuanVideoImageToVideoPipeline
    from alg_amd.text_encoder_clip import CLIPTextEncoderConfig, CLIPTextModel
    enc = CLIPTextModel.from_synthetic(CLIPTextEncoderConfig(vocab_size=300, hidden_size=128, intermediate_size=256,
                                                             num_hidden_layers=2, num_attention_heads=2), seed=2, device=_dev())

    class Tok:
        def __call__(self, texts, padding=None, max_length=None, truncation=None, return_tensors=None):
            rows = [[290] + [ord(ch) % 250 + 1 for ch in t][:max_length - 2] + [299] for t in texts]
            ids = torch.tensor([r + [0] * (max_length - len(r)) for r in rows])
            return type("Enc", (), {"input_ids": ids})()

    t = type("T", (), {"dtype": BF, "config": type("C", (), {"image_condition_type": "token_replace", "in_channels": 16,
                                                               "guidance_embeds": True, "patch_size": 2})()})()
    pipe = HunyuanVideoImageToVideoPipeline(transformer=t, scheduler=FlowMatchEulerDiscreteScheduler(shift=7.0),
                                            text_encoder_2=enc, tokenizer_2=Tok())
    pooled = pipe._get_clip_prompt_embeds(["a red bus", "fog"], device=_dev())
    assert pooled.shape == (2, 128) and pooled.dtype == BF
    ids = Tok()(["a red bus", "fog"], max_length=77).input_ids.to(_dev())
    assert torch.equal(pooled, enc(ids).pooler_output)
    assert int(enc.eos_positions(ids)[1]) == 4            # <bos> f o g <eos>: the pooled row is the end-of-text token
