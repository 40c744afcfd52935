"""GPU parity of the text-encoder row: kernels vs torch eager on the device, the whole HIP encoder
vs the outputs of the real transformers T5EncoderModel / UMT5EncoderModel (tests/golden/t5_vectors.npz)."""
import os

import numpy as np
import pytest
import torch

from alg_amd import _lib
from alg_amd.text_encoder_t5 import T5EncoderConfig, T5EncoderModel, UMT5EncoderModel
from oracle import t5_oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t5_vectors.npz")
BF = torch.bfloat16


def _dev():
    return torch.device("cuda:0")


def test_embed_layernorm_mul_are_bit_exact_vs_eager():
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    table = torch.randn(50, 128, generator=g).to(BF).to(dev)
    ids = torch.randint(0, 50, (3, 7), generator=g).to(dev)
    out = torch.empty(21, 128, dtype=BF, device=dev)
    _lib.embed_rows(ids, table, out)
    assert torch.equal(out, table[ids.reshape(-1)])
    x = (torch.randn(37, 512, generator=g) * 3).to(BF).to(dev)
    w = (1 + 0.2 * torch.randn(512, generator=g)).to(BF).to(dev)
    y = torch.empty_like(x)
    _lib.t5_layernorm(x, w, y, 37, 512, 1e-6)
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)              # transformers T5LayerNorm, op for op
    want = w * (x * torch.rsqrt(var + 1e-6)).to(BF)
    diff = (y.float() - want.float()).abs()
    assert (diff > 0).float().mean().item() < 0.01 and diff.max().item() <= 2.0 ** -6 * want.abs().max().item()
    a, b = torch.randn(1000, generator=g).to(BF).to(dev), torch.randn(1000, generator=g).to(BF).to(dev)
    o = torch.empty_like(a)
    _lib.mul_bf16(a, b, o)
    assert torch.equal(o, a * b)


@pytest.mark.parametrize("B,H,L,masked,bias", [(2, 8, 24, 0, True), (1, 2, 226, 0, True), (2, 4, 40, 15, True),
                                               (1, 3, 512, 100, True), (2, 2, 65, 3, False)])
def test_attention_with_bias_vs_eager_graph(B, H, L, masked, bias):
    """The eager bf16 graph of T5Attention: matmul -> + (position_bias + mask) -> softmax(fp32) -> bf16 -> matmul."""
    dev = _dev()
    g = torch.Generator().manual_seed(L + H)
    inner = H * 64
    qkv = torch.randn(B * L, 3 * inner, generator=g).to(BF).to(dev)
    table = torch.randn(32, H, generator=g).to(BF).to(dev) if bias else None
    enc = T5EncoderModel(T5EncoderConfig(vocab_size=8, d_model=64, d_ff=64, num_layers=1, num_heads=H), device="cpu")
    lut = enc.bucket_lut(L).to(dev)
    mask = torch.ones(B, L, dtype=torch.int32)
    if masked:
        mask[B - 1, L - masked:] = 0
    mask = mask.to(dev)
    out = torch.empty(B * L, inner, dtype=BF, device=dev)
    _lib.attn_bias(qkv, out, table, lut if bias else None, mask if masked else None, B, H, L, scale=1.0)
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2).float() for t in qkv.split(inner, dim=1)]
    s = (q @ k.transpose(-1, -2)).to(BF)
    if bias:
        ctx = torch.arange(L, device=dev)
        pb = table[lut[(ctx[None, :] - ctx[:, None]) + L - 1].long()].permute(2, 0, 1)[None]      # [1, H, Lq, Lk] bf16
        s = s + pb
    if masked:
        s = s.float().masked_fill(mask[:, None, None, :] == 0, float("-inf")).to(BF)
    p = torch.softmax(s.float(), dim=-1).to(BF)
    want = (p.float() @ v).to(BF).transpose(1, 2).reshape(B * L, inner)
    diff = (out.float() - want.float()).abs()
    # fp32 accumulation order differs from the library matmul: an occasional score lands on the other side of a bf16
    # rounding boundary
    assert diff.max().item() <= 2.0 ** -5 * max(1.0, want.abs().max().item()), diff.max().item()
    assert diff.mean().item() < 2e-3


@pytest.mark.parametrize("name", sorted(t5_oracle.GOLDEN_CASES))
def test_encoder_matches_transformers_golden_vectors(name):
    vec = np.load(GOLD)
    cfg, sd, ids, mask = t5_oracle.golden_inputs(name)
    pc = T5EncoderConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                         num_heads=cfg.num_heads)
    model = (UMT5EncoderModel if cfg.per_layer_bias else T5EncoderModel)(pc, device=_dev()).load_state_dict(sd)
    out = model(ids.to(_dev()), attention_mask=None if mask is None else mask.to(_dev()))
    got = out.last_hidden_state.float().cpu().numpy()
    assert out[0] is out.last_hidden_state and got.shape == vec[name + "_out"].shape
    keep = vec[name + "_mask"].astype(bool)
    want = vec[name + "_out"]
    rel = np.linalg.norm((got - want)[keep]) / np.linalg.norm(want[keep])
    # bf16 weights, activations and residual stream vs transformers in fp32
    assert rel < 2e-2, rel


def test_padded_tokens_do_not_reach_valid_rows():
    cfg, sd, ids, mask = t5_oracle.golden_inputs("umt5")
    pc = T5EncoderConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                         num_heads=cfg.num_heads)
    model = UMT5EncoderModel(pc, device=_dev()).load_state_dict(sd)
    a = model(ids.to(_dev()), attention_mask=mask.to(_dev())).last_hidden_state
    ids2 = ids.clone()
    ids2[mask == 0] = 7
    b = model(ids2.to(_dev()), attention_mask=mask.to(_dev())).last_hidden_state
    keep = mask.bool().to(_dev())
    assert torch.equal(a[keep], b[keep])


def test_cogvideox_pipeline_encodes_prompts_through_the_hip_t5():
    """cog:228-268: `tokenizer(prompt, padding="max_length", max_length=226, ...)` then `text_encoder(ids)[0]`; with the
    HIP encoder attached the prompt path equals passing its embeddings as `prompt_embeds` (the tokenizer here is a stand-in:
    sentencepiece vocabularies are checkpoint files)."""
    from alg_amd import CogVideoXDDIMScheduler, CogVideoXImageToVideoPipeline
    from alg_amd.transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig
    from oracle import dit_oracle
    dev = _dev()
    small = dict(num_attention_heads=8, attention_head_dim=64, in_channels=16, out_channels=8, num_layers=1,
                 time_embed_dim=64, text_embed_dim=128, max_text_seq_length=10, sample_width=12, sample_height=8,
                 sample_frames=9, patch_size=2)
    w = dit_oracle.init_weights(dit_oracle.DiTConfig(**small), seed=4, std=0.05, randomize_affine=True)
    tr = CogVideoXTransformer3DModel(CogVideoXTransformerConfig(**small), {k: v.to(BF) for k, v in w.items()}, device=dev)
    t5 = T5EncoderModel.from_synthetic(T5EncoderConfig(vocab_size=256, d_model=128, d_ff=256, num_layers=2, num_heads=2),
                                       seed=3, device=dev)

    class Tok:
        def __call__(self, texts, padding=None, max_length=None, truncation=None, add_special_tokens=None,
                     return_tensors=None):
            rows = [[ord(ch) % 255 + 1 for ch in t][:max_length - 1] + [1] for t in texts]       # </s> = 1, pad = 0
            ids = torch.tensor([r + [0] * (max_length - len(r)) for r in rows])
            return type("Enc", (), {"input_ids": ids})()

    pipe = CogVideoXImageToVideoPipeline(tokenizer=Tok(), text_encoder=t5, transformer=tr,
                                         scheduler=CogVideoXDDIMScheduler()).to(dev)
    g = torch.Generator().manual_seed(1)
    kw = dict(image=None, image_latents=(torch.randn(1, 1, 8, 8, 12, generator=g) * 0.7).to(BF), height=64, width=96,
              num_frames=9, num_inference_steps=2, use_low_pass_guidance=False, output_type="latent",
              max_sequence_length=10)
    a = pipe(prompt="a red bus", negative_prompt="blurry", generator=torch.Generator().manual_seed(2), **kw).frames
    tok = Tok()
    pe = t5(tok(["a red bus"], max_length=10).input_ids.to(dev))[0]
    ne = t5(tok(["blurry"], max_length=10).input_ids.to(dev))[0]
    b = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, generator=torch.Generator().manual_seed(2), **kw).frames
    assert torch.equal(a, b) and bool(torch.isfinite(a.float()).all())


def test_wan_pipeline_encodes_prompts_through_the_hip_umt5():
    """wan:185-234 on HIP: tokenizer (stand-in) -> UMT5 with the attention mask -> rows past each prompt's length zeroed;
    equals running the encoder by hand, and the negative prompt defaults to ''."""
    from alg_amd import UniPCMultistepScheduler, WanImageToVideoPipeline
    um = UMT5EncoderModel.from_synthetic(T5EncoderConfig(vocab_size=256, d_model=128, d_ff=256, num_layers=2, num_heads=2),
                                         seed=4, device=_dev())

    class Tok:
        def __call__(self, texts, max_length=None, **_):
            ids = torch.zeros(len(texts), max_length, dtype=torch.long)
            mask = torch.zeros(len(texts), max_length, dtype=torch.long)
            for i, t in enumerate(texts):
                row = [ord(ch) % 250 + 2 for ch in t][:max_length - 1] + [1]
                ids[i, :len(row)] = torch.tensor(row)
                mask[i, :len(row)] = 1
            return type("Enc", (), {"input_ids": ids, "attention_mask": mask})()

    t = type("T", (), {"dtype": BF, "config": type("C", (), {"patch_size": (1, 2, 2)})()})()
    pipe = WanImageToVideoPipeline(tokenizer=Tok(), text_encoder=um, transformer=t, scheduler=UniPCMultistepScheduler())
    pe, ne = pipe.encode_prompt(["a  red &amp; blue bus", "fog"], None, True, 1, None, None, 64, _dev())
    assert pe.shape == ne.shape == (2, 64, 128) and pe.dtype == BF
    tok = Tok()(["a red & blue bus", "fog"], max_length=64)            # prompt_clean collapsed spaces / entities
    want = um(tok.input_ids.to(_dev()), tok.attention_mask.to(_dev())).last_hidden_state
    n0, n1 = int(tok.attention_mask[0].sum()), int(tok.attention_mask[1].sum())
    assert torch.equal(pe[0, :n0], want[0, :n0]) and torch.equal(pe[1, :n1], want[1, :n1])
    assert bool((pe[0, n0:] == 0).all()) and bool((pe[1, n1:] == 0).all())
    assert bool((ne[:, 1:] == 0).all()) and bool((ne[:, 0] != 0).any())
