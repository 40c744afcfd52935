"""GPU parity of the Wan-DiT building blocks against plain torch fp32 on the device."""
import math

import pytest
import torch

from alg_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def _perm(n):
    return torch.tensor([(i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1) for i in range(n)], device=DEV)


def make_vt(v, s_pad):
    """v [B, S, H*128] -> V^T [B, H*128, s_pad] with kv index bits 2 and 3 swapped, zero padded."""
    B, S, D = v.shape
    vt = torch.zeros(B, D, s_pad, dtype=BF, device=DEV)
    vt[:, :, _perm(s_pad)[:S]] = v.transpose(1, 2)  # logical s sits at position perm(s)
    return vt


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 300, 300), (2, 3, 777, 257), (1, 2, 520, 512), (1, 1, 31, 64),
                                          (1, 2, 1024, 2050)])
def test_flash_attn_d128_matches_sdpa(B, H, Sq, Skv):
    D = H * 128
    q, k, v = _rand((B, Sq, D), 1), _rand((B, Skv, D), 2), _rand((B, Skv, D), 3)
    s_pad = (Skv + 63) // 64 * 64
    vt = make_vt(v, s_pad)
    o = torch.zeros(B, Sq, D, dtype=BF, device=DEV)
    scale = 1.0 / math.sqrt(128)
    _lib.flash_attn_d128(q, k, vt, o, B, H, Sq, Skv, Sq * D, D, Skv * D, D, D * s_pad, s_pad, Sq * D, D, scale)
    qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
    kh = k.float().view(B, Skv, H, 128).transpose(1, 2)
    vh = v.float().view(B, Skv, H, 128).transpose(1, 2)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh
    ref = ref.transpose(1, 2).reshape(B, Sq, D)
    err = (o.float() - ref).abs().max().item()
    assert err < 2e-2, err  # bf16 P and bf16 output on O(1) values
    assert (o.float() - ref).abs().mean().item() < 2e-3


def test_flash_attn_d128_rejects_bad_arguments():
    q = _rand((1, 64, 128), 4)
    with pytest.raises(_lib.AlgHipError):  # vt row stride shorter than Skv rounded up
        _lib.flash_attn_d128(q, q, q, q.clone(), 1, 1, 64, 100, 64 * 128, 128, 64 * 128, 128, 128 * 64, 64, 64 * 128, 128,
                             0.1)
    with pytest.raises(_lib.AlgHipError):
        _lib.flash_attn_d128(q.cpu(), q, q, q.clone(), 1, 1, 64, 64, 64 * 128, 128, 64 * 128, 128, 128 * 64, 64,
                             64 * 128, 128, 0.1)
