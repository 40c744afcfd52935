"""the Wan and HunyuanVideo oracles state the DiT's patch embedding -- the published modules' Conv3d(kernel = stride =
patch) -- as reshape / permute + F.linear, the same patchify-plus-GEMM formulation the product uses.  A shared error in the
(c, dt, dy, dx) ordering would pass every product-vs-oracle test; this pins the unfolded form against the op the reference runs,
F.conv3d(stride = patch).flatten(2).transpose(1, 2), on random input -- temporal patch sizes 1 (Wan, HunyuanVideo) and 2, non-square
spatial patches, C not a power of two."""
import pytest
import torch
import torch.nn.functional as F

from oracle import hy_oracle, wan_oracle


@pytest.mark.parametrize("mod", [wan_oracle, hy_oracle], ids=["wan", "hunyuan"])
@pytest.mark.parametrize("C,D,patch,grid", [(36, 40, (1, 2, 2), (3, 8, 12)), (16, 24, (1, 2, 2), (2, 6, 4)), (5, 8, (2, 2, 3), (4, 6, 9)),
                                            (3, 16, (1, 1, 1), (2, 3, 5))])
def test_unfolded_linear_patch_embedding_equals_conv3d(mod, C, D, patch, grid):
    g = torch.Generator().manual_seed(C * 100 + D)
    x = torch.randn(2, C, *grid, generator=g, dtype=torch.float64)
    w = torch.randn(D, C, *patch, generator=g, dtype=torch.float64)
    b = torch.randn(D, generator=g, dtype=torch.float64)
    ref = F.conv3d(x, w, b, stride=patch).flatten(2).transpose(1, 2)
    got = mod.patch_embed(x, w, b, patch)
    assert got.shape == ref.shape == (2, (grid[0] // patch[0]) * (grid[1] // patch[1]) * (grid[2] // patch[2]), D)
    assert torch.allclose(got, ref, atol=1e-12, rtol=1e-12)
    # token order: (f, h, w) row-major, as flatten(2) of the conv output gives it
    x2 = torch.zeros_like(x)
    x2[:, :, patch[0]:2 * patch[0]] = x[:, :, patch[0]:2 * patch[0]] if grid[0] >= 2 * patch[0] else 0
    if grid[0] >= 2 * patch[0]:
        got2 = mod.patch_embed(x2, w, torch.zeros_like(b), patch)
        per_frame = (grid[1] // patch[1]) * (grid[2] // patch[2])
        assert got2[:, :per_frame].abs().max() == 0 and got2[:, per_frame:2 * per_frame].abs().max() > 0
