import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alg_amd import _lib
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g, device="cuda") * sc).to(BF)
for M, N, K, use_bias in [(256, 256, 128, False), (256, 256, 128, True), (512, 256, 128, False), (256, 512, 128, False), (256, 520, 128, False),
                          (300, 256, 128, False), (300, 520, 128, True), (256, 256, 192, False), (256, 256, 256, False), (256, 256, 320, False)]:
    a, w, bias = rn(M, K), rn(N, K, sc=0.05), rn(N)
    c0 = torch.full((M, N), 7.0, dtype=BF, device="cuda")
    c1 = torch.full((M, N), 7.0, dtype=BF, device="cuda")
    _lib.gemm(a, w, c0, M, N, K, K, K, N, bias=bias if use_bias else None)
    _lib.gemm(a, _lib.PackedB(w), c1, M, N, K, K, K, N, bias=bias if use_bias else None)
    torch.cuda.synchronize()
    d = (c0.float() - c1.float()).abs()
    bad = (d > 0)
    rows = bad.any(dim=1).nonzero().flatten()
    cols = bad.any(dim=0).nonzero().flatten()
    print(M, N, K, use_bias, "differ:", int(bad.sum()), "max", d.max().item(),
          "rows", (rows.min().item(), rows.max().item()) if len(rows) else None, "cols", (cols.min().item(), cols.max().item()) if len(cols) else None)
