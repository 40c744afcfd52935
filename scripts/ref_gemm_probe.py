"""Reference point only (not part of the product): what the vendor library GEMM reaches on this box at the C2 shapes,
under the same 1400 W package cap.  python scripts/ref_gemm_probe.py"""
import torch, time
dev = torch.device("cuda:0"); BF = torch.bfloat16
S, D = 17776, 3072
N = 2
shapes = {"qk": (N * S, 2 * D, D), "v": (N * S, D, D), "out": (N * S, D, D), "ff1": (N * S, 4 * D, D), "ff2": (N * S, D, 4 * D)}
g = torch.Generator(device=dev).manual_seed(0)
for name, (M, Nn, K) in shapes.items():
    a = (torch.randn(M, K, generator=g, device=dev)).to(BF)
    w = (torch.randn(Nn, K, generator=g, device=dev) * 0.02).to(BF)
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): torch.nn.functional.linear(a, w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10; n += 10
    print("%-4s M=%d N=%d K=%d  %.3f ms  %.0f TFLOP/s (sustained, after %d launches)" % (name, M, Nn, K, ms, 2.0 * M * Nn * K / ms / 1e9, n), flush=True)
