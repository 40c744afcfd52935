"""Reference point only: vendor-library GEMM (torch linear) looping on the ff2 shape, for power_probe.sh."""
import sys, time, torch
dev = torch.device("cuda:0"); BF = torch.bfloat16
S, D, N = 17776, 3072, 2
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(N * S, 4 * D, generator=g, device=dev).to(BF)
w = (torch.randn(D, 4 * D, generator=g, device=dev) * 0.02).to(BF)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
torch.nn.functional.linear(a, w); torch.cuda.synchronize()
t0 = time.time(); rates = []
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    rates.append(2.0 * N * S * D * 4 * D * 20 / (e0.elapsed_time(e1) / 1e3) / 1e12)
print("vendor ff2 TF first/min/last", round(rates[0], 1), round(min(rates), 1), round(rates[-1], 1))
