import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alg_amd import _lib
which = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda:0"); BF = torch.bfloat16
N, S, D, H = 2, 17776, 3072, 48
S_pad = (S + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
if which == "attn":
    qk = rn(N, S, 2 * D); vt = rn(N, D, S_pad); att = torch.empty(N, S, D, dtype=BF, device=dev)
    fn = lambda: _lib.flash_attn_d64(qk, qk, vt, att, N, H, S, S * 2 * D, 2 * D, D * S_pad, S_pad, S * D, D, 0.125, k_off=D)
    flop = 4.0 * N * H * S * S * 64
elif which == "attn128":
    S, H = 32760, 40
    D = H * 128
    S_pad = (S + 63) // 64 * 64
    q, k = rn(N, S, D), rn(N, S, D)
    vt = rn(N, D, S_pad)
    att = torch.empty(N, S, D, dtype=BF, device=dev)
    fn = lambda: _lib.flash_attn_d128(q, k, vt, att, N, H, S, S, S * D, D, S * D, D, D * S_pad, S_pad, S * D, D, 128 ** -0.5)
    flop = 4.0 * N * H * S * S * 128
else:
    h = rn(N, S, 4 * D); w = rn(D, 4 * D, sc=0.02); x = rn(N, S, D)
    fn = lambda: _lib.gemm(h, w, x, S, D, 4 * D, 4 * D, 4 * D, D, batch=N, strideA=S * 4 * D, strideC=S * D)
    flop = 2.0 * N * S * D * 4 * D
print("ALG_ATTN_VARIANT", os.environ.get("ALG_ATTN_VARIANT"), "ALG_GEMM_PIPE", os.environ.get("ALG_GEMM_PIPE"))
fn(); torch.cuda.synchronize()
t0 = time.time(); n = 0
rates = []
while time.time() - t0 < secs:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20 if which != 'attn128' else 3): fn()
    b.record(); torch.cuda.synchronize()
    reps = 20 if which != 'attn128' else 3
    rates.append(flop * reps / (a.elapsed_time(b) / 1e3) / 1e12); n += reps
print(which, "launches", n, "TF first/min/last", round(rates[0], 1), round(min(rates), 1), round(rates[-1], 1))
