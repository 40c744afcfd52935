"""Shared parity bookkeeping for the -m gpu tests (VERDICT r1 "weak" 3: no tolerance without a measured floor).

Every HIP-vs-oracle comparison of a DiT forward or a sampler is anchored to the noise floor of the reference's OWN
execution mode: the reference runs its transformers in bf16 with plain eager PyTorch ops (`/root/reference/run.py:38,67`),
so the same CPU oracle is run twice on the same inputs -- once in fp32 (the mathematical reference) and once with bf16
weights / activations in eager op order (what the reference executes) -- and the HIP path has to land within
FACTOR x the bf16-eager deviation from the fp32 result, instead of under a bare 3e-2 / 4e-2:

        err(HIP, fp32 oracle)  <=  1.5 * err(bf16-eager oracle, fp32 oracle)

Measured pairs are appended to gpurun_out/parity_floor.jsonl when ALG_PARITY_REPORT is set (the summary committed under
profiles/ comes from there)."""
import json
import os

import torch

FACTOR = 1.5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def check_floor(name, hip, ref32, eager_bf16, factor=FACTOR, sane=0.5):
    """Assert the anchored bound and return (err_hip, err_floor)."""
    e_hip, e_floor = rel(hip, ref32), rel(eager_bf16, ref32)
    if os.environ.get("ALG_PARITY_REPORT"):
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_floor.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, "err_hip_vs_fp32": e_hip, "err_bf16_eager_vs_fp32": e_floor,
                                "ratio": e_hip / max(e_floor, 1e-30), "factor": factor}) + "\n")
    assert torch.isfinite(hip.float()).all(), name
    assert e_floor < sane, "%s: the bf16-eager oracle itself is %.3e away from fp32 -- the case is ill-conditioned" % (name, e_floor)
    assert e_hip <= factor * e_floor, ("%s: HIP %.3e vs fp32 oracle exceeds %.1f x the bf16-eager floor %.3e"
                                       % (name, e_hip, factor, e_floor))
    return e_hip, e_floor
