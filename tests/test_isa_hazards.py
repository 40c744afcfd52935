"""Static check of the 64-queries-per-wave attention's machine code (no GPU needed: hipcc cross-compiles).

attention128_q64.hip issues its MFMAs from inline asm with compiler-allocated VGPR destinations.  hipcc cannot see an MFMA in asm
text: where such a destination is dead it recycled the registers as temporaries while the MFMA was still writing them -- the cause
of the run-to-run mismatches that kept the kernel out of the product for three rounds (profiles/r4_attention128_q64_probe.txt).
scripts/isa_mfma_shadow_scan.py finds that pattern in the listing; the fixed kernel must have none."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_vector_instruction_touches_an_asm_mfma_destination_in_its_shadow():
    import isa_mfma_shadow_scan as scan

    src = os.path.join(ROOT, "alg_amd", "csrc", "attention128_q64.hip")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Xclang", "-target-feature",
                            "-Xclang", "-packed-fp32-ops", "-I" + os.path.dirname(src), "-c", src, "-o", os.path.join(d, "k.o"),
                            "-save-temps"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lst = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        assert lst, os.listdir(d)
        n_mfma, hits = scan.scan(os.path.join(d, lst[0]))
    assert n_mfma >= 150, n_mfma            # the kernel's QK MFMAs write VGPR blocks (S^T lands where the VALU reads it)
    assert not hits, hits[:5]
