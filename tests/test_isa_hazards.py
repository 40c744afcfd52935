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


def _listing(src, extra=()):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Xclang", "-target-feature",
                            "-Xclang", "-packed-fp32-ops", "-I" + os.path.dirname(src), *extra, "-c", src, "-o", os.path.join(d, "k.o"),
                            "-save-temps"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lst = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        assert lst, os.listdir(d)
        return open(os.path.join(d, lst[0])).read()


# (source, extra flags, least number of asm MFMAs with a VGPR destination the listing must contain): both 64-query kernels keep
# their QK MFMAs' results in literal ArchVGPR blocks INSIDE one generated statement; the scan proves that nothing the compiler
# placed (and nothing in the statement) touches such a block in the MFMA's shadow
CASES = [("attention128_q64.hip", (), 150)]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name,flags,least", CASES)
def test_no_vector_instruction_touches_an_asm_mfma_destination_in_its_shadow(name, flags, least, tmp_path):
    """both build flavours, the shelved d = 64 sibling too, and a window that covers a whole 16-pass MFMA: up to 16
    instructions AND through the next MFMA (span = 2: its issue waits for the pipe, i.e. for the scanned MFMA's last pass, whose
    write-back can still be a few cycles out)."""
    import isa_mfma_shadow_scan as scan

    src = os.path.join(ROOT, "alg_amd", "csrc", name)
    if not os.path.exists(src):
        pytest.skip(name + " is not part of this tree")
    path = tmp_path / "k.s"
    path.write_text(_listing(src, flags))
    n_mfma, hits = scan.scan(str(path), min_nops=4, window=16, span=2)
    assert n_mfma >= least, n_mfma            # the kernel's QK MFMAs write VGPR blocks (S^T lands where the VALU reads it)
    assert not hits, hits[:5]
