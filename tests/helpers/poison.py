"""Test helper: the register / LDS / SGPR poison kernel (reg_poison.hip), built in-tree next to its source with hipcc
(`__graft_entry__.build()` builds it too, so the .so travels to the GPU box; a missing .so is built on first use)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libreg_poison.so")
PATTERNS = {"nan": 0x7FC00000, "big": 0x7F000000, "neg": 0xFF000000, "ones": 0x3F803F80, "allbits": 0xFFFFFFFF, "alt": 0xAAAAAAAA,
            "zero": 0, "eighty": 0x42A042A0, "minus80": 0xC2A0C2A0}
VGPR_LO, VGPR_HI, AGPR_LO, AGPR_HI, LDS, SGPR = 1, 2, 4, 8, 16, 32
ALL = 63


def build(force=False):
    src = os.path.join(HERE, "reg_poison.hip")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w",
                        "-o", SO, src], check=True)
    return SO


def load():
    lib = ctypes.CDLL(build())
    lib.reg_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.reg_peek.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib
