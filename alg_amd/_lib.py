"""ctypes binding of libalg_hip.so (C ABI: include/alg_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every hot-path computation is a
hand-written gfx950 kernel behind the C ABI.  There is NO CPU fallback: if the library is missing or a
call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALG_HIP_LIB") or os.path.join(_HERE, "libalg_hip.so")  # env override: A/B builds

ALG_F32, ALG_BF16 = 0, 1
ACT_NONE, ACT_GELU_TANH, ACT_SILU = 0, 1, 2
GEMM_BIAS_PER_ROW, GEMM_PERMUTE_COLS, GEMM_GATE_F32, GEMM_GATE_SEG_STRIDE = 1, 4, 8, 16
GEMM_B_PACKED11 = 32     # B is the panel pack_b_p11 wrote (GEMM schedule 11: the weight straight into registers in fragment order)

EXPORTS = (
    "alg_version", "alg_last_error", "alg_reload_env", "alg_down_up", "alg_gaussian_blur", "alg_cfg_ddim_step", "alg_gemm_bf16", "alg_gemm_bf16_pair", "alg_gemm_bf16_pair_qk", "alg_pack_b_p11", "alg_pack_b_p11_bytes",
    "alg_flash_attn_d64", "alg_layernorm_modulate", "alg_qk_norm_rope", "alg_patchify", "alg_unpatchify",
    "alg_timestep_embedding", "alg_cfg_combine", "alg_lincomb", "alg_unipc_update", "alg_concat_cast", "alg_flash_attn_d128", "alg_layernorm_mod_f32", "alg_layernorm_mod_f32_fp8", "alg_rmsnorm_rope",
    "alg_wan_modulation", "alg_patchify3d", "alg_unpatchify3d", "alg_timestep_embedding_f32", "alg_linear_f32",
    "alg_gelu_erf", "alg_layernorm_modulate_seg", "alg_headnorm_rope", "alg_masked_mean", "alg_silu", "alg_gemm_fp8", "alg_quantize_fp8_rows",
    "alg_conv_cl_bf16", "alg_vae_groupnorm_workspace", "alg_vae_groupnorm_stats", "alg_vae_spatial_norm", "alg_vae_upsample",
    "alg_vae_pack_latent", "alg_vae_unpack_video", "alg_vae_group_norm", "alg_vae_pad", "alg_vae_repitch",
    "alg_vae_unpack_planes", "alg_rms_norm_rows", "alg_softmax_hilo", "alg_flash_attn_d128_ex", "alg_flash_attn_d128_dual", "alg_rope_half", "alg_patchify_t", "alg_unpatchify_t", "alg_qk_norm_rope_scaled", "alg_flash_attn_d64_ex", "alg_embed_rows", "alg_t5_layernorm", "alg_attn_bias", "alg_mul_bf16", "alg_quick_gelu",
    "alg_lowpass_tables_bytes", "alg_lowpass_tables_build", "alg_down_up_workspace_bytes", "alg_gaussian_blur_workspace_bytes",
    "alg_flash_attn_d64_workspace_bytes", "alg_calib_mfma_bf16", "alg_wall_clock_khz", "alg_attn_clock_tap",
)
_RET_I64 = ("alg_vae_groupnorm_workspace", "alg_lowpass_tables_bytes", "alg_down_up_workspace_bytes",
            "alg_gaussian_blur_workspace_bytes", "alg_flash_attn_d64_workspace_bytes", "alg_pack_b_p11_bytes")


class AlgHipError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("R", c_void_p), ("gate", c_void_p),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldr", c_int64),
        ("strideA", c_int64), ("strideB", c_int64), ("strideC", c_int64), ("strideR", c_int64),
        ("strideGate", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("batch", c_int32),
        ("seg_split", c_int32), ("act", c_int32), ("flags", c_int32),
        ("gate_seg_stride", c_int64), ("perm_col0", c_int32), ("conv_cin_log2", c_int32),
        ("a_scale", c_void_p), ("b_scale", c_void_p), ("strideAScale", c_int64), ("strideBScale", c_int64),
        ("conv_wp", c_int32), ("conv_hpwp", c_int32), ("conv_kw", c_int32), ("reserved1", c_int32),
    ]


class QkNormRopeArgs(Structure):
    _fields_ = [("wq", c_void_p), ("bq", c_void_p), ("wk", c_void_p), ("bk", c_void_p), ("cos_tab", c_void_p),
                ("sin_tab", c_void_p), ("heads", c_int32), ("text_len", c_int32), ("eps", c_float),
                ("q_scale", c_float)]


class VaeGeom(Structure):
    _fields_ = [(n, c_int32) for n in ("frames", "H", "W", "C", "first_len", "seg_len", "lat_first_single", "lat_rate",
                                       "lat_scale", "lat_h", "lat_w")]


_lib = None


def build_library(verbose=False):
    """Compile libalg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    import subprocess

    src = os.path.join(_HERE, "csrc")
    r = subprocess.run(["make", "-C", src, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise AlgHipError("building libalg_hip.so failed (see output above)")
    return LIB_PATH


def load_library():
    """dlopen the library and declare the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AlgHipError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C alg_amd/csrc`).  There is no CPU fallback for the ALG hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.alg_version.restype = c_int
    lib.alg_last_error.restype = c_char_p
    lib.alg_reload_env.restype = None
    lib.alg_reload_env.argtypes = []
    lib.alg_down_up.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_int64, c_void_p]
    lib.alg_gaussian_blur.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int64,
                                      c_void_p]
    lib.alg_lowpass_tables_bytes.argtypes = [c_int] * 4
    lib.alg_lowpass_tables_build.argtypes = [c_void_p, c_int64] + [c_int] * 4 + [c_void_p]
    lib.alg_down_up_workspace_bytes.argtypes = [c_int64] + [c_int] * 4
    lib.alg_gaussian_blur_workspace_bytes.argtypes = [c_int64] + [c_int] * 3
    lib.alg_flash_attn_d64_workspace_bytes.argtypes = [c_int] * 4
    lib.alg_pack_b_p11_bytes.argtypes = [c_int, c_int]
    lib.alg_pack_b_p11_bytes.restype = c_int64
    lib.alg_pack_b_p11.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]
    lib.alg_cfg_ddim_step.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_float, c_float, c_float,
                                      c_float, c_float, c_void_p]
    lib.alg_cfg_combine.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_void_p]
    lib.alg_lincomb.argtypes = [POINTER(c_void_p), POINTER(c_float), POINTER(c_int), c_int, c_void_p, c_int, c_int64,
                                c_void_p]
    lib.alg_concat_cast.argtypes = [POINTER(c_void_p), c_int, POINTER(c_void_p), c_int, c_int] + [c_int64] * 7 + [
        c_void_p, c_int, c_void_p]
    lib.alg_flash_attn_d128.argtypes = [c_void_p] * 4 + [c_int] * 4 + [c_int64] * 8 + [c_float, c_void_p]
    lib.alg_flash_attn_d128_ex.argtypes = [c_void_p] * 4 + [c_int] * 4 + [c_int64] * 8 + [c_float, c_int, c_int, c_void_p]
    lib.alg_flash_attn_d128_dual.argtypes = ([c_void_p] * 3 + [c_int] + [c_int64] * 4 + [c_void_p] * 2 + [c_int] + [c_int64] * 4 +
                                             [c_void_p] + [c_int] * 3 + [c_int64] * 4 + [c_float, c_void_p])
    lib.alg_rope_half.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]
    lib.alg_layernorm_mod_f32.argtypes = [c_void_p] * 6 + [c_int64, c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_layernorm_mod_f32_fp8.argtypes = [c_void_p] * 7 + [c_int64, c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_rmsnorm_rope.argtypes = [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_wan_modulation.argtypes = [c_void_p] * 3 + [c_int] * 5 + [c_void_p]
    lib.alg_patchify3d.argtypes = [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]
    lib.alg_unpatchify3d.argtypes = [c_void_p, c_int64, c_void_p] + [c_int] * 8 + [c_void_p]
    lib.alg_layernorm_modulate_seg.argtypes = [c_void_p] * 6 + [c_int64, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                               c_int, c_float, c_void_p]
    lib.alg_headnorm_rope.argtypes = [c_void_p] * 4 + [c_int64, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_masked_mean.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.alg_silu.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    lib.alg_rms_norm_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]
    lib.alg_softmax_hilo.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, ctypes.c_float, c_void_p]
    lib.alg_gemm_fp8.argtypes = [POINTER(GemmArgs), c_void_p]
    lib.alg_conv_cl_bf16.argtypes = [c_void_p] * 5 + [c_int] * 7 + [c_void_p]
    lib.alg_vae_groupnorm_workspace.argtypes = [POINTER(VaeGeom)]
    lib.alg_vae_groupnorm_stats.argtypes = [c_void_p, POINTER(VaeGeom), c_float, c_void_p, c_void_p, c_void_p]
    lib.alg_vae_spatial_norm.argtypes = [c_void_p] * 6 + [POINTER(VaeGeom), c_int, c_void_p]
    lib.alg_vae_group_norm.argtypes = [c_void_p] * 5 + [POINTER(VaeGeom), c_int, c_void_p]
    lib.alg_vae_pad.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]
    lib.alg_vae_repitch.argtypes = [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]
    lib.alg_vae_unpack_planes.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]
    lib.alg_embed_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]
    lib.alg_t5_layernorm.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]
    lib.alg_attn_bias.argtypes = [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_int64, c_int64, c_float, c_int, c_void_p]
    lib.alg_quick_gelu.argtypes = [c_void_p, c_int64, c_void_p]
    lib.alg_mul_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.alg_vae_upsample.argtypes = [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]
    lib.alg_vae_pack_latent.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_vae_unpack_video.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.alg_quantize_fp8_rows.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p]
    lib.alg_timestep_embedding_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.alg_linear_f32.argtypes = [c_void_p] * 6 + [c_int] * 4 + [c_void_p]
    lib.alg_gelu_erf.argtypes = [c_void_p, c_int64, c_void_p]
    lib.alg_unipc_update.argtypes = [c_void_p] * 5 + [c_int64] + [c_float] * 6 + [c_void_p]
    lib.alg_gemm_bf16.argtypes = [POINTER(GemmArgs), c_void_p]
    lib.alg_gemm_bf16_pair.argtypes = [POINTER(GemmArgs), POINTER(GemmArgs), c_void_p]
    lib.alg_gemm_bf16_pair_qk.argtypes = [POINTER(GemmArgs), POINTER(GemmArgs), POINTER(QkNormRopeArgs), c_void_p]
    lib.alg_flash_attn_d64.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64,
                                       c_int64, c_int64, c_int64, c_int64, c_float, c_void_p]
    lib.alg_layernorm_modulate.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                           c_int, c_int, c_int64, c_int64, c_int, c_float, c_void_p]
    lib.alg_qk_norm_rope.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int, c_int, c_int, c_float, c_void_p]
    lib.alg_qk_norm_rope_scaled.argtypes = [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]
    lib.alg_flash_attn_d64_ex.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64,
                                          c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_void_p, c_int64, c_void_p]
    lib.alg_patchify.argtypes = [c_void_p, c_int64, POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_void_p]
    lib.alg_unpatchify.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.alg_patchify_t.argtypes = [c_void_p, c_int64, POINTER(c_void_p), c_void_p] + [c_int] * 7 + [c_void_p]
    lib.alg_unpatchify_t.argtypes = [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]
    lib.alg_timestep_embedding.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.alg_calib_mfma_bf16.argtypes = [c_void_p, c_int, ctypes.c_uint, c_int, c_void_p, c_void_p]
    lib.alg_attn_clock_tap.argtypes = [c_void_p, c_int]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name == "alg_attn_clock_tap":
            fn.restype = None
        elif name not in ("alg_version", "alg_last_error", "alg_reload_env"):
            fn.restype = c_int64 if name in _RET_I64 else c_int
    _lib = lib
    return lib


def reload_env():
    """The library reads its ALG_* options once, at load (include/alg_hip.h); call this after changing one in os.environ."""
    load_library().alg_reload_env()


def _check(rc, what):
    if rc != 0:
        msg = load_library().alg_last_error()
        raise AlgHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if t.dtype == torch.float32:
        return ALG_F32
    if t.dtype == torch.bfloat16:
        return ALG_BF16
    raise AlgHipError("unsupported dtype %s (the HIP path handles float32 and bfloat16)" % t.dtype)


def _dev(t, name):
    if not t.is_cuda:
        raise AlgHipError("%s must be a device tensor: the ALG hot path is HIP-only (no CPU fallback)" % name)
    return t


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


# ---------------------------------------------------------------------------------------------------
# thin typed wrappers (shape/dtype checks live here; arithmetic lives in the kernels)
# ---------------------------------------------------------------------------------------------------

# The library never allocates (include/alg_hip.h): scratch space and the down_up tap tables are torch tensors owned here.
# Tables depend on the shape only, so they are built once per (device, stream, shape) and kept in a small LRU; a scratch
# buffer is kept per (device, stream, tag) and only ever grows -- launches on one stream are ordered, so the calls that
# share it never overlap.
#
# hipGraph capture: a captured launch holds the RAW pointer of the table / scratch buffer it was handed, so
#   * a buffer that was handed out while the stream was capturing is PINNED: it leaves the caches like any other (LRU eviction,
#     a larger scratch request) but is never freed -- a replay may dereference it at any later time (release_captured_buffers()
#     is the explicit way out once the graphs are gone);
#   * a table first BUILT inside a capture is filled by the replay, not now: it is pinned for the graph and NOT put into the
#     eager cache, where a later eager call would read it before any replay has run.
_TABLES = OrderedDict()
_TABLES_MAX = 64
_SCRATCH = {}
_PINNED = {}          # id(tensor) -> tensor


def _stream_key(t):
    return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _pin(t):
    _PINNED[id(t)] = t
    return t


def release_captured_buffers():
    """Drop the buffers pinned for hipGraph captures.  Only call this when every graph captured so far has been destroyed."""
    _PINNED.clear()


def clear_caches():
    """Drop the table / scratch caches (a long-running host that has retired a stream); pinned buffers stay."""
    _TABLES.clear()
    _SCRATCH.clear()


def lowpass_tables(x, h1, w1):
    """The caller-owned tap-table blob of alg_down_up for x's plane shape (alg_lowpass_tables_build, cached)."""
    lib = load_library()
    H, W = x.shape[-2:]
    key = _stream_key(x) + (H, W, h1, w1)
    cap = _capturing()
    t = _TABLES.get(key)
    if t is not None:
        _TABLES.move_to_end(key)
        return _pin(t) if cap else t
    n = int(lib.alg_lowpass_tables_bytes(H, W, h1, w1))
    if n <= 0:
        return None
    t = torch.empty(n, dtype=torch.uint8, device=x.device)
    _check(lib.alg_lowpass_tables_build(_ptr(t), n, H, W, h1, w1, _stream()), "alg_lowpass_tables_build")
    if cap:
        return _pin(t)          # built by the graph's own replay: lives with the graph, invisible to eager calls
    _TABLES[key] = t
    while len(_TABLES) > _TABLES_MAX:
        _TABLES.popitem(last=False)
    return t


def scratch(ref, nbytes, tag):
    """A 16-byte aligned uint8 buffer of >= nbytes on ref's device for the current stream (None when nbytes == 0)."""
    if nbytes <= 0:
        return None
    key = _stream_key(ref) + (tag,)
    t = _SCRATCH.get(key)
    cap = _capturing()
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=ref.device)
        if cap:
            # allocated inside a capture: the block lives in the graph's private pool and belongs to that graph's replays.
            # It must not reach later EAGER calls (a replay on another stream could run beside them on the same bytes;
            # ADVICE r4) -- same rule as lowpass_tables: pinned for the graph, never entered into the eager cache.
            return _pin(t)
        _SCRATCH[key] = t       # a superseded buffer that a capture saw stays alive in _PINNED
    return _pin(t) if cap else t


def calib_mfma_bf16(sink, iters, seed=1, blocks=0, clocks=None):
    """One launch of the register-only random-operand MFMA loop (csrc/calibrate.hip) on the current stream; returns the number
    of workgroups launched (each 4 waves x 32 * iters MFMAs of 32,768 FLOP)."""
    lib = load_library()
    if clocks is not None:
        # the kernel writes 4 uint64 per workgroup: resolve "two per CU" here so the buffer can be checked against it
        if int(blocks) <= 0:
            blocks = 2 * torch.cuda.get_device_properties(sink.device).multi_processor_count
        if not (clocks.is_cuda and clocks.device == sink.device and clocks.dtype == torch.int64 and clocks.is_contiguous()
                and clocks.numel() >= 4 * int(blocks)):
            raise AlgHipError("calib_mfma_bf16: clocks must be a contiguous int64 tensor on the sink's device with >= 4 * %d "
                              "elements" % int(blocks))
    rc = lib.alg_calib_mfma_bf16(_ptr(_dev(sink, "sink")), int(iters), int(seed), int(blocks), _ptr(clocks), _stream())
    if rc <= 0:
        _check(rc if rc < 0 else -1, "alg_calib_mfma_bf16")
    return rc


def wall_clock_khz():
    return int(load_library().alg_wall_clock_khz())


def attn_clock_tap(buffer):
    """buffer: int64 device tensor [slots, 4] (or None to switch the taps off) -- see include/alg_hip.h: alg_attn_clock_tap."""
    lib = load_library()
    if buffer is None:
        lib.alg_attn_clock_tap(None, 0)
        return
    if not (buffer.is_cuda and buffer.dtype == torch.int64 and buffer.is_contiguous() and buffer.dim() == 2 and buffer.shape[1] == 4):
        raise AlgHipError("attn_clock_tap needs a contiguous int64 device tensor [slots, 4]")
    lib.alg_attn_clock_tap(_ptr(buffer), int(buffer.shape[0]))


def clock_mhz_from_taps(taps, wall_khz):
    """taps: int64 [n, 4] = {cycles0, wall0, cycles1, wall1} per sampled workgroup -> (mean MHz, min, max, n used)."""
    t = taps.detach().cpu().to(torch.float64)
    dc, dw = t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]
    ok = (dc > 0) & (dw > 0)
    if not bool(ok.any()) or wall_khz <= 0:
        return None
    mhz = dc[ok] / dw[ok] * (wall_khz / 1e3)
    return {"mean": float(mhz.mean()), "min": float(mhz.min()), "max": float(mhz.max()), "workgroups": int(ok.sum())}


def down_up(x, h1, w1, round_intermediate=True):
    """x: [..., H, W] contiguous device tensor -> new tensor, antialiased bilinear down to (h1, w1) and back."""
    lib = load_library()
    _dev(x, "x")
    if not x.is_contiguous():
        raise AlgHipError("down_up needs a contiguous tensor")
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W) if x.numel() else 0
    out = torch.empty_like(x)
    if planes == 0:
        return out
    wsb = int(lib.alg_down_up_workspace_bytes(planes, H, W, h1, w1))
    ws = scratch(x, wsb, "down_up")
    tables = lowpass_tables(x, h1, w1) if wsb == 0 else None      # the global-memory passes derive their own taps
    _check(lib.alg_down_up(_ptr(x), _ptr(out), planes, H, W, h1, w1, _dt(x), 1 if round_intermediate else 0,
                           _ptr(tables), _ptr(ws), wsb, _stream()), "alg_down_up")
    return out


def gaussian_blur(x, ksize, sigma):
    lib = load_library()
    _dev(x, "x")
    if not x.is_contiguous():
        raise AlgHipError("gaussian_blur needs a contiguous tensor")
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W) if x.numel() else 0
    out = torch.empty_like(x)
    wsb = int(lib.alg_gaussian_blur_workspace_bytes(planes, H, W, int(ksize))) if planes else 0
    ws = scratch(x, wsb, "gaussian")
    _check(lib.alg_gaussian_blur(_ptr(x), _ptr(out), planes, H, W, int(ksize), float(sigma), _dt(x), _ptr(ws), wsb,
                                 _stream()), "alg_gaussian_blur")
    return out


def cfg_ddim_step_(pred, latents, n_pass, guidance_scale, sqrt_alpha_t, sqrt_beta_t, coef_a, coef_b):
    """In place on ``latents``; pred [n_pass, *latents.shape[1:]...] flattened as [n_pass, numel]."""
    lib = load_library()
    _dev(pred, "pred")
    _dev(latents, "latents")
    if not (pred.is_contiguous() and latents.is_contiguous()):
        raise AlgHipError("cfg_ddim_step_ needs contiguous tensors")
    numel = latents.numel()
    if pred.numel() != n_pass * numel:
        raise AlgHipError("pred has %d elements, expected n_pass*numel = %d" % (pred.numel(), n_pass * numel))
    _check(lib.alg_cfg_ddim_step(_ptr(pred), _dt(pred), _ptr(latents), _dt(latents), n_pass, numel,
                                 float(guidance_scale), float(sqrt_alpha_t), float(sqrt_beta_t), float(coef_a),
                                 float(coef_b), _stream()), "alg_cfg_ddim_step")
    return latents


def cfg_combine(pred, n_pass, guidance_scale):
    """pred [n_pass * B, ...] -> [B, ...] = u0 + g * (text - u), rounded per op in pred's dtype."""
    lib = load_library()
    _dev(pred, "pred")
    if not pred.is_contiguous() or pred.shape[0] % n_pass:
        raise AlgHipError("cfg_combine needs a contiguous [n_pass * B, ...] tensor")
    out = torch.empty((pred.shape[0] // n_pass,) + tuple(pred.shape[1:]), device=pred.device, dtype=pred.dtype)
    _check(lib.alg_cfg_combine(_ptr(pred), _ptr(out), _dt(pred), n_pass, out.numel(), float(guidance_scale), _stream()),
           "alg_cfg_combine")
    return out


def lincomb(terms, out_dtype, out=None):
    """sum_i c_i * x_i for 1..4 (coef, tensor) pairs of equal shape -> tensor of ``out_dtype`` (``out`` may alias a term)."""
    lib = load_library()
    xs = [t for _, t in terms]
    for t in xs:
        _dev(t, "term")
        if not t.is_contiguous() or t.shape != xs[0].shape:
            raise AlgHipError("lincomb needs contiguous tensors of one shape")
    n = len(xs)
    if out is None:
        out = torch.empty(xs[0].shape, device=xs[0].device, dtype=out_dtype)
    elif out.dtype != out_dtype or out.shape != xs[0].shape or not out.is_contiguous():
        raise AlgHipError("lincomb: `out` must be a contiguous tensor of the requested dtype and shape")
    arr = (c_void_p * n)(*[t.data_ptr() for t in xs])
    cf = (c_float * n)(*[float(c) for c, _ in terms])
    dts = (c_int * n)(*[_dt(t) for t in xs])
    _check(lib.alg_lincomb(arr, cf, dts, n, _ptr(out), ALG_F32 if out_dtype == torch.float32 else ALG_BF16,
                           out.numel(), _stream()), "alg_lincomb")
    return out


def concat_cast(src0, src1, O, A0, A1, R, s0_ostride, s1_ostride, a1_off, out_dtype):
    """out[i, o, a, r] = a < A0 ? src0[i][o, a, r] : src1[i][o, a1_off + a - A0, r]; src0/src1: lists of tensors
    (contiguous device tensors, one per output sample).  Returns a flat [n, O, A0 + A1, R] tensor of out_dtype."""
    lib = load_library()
    n = len(src0)
    for t in list(src0) + list(src1):
        _dev(t, "concat source")
        if not t.is_contiguous():
            raise AlgHipError("concat_cast needs contiguous sources")
    out = torch.empty((n, O, A0 + A1, R), device=src0[0].device, dtype=out_dtype)
    a0 = (c_void_p * n)(*[t.data_ptr() for t in src0])
    a1 = (c_void_p * n)(*[t.data_ptr() for t in src1])
    _check(lib.alg_concat_cast(a0, _dt(src0[0]), a1, _dt(src1[0]), n, O, A0, A1, R, s0_ostride, s1_ostride, a1_off,
                               _ptr(out), ALG_F32 if out_dtype == torch.float32 else ALG_BF16, _stream()),
           "alg_concat_cast")
    return out


def rope_half_(x, cos, sin, pos, rows, heads, x_rstride, x_off=0):
    """Llama rotary embedding (rotate_half form) in place on [rows][heads][128] bf16."""
    _check(load_library().alg_rope_half(_p(x, x_off), _p(cos), _p(sin), _p(pos), rows, heads, x_rstride, _stream()),
           "alg_rope_half")
    return x


def flash_attn_d128_dual(q, k, vt, Skv, k_bs, k_rs, vt_bs, vt_rs, k2, vt2, Skv2, k2_bs, k2_rs, vt2_bs, vt2_rs, o, batch, heads, Sq,
                         q_bs, q_rs, o_bs, o_rs, scale):
    """o = bf16(bf16(attn(q, k, vt)) + bf16(attn(q, k2, vt2))) in one launch, head_dim 128: the text + image cross-attention of the
    Wan I2V DiT (two short key / value sets for the same queries).  Bit-identical to two flash_attn_d128 calls + lincomb."""
    lib = load_library()
    for t in (q, k, vt, k2, vt2, o):
        _dev(t, "attention operand")
    _check(lib.alg_flash_attn_d128_dual(_p(q), _p(k), _p(vt), Skv, k_bs, k_rs, vt_bs, vt_rs, _p(k2), _p(vt2), Skv2, k2_bs, k2_rs,
                                        vt2_bs, vt2_rs, _p(o), batch, heads, Sq, q_bs, q_rs, o_bs, o_rs, float(scale), _stream()),
           "alg_flash_attn_d128_dual")
    return o


def flash_attn_d128(q, k, vt, o, batch, heads, Sq, Skv, q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs, scale,
                    q_off=0, k_off=0, vt_off=0, o_off=0, kv_group=1, causal=False):
    """softmax(q k^T * scale) v for head_dim 128; strides / offsets in elements (see include/alg_hip.h); kv_group > 1:
    grouped-query attention; causal: query i sees keys 0..i."""
    lib = load_library()
    for t in (q, k, vt, o):
        _dev(t, "attention operand")
    at = lambda t, off: c_void_p(t.data_ptr() + 2 * off)
    if kv_group != 1 or causal:
        _check(lib.alg_flash_attn_d128_ex(at(q, q_off), at(k, k_off), at(vt, vt_off), at(o, o_off), batch, heads, Sq, Skv, q_bs,
                                          q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs, float(scale), int(kv_group),
                                          int(bool(causal)), _stream()), "alg_flash_attn_d128_ex")
        return o
    _check(lib.alg_flash_attn_d128(at(q, q_off), at(k, k_off), at(vt, vt_off), at(o, o_off), batch, heads, Sq, Skv, q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs, float(scale),
                                   _stream()), "alg_flash_attn_d128")
    return o


def _p(t, off=0):
    """device pointer of tensor t advanced by off ELEMENTS (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    _dev(t, "tensor")
    return c_void_p(t.data_ptr() + off * t.element_size())


def layernorm_mod_f32(x, y, weight, bias, scale, shift, mod_bstride, batch, rows, D, eps, scale_off=0, shift_off=0):
    _check(load_library().alg_layernorm_mod_f32(_p(x), _p(y), _p(weight), _p(bias), _p(scale, scale_off),
                                                _p(shift, shift_off), mod_bstride, batch, rows, D, float(eps), _stream()),
           "alg_layernorm_mod_f32")
    return y


def layernorm_mod_f32_fp8(x, q8, q8_scale, weight, bias, scale, shift, mod_bstride, batch, rows, D, eps, scale_off=0,
                          shift_off=0):
    """layernorm_mod_f32 followed by quantize_fp8_rows, in one pass (bit-identical bytes and scales)."""
    _check(load_library().alg_layernorm_mod_f32_fp8(_p(x), _p(q8), _p(q8_scale), _p(weight), _p(bias),
                                                    _p(scale, scale_off), _p(shift, shift_off), mod_bstride, batch, rows,
                                                    D, float(eps), _stream()), "alg_layernorm_mod_f32_fp8")
    return q8


def rmsnorm_rope_(x, weight, cos, sin, x_rstride, batch, rows, D, eps, x_off=0):
    _check(load_library().alg_rmsnorm_rope(_p(x, x_off), _p(weight), _p(cos), _p(sin), x_rstride, batch, rows, D,
                                           float(eps), _stream()), "alg_rmsnorm_rope")
    return x


def wan_modulation(table, vec, out, layers, batch, J, D, vec_per_j):
    _check(load_library().alg_wan_modulation(_p(table), _p(vec), _p(out), layers, batch, J, D, int(vec_per_j), _stream()),
           "alg_wan_modulation")
    return out


def patchify3d(x, out, n, C, F, H, W, ph, pw, Kpad):
    _check(load_library().alg_patchify3d(_p(x), _p(out), n, C, F, H, W, ph, pw, Kpad, _stream()), "alg_patchify3d")
    return out


def unpatchify3d(x, ldin, out, n, C, F, H, W, ph, pw, channel_major=False):
    _check(load_library().alg_unpatchify3d(_p(x), ldin, _p(out), n, C, F, H, W, ph, pw, int(channel_major), _stream()),
           "alg_unpatchify3d")
    return out


def layernorm_modulate_seg(x, y, weight, bias, scale, shift, mod_bstride, seg_stride, batch, rows, D, seg_split, eps,
                           x_bstride=None, y_bstride=None, x_off=0, y_off=0, scale_off=0, shift_off=0):
    """alg_layernorm_modulate with an explicit distance between the two row segments' modulation vectors."""
    x_bstride = rows * D if x_bstride is None else x_bstride
    y_bstride = rows * D if y_bstride is None else y_bstride
    _check(load_library().alg_layernorm_modulate_seg(_p(x, x_off), _p(y, y_off), _p(weight), _p(bias), _p(scale, scale_off),
                                                     _p(shift, shift_off), mod_bstride, seg_stride, batch, rows, D,
                                                     x_bstride, y_bstride, seg_split, float(eps), _stream()),
           "alg_layernorm_modulate_seg")
    return y


def headnorm_rope_(x, weight, cos, sin, x_rstride, x_bstride, batch, rows, heads, rope_tokens, eps, x_off=0):
    _check(load_library().alg_headnorm_rope(_p(x, x_off), _p(weight), _p(cos), _p(sin), x_rstride, x_bstride, batch, rows,
                                            heads, rope_tokens, float(eps), _stream()), "alg_headnorm_rope")
    return x


def masked_mean(x, valid, out, batch, L, D):
    _check(load_library().alg_masked_mean(_p(x), _p(valid), _p(out), batch, L, D, _stream()), "alg_masked_mean")
    return out


def silu(x, y):
    _check(load_library().alg_silu(_p(x), _p(y), x.numel(), _stream()), "alg_silu")
    return y


def rms_norm_rows(x, gamma, y, rows, C, Cp, silu=True):
    """WanRMS_norm (+ SiLU) over rows of Cp (padded) channels, statistics over the first C."""
    _check(load_library().alg_rms_norm_rows(_p(x), _p(gamma), _p(y), rows, C, Cp, int(silu), _stream()), "alg_rms_norm_rows")
    return y


def softmax_hilo(neg_hi, lo, p, rows, cols, ld, scale):
    _check(load_library().alg_softmax_hilo(_p(neg_hi), _p(lo), _p(p), rows, cols, ld, float(scale), _stream()),
           "alg_softmax_hilo")
    return p


CONV_PLAIN, CONV_PAIR, CONV_STRIDE2 = 0, 1, 2


def conv_cl(x, w, bias, res, y, frames, Hp, Wp, Cin, Cout, kt, pair=False, stride2=False, x_off=0, y_off=0, res_off=0):
    """alg_conv_cl_bf16 on flat bf16 buffers (offsets in elements); pair: w / bias in the two-voxel packing; stride2: the
    downsampler convolution (output rows at the input's pitch)."""
    mode = CONV_STRIDE2 if stride2 else (CONV_PAIR if pair else CONV_PLAIN)
    _check(load_library().alg_conv_cl_bf16(_p(x, x_off), _p(w), _p(bias), _p(res, res_off), _p(y, y_off), frames, Hp, Wp,
                                           Cin, Cout, kt, mode, _stream()), "alg_conv_cl_bf16")
    return y


def embed_rows(ids, table, out):
    _check(load_library().alg_embed_rows(_p(ids), _p(table), _p(out), ids.numel(), table.shape[1], table.shape[0],
                                         _stream()), "alg_embed_rows")
    return out


def t5_layernorm(x, weight, y, rows, D, eps):
    _check(load_library().alg_t5_layernorm(_p(x), _p(weight), _p(y), rows, D, eps, _stream()), "alg_t5_layernorm")
    return y


def attn_bias(qkv, out, bias_table, rel_bucket, key_mask, batch, heads, L, scale=1.0, head_dim=64, causal=False):
    """qkv: [batch*L, 3*heads*head_dim] fused projections (q | k | v); out: [batch*L, heads*head_dim]."""
    inner = heads * head_dim
    _check(load_library().alg_attn_bias(_p(qkv), _p(qkv, inner), _p(qkv, 2 * inner), _p(out), _p(bias_table),
                                        _p(rel_bucket), _p(key_mask), batch, heads, head_dim, L, 3 * inner, inner,
                                        float(scale), int(causal), _stream()), "alg_attn_bias")
    return out


def quick_gelu_(x):
    _check(load_library().alg_quick_gelu(_p(x), x.numel(), _stream()), "alg_quick_gelu")
    return x


def mul_bf16(a, b, out):
    _check(load_library().alg_mul_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "alg_mul_bf16")
    return out


def vae_group_norm(x, stats, gamma, beta, out, g, silu=True):
    _check(load_library().alg_vae_group_norm(_p(x), _p(stats), _p(gamma), _p(beta), _p(out), byref(g), int(silu),
                                             _stream()), "alg_vae_group_norm")
    return out


def vae_pad(x, out, frames, H, W, C):
    _check(load_library().alg_vae_pad(_p(x), _p(out), frames, H, W, C, _stream()), "alg_vae_pad")
    return out


def vae_repitch(x, out, frames, H, W, C, src_rows, src_wp):
    _check(load_library().alg_vae_repitch(_p(x), _p(out), frames, H, W, C, src_rows, src_wp, _stream()), "alg_vae_repitch")
    return out


def vae_unpack_planes(x, out, frames, H, W, C):
    _check(load_library().alg_vae_unpack_planes(_p(x), _p(out), frames, H, W, C, _stream()), "alg_vae_unpack_planes")
    return out


def pack_conv_pair(w, bias, taps_t):
    """[Cout][taps_t*9][Cin] / [Cout] -> the two-voxel packing of alg_conv_cl_bf16: [2*Cout][taps_t*12][Cin] / [2*Cout]."""
    co = w.shape[0]
    w = w.reshape(co, taps_t, 3, 3, -1)
    wp = w.new_zeros(2, co, taps_t, 3, 4, w.shape[-1])
    wp[0, :, :, :, 0:3] = w
    wp[1, :, :, :, 1:4] = w
    return wp.reshape(2 * co, -1).contiguous(), torch.cat([bias, bias]).contiguous()


def vae_geom(**kw):
    g = VaeGeom()
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def vae_groupnorm_workspace(g):
    n = load_library().alg_vae_groupnorm_workspace(byref(g))
    if n < 0:
        _check(1, "alg_vae_groupnorm_workspace")
    return n


def vae_groupnorm_stats(x, g, eps, workspace, stats):
    _check(load_library().alg_vae_groupnorm_stats(_p(x), byref(g), eps, _p(workspace), _p(stats), _stream()),
           "alg_vae_groupnorm_stats")
    return stats


def vae_spatial_norm(x, stats, gamma, beta, zyb, out, g, silu=True):
    _check(load_library().alg_vae_spatial_norm(_p(x), _p(stats), _p(gamma), _p(beta), _p(zyb), _p(out), byref(g),
                                               int(silu), _stream()), "alg_vae_spatial_norm")
    return out


def vae_upsample(x, out, frames_out, H, W, C, compress_time, first_single):
    _check(load_library().alg_vae_upsample(_p(x), _p(out), frames_out, H, W, C, int(compress_time), int(first_single),
                                           _stream()), "alg_vae_upsample")
    return out


def vae_pack_latent(z, c_stride, frame_stride, out, frames, h, w, channels, scale, z_off=0):
    _check(load_library().alg_vae_pack_latent(_p(z, z_off), c_stride, frame_stride, _p(out), frames, h, w, channels, scale,
                                              _stream()), "alg_vae_pack_latent")
    return out


def vae_unpack_video(x, out, frames, H, W, to_uint8=False):
    _check(load_library().alg_vae_unpack_video(_p(x), _p(out), frames, H, W, int(to_uint8), _stream()),
           "alg_vae_unpack_video")
    return out


def timestep_embedding_f32(t, out, n, dim):
    _check(load_library().alg_timestep_embedding_f32(_p(t), _p(out), n, dim, _stream()), "alg_timestep_embedding_f32")
    return out


def linear_f32(x, W, b, y, y_bf16, y_silu_bf16, M, N, K, act=0):
    _check(load_library().alg_linear_f32(_p(x), _p(W), _p(b), _p(y), _p(y_bf16), _p(y_silu_bf16), M, N, K, act, _stream()),
           "alg_linear_f32")


def gelu_erf_(x):
    _check(load_library().alg_gelu_erf(_p(x), x.numel(), _stream()), "alg_gelu_erf")
    return x


def unipc_update(x, m0, m1, m_new, r, c, k, rk=1.0, rho0=0.0, rho_new=0.0):
    """(r*x - c*m0) - k*(rho0*((m1-m0)/rk) + rho_new*(m_new-m0)) on fp32 tensors; m1 / m_new may be None."""
    lib = load_library()
    for t in (x, m0, m1, m_new):
        if t is None:
            continue
        _dev(t, "unipc term")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != x.shape:
            raise AlgHipError("unipc_update needs contiguous fp32 tensors of one shape")
    out = torch.empty_like(x)
    p = lambda t: None if t is None else _ptr(t)
    _check(lib.alg_unipc_update(p(x), p(m0), p(m1), p(m_new), _ptr(out), x.numel(), float(r), float(c), float(k),
                                float(rk), float(rho0), float(rho_new), _stream()), "alg_unipc_update")
    return out


def gemm_args(A, B, C, M, N, K, lda, ldb, ldc, bias=None, R=None, ldr=0, gate=None, batch=1, strideA=0, strideB=0,
              strideC=0, strideR=0, strideGate=0, seg_split=0, act=ACT_NONE, flags=0, a_off=0, b_off=0, c_off=0, r_off=0,
              gate_off=0, gate_seg_stride=None, bias_off=0, perm_col0=0, a_scale=None, b_scale=None, strideAScale=0,
              strideBScale=0, a_scale_off=0, b_scale_off=0):
    """The alg_gemm_args struct of one call (offsets in elements)."""
    args = GemmArgs()
    fp8 = a_scale is not None
    if isinstance(B, PackedB):
        if (B.N, B.K) != (N, K) or b_off or strideB:
            raise AlgHipError("packed B was built for N = %d, K = %d (shared by the batch, no offset)" % (B.N, B.K))
        flags |= GEMM_B_PACKED11
    args.A = A.data_ptr() + A.element_size() * a_off
    args.B = B.data_ptr() + B.element_size() * b_off
    if fp8:
        args.a_scale = a_scale.data_ptr() + 4 * a_scale_off
        args.b_scale = b_scale.data_ptr() + 4 * b_scale_off
        args.strideAScale, args.strideBScale = strideAScale, strideBScale
    args.C = C.data_ptr() + 2 * c_off
    args.bias = (bias.data_ptr() + 2 * bias_off) if bias is not None else None
    if gate_seg_stride is not None:
        flags |= GEMM_GATE_SEG_STRIDE
        args.gate_seg_stride = gate_seg_stride
    args.R = (R.data_ptr() + 2 * r_off) if R is not None else None
    args.gate = (gate.data_ptr() + gate.element_size() * gate_off) if gate is not None else None
    args.lda, args.ldb, args.ldc, args.ldr = lda, ldb, ldc, ldr
    args.strideA, args.strideB, args.strideC, args.strideR, args.strideGate = strideA, strideB, strideC, strideR, strideGate
    args.M, args.N, args.K, args.batch = M, N, K, batch
    args.seg_split, args.act, args.flags = seg_split, act, flags
    args.perm_col0 = perm_col0
    return args, fp8


class PackedB:
    """An nn.Linear weight [N, K] (bf16) re-ordered for GEMM schedule 11 (alg_pack_b_p11): pass it as `B` of `gemm` and the call
    sets ALG_GEMM_B_PACKED11 itself.  Packed once when a model is loaded; the values are the weight's, only their order differs."""

    def __init__(self, w):
        lib = load_library()
        _dev(w, "weight")
        if w.dtype != torch.bfloat16 or w.dim() != 2 or w.stride(1) != 1:
            raise AlgHipError("PackedB takes a 2-D bf16 weight with unit column stride")
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        nbytes = int(lib.alg_pack_b_p11_bytes(self.N, self.K))
        if nbytes <= 0:
            raise AlgHipError("PackedB: K = %d must be a positive multiple of 64" % self.K)
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _check(lib.alg_pack_b_p11(_ptr(w), _ptr(self.data), self.N, self.K, int(w.stride(0)), _stream()), "alg_pack_b_p11")

    def data_ptr(self):
        return self.data.data_ptr()

    def element_size(self):
        return 2


def gemm(*a, **kw):
    """Raw pointer-level GEMM: C = R + gate * act(A @ B^T + bias); offsets in elements.  With a_scale / b_scale (float32
    row scales) A and B are OCP e4m3 bytes and the call goes to alg_gemm_fp8."""
    lib = load_library()
    args, fp8 = gemm_args(*a, **kw)
    if fp8:
        _check(lib.alg_gemm_fp8(ctypes.byref(args), _stream()), "alg_gemm_fp8")
    else:
        _check(lib.alg_gemm_bf16(ctypes.byref(args), _stream()), "alg_gemm_bf16")


def gemm_pair(first, second):
    """Two independent plain bf16 GEMMs -- each a (args, kwargs) pair of `gemm` -- as one persistent launch
    (alg_gemm_bf16_pair); bit-identical to the two separate calls."""
    lib = load_library()
    a, fa = gemm_args(*first[0], **first[1])
    b, fb = gemm_args(*second[0], **second[1])
    if fa or fb:
        raise AlgHipError("gemm_pair takes bf16 problems")
    _check(lib.alg_gemm_bf16_pair(ctypes.byref(a), ctypes.byref(b), _stream()), "alg_gemm_bf16_pair")


def gemm_pair_qk(first, second, wq, bq, wk, bk, cos, sin, heads, text_len, eps, q_scale=1.0):
    """`gemm_pair` whose FIRST problem is the CogVideoX Q|K projection, with the per-head LayerNorm + rotary embedding of
    `qk_norm_rope_` applied in its store loop (alg_gemm_bf16_pair_qk); bit-identical to gemm_pair + qk_norm_rope_."""
    lib = load_library()
    a, fa = gemm_args(*first[0], **first[1])
    b, fb = gemm_args(*second[0], **second[1])
    if fa or fb:
        raise AlgHipError("gemm_pair_qk takes bf16 problems")
    ptrs = [t.data_ptr() if t is not None else None for t in (wq, bq, wk, bk, cos, sin)]
    e = QkNormRopeArgs(*ptrs, heads, text_len, float(eps), float(q_scale))
    _check(lib.alg_gemm_bf16_pair_qk(ctypes.byref(a), ctypes.byref(b), ctypes.byref(e), _stream()), "alg_gemm_bf16_pair_qk")


def quantize_fp8_rows(x, q, scale, rows, K, x_rstride=None, x_off=0, q_off=0, scale_off=0):
    """Row-wise OCP e4m3 quantisation: q (uint8 / float8 bytes), scale (float32) filled in place."""
    x_rstride = K if x_rstride is None else x_rstride
    _check(load_library().alg_quantize_fp8_rows(_p(x, x_off), x_rstride, _p(q, q_off), _p(scale, scale_off), rows, K,
                                                _stream()), "alg_quantize_fp8_rows")
    return q, scale


ATTN_Q_PRESCALED = 1


def flash_attn_d64(q, k, vt, o, batch, heads, S, q_bstride, q_rstride, vt_bstride, vt_rstride, o_bstride, o_rstride,
                   scale, q_off=0, k_off=0, q_prescaled=False):
    """q_prescaled: q already carries scale * log2(e) (qk_norm_rope_ with q_scale): alg_flash_attn_d64_ex."""
    lib = load_library()
    flags = ATTN_Q_PRESCALED if q_prescaled else 0
    wsb = int(lib.alg_flash_attn_d64_workspace_bytes(batch, heads, S, flags))     # split-KV tail partials (0: one launch)
    ws = scratch(o, wsb, "attn_d64")
    _check(lib.alg_flash_attn_d64_ex(c_void_p(q.data_ptr() + 2 * q_off), c_void_p(k.data_ptr() + 2 * k_off), _ptr(vt),
                                     _ptr(o), batch, heads, S, q_bstride, q_rstride, vt_bstride, vt_rstride, o_bstride,
                                     o_rstride, float(scale), flags, _ptr(ws), wsb, _stream()),
           "alg_flash_attn_d64")


def layernorm_modulate(x, y, weight, bias, scale, shift, mod_bstride, batch, rows, D, seg_split, eps,
                       x_bstride=None, y_bstride=None, x_off=0, y_off=0, scale_off=0, shift_off=0):
    """Pointer-level LayerNorm(+modulate); *_off are element offsets into the given tensors."""
    lib = load_library()
    x_bstride = rows * D if x_bstride is None else x_bstride
    y_bstride = rows * D if y_bstride is None else y_bstride
    sc = c_void_p(scale.data_ptr() + 2 * scale_off) if scale is not None else c_void_p(0)
    sh = c_void_p(shift.data_ptr() + 2 * shift_off) if shift is not None else c_void_p(0)
    _check(lib.alg_layernorm_modulate(c_void_p(x.data_ptr() + 2 * x_off), c_void_p(y.data_ptr() + 2 * y_off),
                                      _ptr(weight), _ptr(bias), sc, sh, mod_bstride, batch, rows, D, x_bstride,
                                      y_bstride, seg_split, float(eps), _stream()), "alg_layernorm_modulate")


def qk_norm_rope_(qk, wq, bq, wk, bk, cos, sin, batch, S, heads, text_len, eps, q_scale=1.0):
    lib = load_library()
    _check(lib.alg_qk_norm_rope_scaled(_ptr(qk), _ptr(wq), _ptr(bq), _ptr(wk), _ptr(bk), _ptr(cos), _ptr(sin), batch, S,
                                       heads, text_len, float(eps), float(q_scale), _stream()), "alg_qk_norm_rope")


def patchify(latents, lat_bstride, conds, out, n_samples, frames, C, H, W, p, p_t=1):
    lib = load_library()
    arr = (c_void_p * n_samples)(*[c.data_ptr() for c in conds])
    _check(lib.alg_patchify_t(_ptr(latents), lat_bstride, arr, _ptr(out), n_samples, frames, C, H, W, p, p_t, _stream()),
           "alg_patchify")


def unpatchify(x, out, n_samples, frames, C, H, W, p, p_t=1):
    lib = load_library()
    _check(lib.alg_unpatchify_t(_ptr(x), _ptr(out), n_samples, frames, C, H, W, p, p_t, _stream()), "alg_unpatchify")


def timestep_embedding(t, out, n, dim, flip_sin_to_cos=True):
    lib = load_library()
    _check(lib.alg_timestep_embedding(_ptr(t), _ptr(out), n, dim, 1 if flip_sin_to_cos else 0, _stream()),
           "alg_timestep_embedding")
