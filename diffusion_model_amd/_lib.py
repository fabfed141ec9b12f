"""ctypes binding of ``libdm_hip.so`` (the C ABI declared in ``include/dm_hip.h``).

The library is built in-tree by ``build.py`` (``hipcc --offload-arch=gfx950``).  There is no
fallback: if it is missing or a call fails, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DM_HIP_LIBRARY=<path>: load another build of the same ABI (the diagnostic timeline build of `python build.py --timeline`)
LIB_PATH = os.environ.get("DM_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libdm_hip.so")

DM_F32, DM_BF16 = 0, 1
EPI_PLAIN, EPI_GEGLU, EPI_NCHW_F32, EPI_QUICKGELU = 0, 1, 2, 3
TILE_AUTO, TILE_128x128, TILE_64x128, TILE_64x64, TILE_256x128, TILE_32x64 = 0, 1, 2, 3, 4, 5
(TILE_D128x128_S2, TILE_D128x128_S3, TILE_D256x128_S3, TILE_D64x128_S3, TILE_D64x128_S4, TILE_D64x64_S4,
 TILE_D32x64_S4, TILE_D256x64_S3, TILE_D128x64_S4) = range(11, 20)
TILE_DIMS = {TILE_128x128: (128, 128), TILE_64x128: (64, 128), TILE_64x64: (64, 64),
             TILE_256x128: (256, 128), TILE_32x64: (32, 64),
             TILE_D128x128_S2: (128, 128), TILE_D128x128_S3: (128, 128), TILE_D256x128_S3: (256, 128),
             TILE_D64x128_S3: (64, 128), TILE_D64x128_S4: (64, 128), TILE_D64x64_S4: (64, 64),
             TILE_D32x64_S4: (32, 64), TILE_D256x64_S3: (256, 64), TILE_D128x64_S4: (128, 64)}
TILE_D256x64_S2, TILE_D128x64_S2, TILE_D64x64_S2 = 20, 21, 22
TILE_D256x256_S2 = 23
TILE_E128x128, TILE_E128x64, TILE_E64x64, TILE_E64x128 = 24, 25, 26, 27  # early-refill loop, 2 LDS buffers
TILE_DIMS.update({TILE_E128x128: (128, 128), TILE_E128x64: (128, 64), TILE_E64x64: (64, 64), TILE_E64x128: (64, 128)})
TILE_DIMS.update({TILE_D256x64_S2: (256, 64), TILE_D128x64_S2: (128, 64), TILE_D64x64_S2: (64, 64),
                  TILE_D256x256_S2: (256, 256)})
# two k-tiles per barrier (four stages as two halves)
TILE_U64x64 = 28
PAIR_TILES = (TILE_U64x64,)
TILE_DIMS.update({TILE_U64x64: (64, 64)})
ALL_TILES = tuple(TILE_DIMS)  # tiles every dense / conv problem may use
# bf16 stride-1 3x3 convolutions with the chunk's activation (and its border) resident in LDS; own weight order
TILE_H128x128 = 29
TILE_DIMS.update({TILE_H128x128: (128, 128)})
# row-panel tiles (bf16, dense, K = 320 / 640): a block owns `bm` rows and all column tiles; DmGemmDesc.splitk counts
# column slices for them
TILE_P64, TILE_P128 = 31, 32
PANEL_TILES = {TILE_P64: 64, TILE_P128: 128}
TILE_DIMS.update({TILE_P64: (64, 64), TILE_P128: (128, 64)})

_ERR = {1: "invalid argument / unsupported shape", 2: "HIP launch error", 3: "misaligned pointer or stride",
        4: "unsupported attention head dimension (kernels exist for 40, 64, 80 and 160; wider heads go through the GEMM route)",
        5: "problem too large for the fixed scratch buffer of this allocation-free entry point"}


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("lda", C.c_long),
        ("w", C.c_void_p),
        ("c", C.c_void_p), ("ldc", C.c_long),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("ld_rowbias", C.c_long), ("rows_per_batch", C.c_int),
        ("r", C.c_void_p), ("ldr", C.c_long),
        ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
        ("npad", C.c_int), ("kpad", C.c_int),
        ("dtype", C.c_int),
        ("conv", C.c_int),
        ("hin", C.c_int), ("win", C.c_int), ("hout", C.c_int), ("wout", C.c_int),
        ("cin", C.c_int), ("stride", C.c_int), ("up", C.c_int),
        ("epi", C.c_int),
        ("tile", C.c_int),
        ("splitk", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_long),
        ("stats", C.c_void_p), ("ld_stats", C.c_long), ("stats_cpg", C.c_int), ("stats_ch0", C.c_int),
        ("stats2", C.c_void_p), ("ld_stats2", C.c_long), ("stats2_cpg", C.c_int), ("stats2_ch0", C.c_int),
        ("ln_in", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_out", C.c_void_p),
        ("a2", C.c_void_p), ("lda2", C.c_long), ("cin2", C.c_int),
        ("pad_hi_only", C.c_int),
        ("gn_sums", C.c_void_p), ("ld_gn", C.c_long), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),
        ("gn_eps", C.c_float), ("gn_silu", C.c_int), ("gn_groups", C.c_int),
        ("w_split", C.c_int),
        ("conv_chunk_major", C.c_int),
    ]


class FfnDesc(C.Structure):
    """DmFfnDesc: the feed-forward half of an AttentionBlock as one launch (dm_ffn_geglu)."""
    _fields_ = [
        ("a", C.c_void_p), ("lda", C.c_long),
        ("w1", C.c_void_p), ("bias1", C.c_void_p),
        ("ln_in", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("w2", C.c_void_p), ("bias2", C.c_void_p), ("npad2", C.c_int),
        ("r", C.c_void_p), ("ldr", C.c_long),
        ("c", C.c_void_p), ("ldc", C.c_long),
        ("m", C.c_int), ("c_model", C.c_int), ("dtype", C.c_int),
        ("rows_per_batch", C.c_int),
        ("stats", C.c_void_p), ("ld_stats", C.c_long), ("stats_cpg", C.c_int), ("stats_ch0", C.c_int),
        ("stats2", C.c_void_p), ("ld_stats2", C.c_long), ("stats2_cpg", C.c_int), ("stats2_ch0", C.c_int),
    ]


_vp, _l, _i, _f = C.c_void_p, C.c_long, C.c_int, C.c_float
_PROTOS = {
    "dm_abi_version": (C.c_int, []),
    "dm_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "dm_gemm_plan": (None, [_i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "dm_gemm_workspace_bytes": (C.c_long, [_i, _i, _i]),
    "dm_gemm_streamk_workspace_bytes": (C.c_long, [_i, _i, _i, _i]),
    "dm_conv_halo_supported": (C.c_int, [_i, _i, _i, _i, C.c_long]),
    "dm_gemm_stats_supported": (C.c_int, [_i, _i, _i]),
    "dm_gemm_panel_supported": (C.c_int, [C.POINTER(GemmDesc), _i, _i]),
    "dm_gemm_gn_supported": (C.c_int, [_i, _i, _i, _i, _i, _i, _i]),
    "dm_ffn_geglu": (C.c_int, [C.POINTER(FfnDesc), _vp]),
    "dm_ffn_geglu_supported": (C.c_int, [_i, _i, _i, _i, _i]),
    "dm_groupnorm_apply_sums": (C.c_int, [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _f, _i, _vp]),
    "dm_zero": (C.c_int, [_vp, _l, _vp]),
    "dm_pack_weight_order": (C.c_int, [_i, _i, _i, _vp]),
    "dm_pack_rows": (C.c_int, [_vp, _l, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "dm_softmax_rows": (C.c_int, [_vp, _l, _i, _i, _f, _i, _vp]),
    "dm_vae_sample": (C.c_int, [_vp, _vp, _vp, _i, _l, _f, _vp]),
    "dm_groupnorm_stats": (C.c_int, [_vp, _l, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "dm_groupnorm_apply": (C.c_int, [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f, _i, _vp]),
    "dm_groupnorm_default_nsplit": (C.c_int, [_i, _i]),
    "dm_init": (C.c_int, []),
    "dm_groupnorm_noalloc": (C.c_int, [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp]),
    "dm_layernorm": (C.c_int, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _f, _vp]),
    "dm_attention": (C.c_int, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dm_attention_split_f32": (C.c_int, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "dm_embed_tokens": (C.c_int, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp]),
    "dm_rescale": (C.c_int, [_vp, _vp, _l, _i, _f, _f, _f, _f, _i, _f, _f, _vp]),
    "dm_u8_to_images": (C.c_int, [_vp, _i, _i, _l, _f, _f, _f, _f, _vp, _i, _vp]),
    "dm_images_to_u8": (C.c_int, [_vp, _i, _i, _i, _l, _f, _f, _f, _f, _f, _f, _vp, _vp]),
    "dm_linear_rows": (C.c_int, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "dm_timestep_embedding": (C.c_int, [_vp, _i, _i, _f, _vp, _vp]),
    "dm_ddpm_step": (C.c_int, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp]),
    "dm_ddim_step": (C.c_int, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp]),
    "dm_ksampler_input": (C.c_int, [_vp, _vp, _l, _i, _vp, _vp, _vp]),
    "dm_ksampler_step": (C.c_int, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _f, _i, _vp]),
    "dm_klms_step": (C.c_int, [_vp, _vp, _vp, _l, _vp, _vp, _i, _i, _vp, _f, _i, _vp]),
    "dm_scale_add": (C.c_int, [_vp, _vp, _vp, _i, _l, _vp, _vp, _vp, _f, _vp]),
    "dm_axpbypcz": (C.c_int, [_vp, _f, _vp, _f, _vp, _f, _vp, _l, _vp]),
    "dm_nchw_to_rows": (C.c_int, [_vp, _i, _i, _i, _vp, _l, _i, _i, _vp]),
    "dm_rows_to_nchw": (C.c_int, [_vp, _l, _i, _i, _i, _vp, _i, _vp]),
    "dm_convert": (C.c_int, [_vp, _i, _vp, _i, _l, _vp]),
    "dm_gather_row": (C.c_int, [_vp, _l, _vp, _vp, _l, _i, _i, _vp]),
    "dm_advance_counter": (C.c_int, [_vp, _i, _vp]),
    "dm_graph_begin": (C.c_int, [_vp]),
    "dm_graph_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dm_graph_launch": (C.c_int, [_vp, _vp]),
    "dm_graph_destroy": (C.c_int, [_vp]),
    "dm_event_create": (C.c_int, [C.POINTER(_vp)]),
    "dm_event_record": (C.c_int, [_vp, _vp]),
    "dm_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(_f)]),
    "dm_event_destroy": (C.c_int, [_vp]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the HIP extension; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python build.py` (needs hipcc); "
                "diffusion_model_amd has no CPU or PyTorch fallback.")
        # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already mapped when this
        # library's dependency on libamdhip64 is resolved, or the process ends up with the system runtime serving
        # this library and a device it cannot see (hipErrorNoDevice at the first launch).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.dm_abi_version() != 13:
            raise RuntimeError("libdm_hip.so ABI version mismatch; rebuild with `python build.py`")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libdm_hip {what} failed: {_ERR.get(rc, rc)}")
