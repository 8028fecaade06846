"""ctypes binding of libskyemb.so (the C ABI declared in include/skyemb.h).

The product path has NO fallback: if the HIP library is missing or fails to load,
every op raises ``SkyembLibraryError``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE
# libskyemb.so is dlopen'ed so that both resolve to ONE runtime (otherwise kernels launched on
# torch's streams fail with "no ROCm-capable device is detected").
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SKYEMB_LIB") or os.path.join(_HERE, "libskyemb.so")   # SKYEMB_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")

BF16, F32, F16 = 0, 1, 2
ABI_VERSION = 110          # skyemb_version() of the library this binding was written against (csrc/api.cpp)
KC, RC = 0, 1
ACT_NONE, ACT_GELU, ACT_DGELU = 0, 1, 2


class SkyembLibraryError(RuntimeError):
    pass


class SkyembError(RuntimeError):
    pass


c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", c_vp), ("B", c_vp), ("lda", c_i64), ("ldb", c_i64), ("a_layout", c_i32), ("b_layout", c_i32),
        ("M", c_i32), ("N", c_i32), ("K", c_i32), ("dtype", c_i32), ("alpha", c_f32),
        ("bias", c_vp), ("table", c_vp), ("tab_row", c_vp), ("ldt", c_i64), ("dst_row", c_vp),
        ("resid", c_vp), ("ldr", c_i64), ("aux", c_vp), ("ldaux", c_i64), ("act", c_i32),
        ("out_f32", c_vp), ("ldo32", c_i64), ("out", c_vp), ("ldo", c_i64), ("out2", c_vp), ("ldo2", c_i64),
        ("tile", c_i32), ("colsum_a", c_vp), ("ws", c_vp), ("ws_bytes", c_i64), ("split_k", c_i32), ("prefetch_wgs", c_i32),
        ("prefetch", c_vp), ("prefetch_bytes", c_i64),
    ]


class AdamwDesc(ctypes.Structure):
    """skyemb_adamw_desc (include/skyemb.h): the optimiser step fused into a grouped weight-gradient launch."""
    _fields_ = [("g_base", c_vp), ("p", c_vp), ("m", c_vp), ("v", c_vp), ("p_lp", c_vp), ("hyper", c_vp), ("n_decay", c_i64),
                ("beta1", c_f32), ("beta2", c_f32), ("eps", c_f32), ("weight_decay", c_f32), ("grad_scale", c_f32), ("enabled", c_i32)]


class LnBwdSide(ctypes.Structure):
    """skyemb_ln_bwd_side (include/skyemb.h): a LayerNorm backward riding in a grouped weight-gradient launch."""
    _fields_ = [("dy", c_vp), ("x", c_vp), ("gamma", c_vp), ("mean", c_vp), ("rstd", c_vp), ("g_in", c_vp), ("g_out", c_vp), ("g_lp", c_vp),
                ("part", c_vp), ("M", c_i32), ("D", c_i32)]


class GemmGroupInfo(ctypes.Structure):
    _fields_ = [("total_blocks", c_i32), ("tile", c_i32), ("class_mask", c_i32), ("reserved", c_i32)]


# name -> (restype, argtypes); must list every symbol include/skyemb.h declares (tests check this)
PROTOTYPES = {
    "skyemb_last_error": (ctypes.c_char_p, []),
    "skyemb_version": (c_i32, []),
    "skyemb_debug_skip": (c_i32, [c_i32]),
    "skyemb_gemm_launch_counts": (c_i32, [c_vp, c_i32, c_i32]),
    "skyemb_gemm": (c_i32, [ctypes.POINTER(GemmArgs), c_vp]),
    "skyemb_gemm_group_blob_bytes": (c_i64, [c_i32]),
    "skyemb_gemm_group_plan": (c_i32, [ctypes.POINTER(GemmArgs), c_i32, c_i32, c_vp, c_i64, ctypes.POINTER(GemmGroupInfo)]),
    "skyemb_set_scalars": (c_i32, [c_vp, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "skyemb_gemm_group_plan_adamw": (c_i32, [ctypes.POINTER(GemmArgs), c_i32, c_i32, ctypes.POINTER(AdamwDesc), c_vp, c_i64,
                                             ctypes.POINTER(GemmGroupInfo)]),
    "skyemb_gemm_group_plan_side_adamw": (c_i32, [ctypes.POINTER(GemmArgs), c_i32, c_i32, ctypes.POINTER(AdamwDesc), c_i32, c_i64, c_i64, c_i32,
                                                  c_vp, c_i64, ctypes.POINTER(GemmGroupInfo)]),
    "skyemb_gemm_group_attach_ln_bwd": (c_i32, [c_vp, c_i64, ctypes.POINTER(GemmGroupInfo), ctypes.POINTER(LnBwdSide)]),
    "skyemb_gemm_group_launch": (c_i32, [c_vp, ctypes.POINTER(GemmGroupInfo), c_vp]),
    "skyemb_colsum": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "skyemb_random_mask_from_noise": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "skyemb_simmim_mask_from_noise": (c_i32, [c_vp, c_vp, ctypes.c_double, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "skyemb_patch_gather": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32,
                                    c_f32, c_vp]),
    "skyemb_patch_gather_bwd_pmv": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                            c_vp]),
    "skyemb_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp]),
    "skyemb_patch_gather_blend": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp]),
    "skyemb_patch_gather_bwd_pmv_blend": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_radec_token_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "skyemb_radec_token_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "skyemb_simmim_pixel_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                         c_f32, c_f32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "skyemb_gather_rows_host": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i32]),
    "skyemb_h5_unchunk_host": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32]),
    "skyemb_fits_decode_tiles_host": (c_i32, [c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i32]),
    "skyemb_fits_dequantise_tiles_host": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                                  c_vp, c_i64, c_i64, c_i32, c_i32]),
    "skyemb_augment": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_attnpool_q": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "skyemb_attnpool_fwd": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_attnpool_bwd": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_attnpool_q_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "skyemb_tile_cutouts": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_i32, c_i32, c_vp, c_vp]),
    "skyemb_clip_crop": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_i32, c_vp]),
    "skyemb_layernorm_bwd_blocks": (c_i32, [c_i32]),
    "skyemb_layernorm_bwd_reduce_batch": (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    "skyemb_layernorm_bwd": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_i32, c_i32, c_vp]),
    "skyemb_mha_fwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_mha_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_fill_mask_tokens": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_gather_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "skyemb_rowsum_select": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "skyemb_masked_patch_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32,
                                         c_i32, c_i32, c_f32, c_f32, c_i32, c_i32, c_f32, c_vp]),
    "skyemb_adamw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_f32, c_f32, c_f32, c_f32,
                             c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_vp]),
    "skyemb_cast": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_vp]),
    "skyemb_standardise": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "skyemb_weighted_norms": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "skyemb_cosine_topk_chunks": (c_i32, [c_i64, c_i32, c_i32, c_i32]),
    "skyemb_cosine_topk": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_i64, c_i32, c_vp, c_vp,
                                   c_vp, c_vp]),
    "skyemb_kth_largest_floor": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "skyemb_cosine_sample_floor": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "skyemb_cosine_sample_floor_applicable": (c_i32, [c_i32, c_i64, c_i32, c_i32]),
    "skyemb_topk_merge": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "skyemb_topk_prefilter_applicable": (c_i32, [c_i32, c_i64, c_i32, c_i32]),
    "skyemb_topk_prefilter_ws_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "skyemb_bank16_bytes": (c_i64, [c_i64, c_i32]),
    "skyemb_bank16_rowp_rows": (c_i64, [c_i64]),
    "skyemb_bank16_prepare": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "skyemb_cosine_topk_prefiltered": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i64, c_vp,
                                               c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "skyemb_cosine_scores": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_f32, c_vp, c_vp]),
}

_LIB = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into sky_embeddings_amd/libskyemb.so (hipcc, no GPU needed)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return SO_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise SkyembLibraryError(
                f"{SO_PATH} is missing: the HIP hot-path library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C sky_embeddings_amd/csrc`). "
                "There is no CPU fallback.")
        try:
            L = ctypes.CDLL(SO_PATH)
        except OSError as e:  # pragma: no cover
            raise SkyembLibraryError(f"cannot load {SO_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise SkyembLibraryError(f"{SO_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        if L.skyemb_version() != ABI_VERSION:    # struct layouts / argument lists changed between versions: never call across them
            raise SkyembLibraryError(f"{SO_PATH} reports ABI version {L.skyemb_version()}, this binding is written against "
                                     f"{ABI_VERSION}: rebuild the library (make -C sky_embeddings_amd/csrc)")
        _LIB = L
    return _LIB


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().skyemb_last_error()
        raise SkyembError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
