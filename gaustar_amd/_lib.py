"""ctypes binding of the C-ABI HIP library (include/gsr.h -> gaustar_amd/libgsr_hip.so).

There is deliberately NO fallback: if the shared library is missing or fails to load, every
entry point raises.  The product path never touches oracle/.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB_PATH", os.path.join(_HERE, "libgsr_hip.so"))   # override: experiment variants

ABI_VERSION = 16
ALLOC_FN = ctypes.CFUNCTYPE(c_void_p, c_void_p, c_size_t)

# name -> (restype, argtypes); mirrors include/gsr.h one to one (tests check both directions).
SIGNATURES = {
    "gsr_abi_version": (c_int, []),
    "gsr_last_error": (c_char_p, []),
    "gsr_geom_bytes": (c_size_t, [c_int]),
    "gsr_image_bytes": (c_size_t, [c_int, c_int]),
    "gsr_binning_bytes": (c_size_t, [c_int, c_int]),
    "gsr_binning_bytes_mt": (c_size_t, [c_int, c_int, c_int]),
    "gsr_grad_scratch_bytes": (c_size_t, [c_int]),
    "gsr_forward_stage1": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                   c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                   c_void_p]),
    "gsr_forward_stage2": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "gsr_forward_stage2_mt": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_forward_fused": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                  POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p]),
    "gsr_plan_bytes": (c_size_t, [c_int, c_int]),
    "gsr_plan_info_new": (POINTER(c_int), []),
    "gsr_plan_info_free": (None, [POINTER(c_int)]),
    "gsr_forward_planned": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                    POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p, POINTER(c_int),
                                    POINTER(c_int), c_void_p]),
    "gsr_camera_key": (c_int, [c_void_p, c_longlong, c_longlong, POINTER(ctypes.c_ulonglong)]),
    "gsr_camera_key_begin": (c_int, [c_void_p, c_longlong, c_longlong]),
    "gsr_camera_key_end": (c_int, [POINTER(ctypes.c_ulonglong)]),
    "gsr_release_stream_state": (c_int, [c_void_p]),
    "gsr_forward": (c_int, [ALLOC_FN, ALLOC_FN, ALLOC_FN, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, POINTER(c_int),
                            c_void_p]),
    "gsr_backward": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_backward_mt": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gsr_mark_visible": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_debug_export": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_debug_export_masks": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gsr_sh_to_rgb": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsr_sh_to_rgb_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "gsr_sh_to_rgbd": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gsr_sh_to_rgbd_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "gsr_sh_colors_split": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "gsr_sh_colors_split_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gsr_mesh_gaussians": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                   c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gsr_mesh_gaussians_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gsr_l1_ssim_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gsr_l1_ssim": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_longlong,
                            c_longlong, c_longlong, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong,
                            c_longlong, c_void_p]),
    "gsr_depth_l1_workspace_bytes": (c_size_t, []),
    "gsr_depth_l1": (c_int, [c_int, c_int, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_float,
                             c_float, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p]),
    "gsr_l1_ssim_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_longlong,
                                     c_longlong, c_longlong, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong,
                                     c_longlong, c_void_p]),
    "gsr_depth_l1_backward": (c_int, [c_int, c_int, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_float,
                                      c_float, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p]),
    "gsr_rgb_depth_loss": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_longlong,
                                   c_longlong, c_longlong, c_float, c_void_p, c_int, c_int, c_void_p, c_longlong, c_longlong,
                                   c_void_p, c_longlong, c_longlong, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "gsr_rgb_depth_loss_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_longlong,
                                            c_longlong, c_longlong, c_float, c_void_p, c_int, c_int, c_void_p, c_longlong, c_longlong,
                                            c_void_p, c_longlong, c_longlong, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                            c_longlong, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p]),
    "gsr_adam_step": (c_int, [c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int,
                              c_void_p]),
    "gsr_adam_step_multi": (c_int, [c_int, POINTER(c_longlong), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                    POINTER(c_void_p), POINTER(c_double), c_double, c_double, c_double, c_int, c_void_p]),
    "gsr_debug_set_bwd_order": (c_int, [c_void_p]),
    "gsr_debug_preprocess_occupancy": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "gsr_debug_set_trace": (c_int, [c_void_p]),
    "gsr_num_stages": (c_int, []),
    "gsr_stage_name": (c_char_p, [c_int]),
    "gsr_profile_enable": (c_int, [c_int]),
    "gsr_profile_read": (c_int, [POINTER(c_float), POINTER(c_int), c_int]),
    "gsr_debug_host_wait": (c_int, [POINTER(c_longlong), POINTER(c_longlong), c_int]),
}

_lib = None


class GsrError(RuntimeError):
    """A C-ABI call returned non-zero (message from gsr_last_error)."""


def load():
    """Load libgsr_hip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"gaustar_amd: HIP extension not found at {LIB_PATH}. Build it with "
            "`python -m gaustar_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.gsr_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"gaustar_amd: libgsr_hip.so has ABI {v}, Python side expects {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gsr_last_error()
        raise GsrError(f"{what} failed: {msg.decode() if msg else rc}")
