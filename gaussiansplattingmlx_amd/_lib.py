"""ctypes binding of libgsplat_hip.so (include/gsplat.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU
is usable, loading / context creation raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPLAT_LIB", os.path.join(HERE, "libgsplat_hip.so"))   # override: kernel experiments

GS_OK = 0
STATUS = {1: "GS_ERR_INVALID_ARG", 2: "GS_ERR_SIZE_MISMATCH", 3: "GS_ERR_WORKSPACE_OVERFLOW", 4: "GS_ERR_HIP",
          5: "GS_ERR_NO_FORWARD", 6: "GS_ERR_NO_DEVICE", 7: "GS_ERR_IO", 8: "GS_ERR_COMM", 9: "GS_ERR_REPLICA_MISMATCH"}
GS_ERR_REPLICA_MISMATCH = 9


class GsplatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class gs_camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_center", C.c_float * 3),
                ("fov_x", C.c_float), ("fov_y", C.c_float), ("focal_x", C.c_float), ("focal_y", C.c_float)]


class gs_dp_step_args(C.Structure):
    _fields_ = [("cot_color", C.c_void_p), ("cot_depth", C.c_void_p), ("cot_alpha", C.c_void_p),
                ("params_base", C.c_void_p), ("grads_base", C.c_void_p), ("m_base", C.c_void_p), ("v_base", C.c_void_p),
                ("n_arena", C.c_longlong), ("geom_numel", C.c_longlong), ("nseg", C.c_int),
                ("seg_end", C.c_longlong * 8), ("seg_lr", C.c_float * 8),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("cam_centers", C.c_void_p), ("color_cot_local", C.c_void_p), ("color_cot_all", C.c_void_p)]


GS_DP_ALLREDUCE, GS_DP_SH_COMPRESSED = 0, 1
GS_DP_UNIQUE_ID_BYTES = 128

_vp = C.c_void_p
_SIGS = {
    "gs_dp_unique_id": (C.c_int, [_vp]),
    "gs_dp_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "gs_dp_attach": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "gs_dp_shutdown": (C.c_int, [_vp]),
    "gs_dp_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gs_dp_step": (C.c_int, [_vp, C.c_int, C.POINTER(gs_dp_step_args)]),
    "gs_dp_allreduce_sum": (C.c_int, [_vp, _vp, C.c_longlong]),
    "gs_dp_check_overflow": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "gs_dp_exchange_timing": (C.c_int, [_vp, C.c_int]),
    "gs_dp_exchange_read": (C.c_int, [_vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gs_abi_version": (C.c_int, []),
    "gs_ctx_create": (C.c_int, [C.c_int] * 7 + [C.POINTER(_vp)]),
    "gs_ctx_destroy": (C.c_int, [_vp]),
    "gs_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "gs_ctx_reserve": (C.c_int, [_vp, C.c_int, C.c_longlong]),
    "gs_workspace_bytes": (C.c_size_t, [_vp]),
    "gs_sync": (C.c_int, [_vp]),
    "gs_overflow_pending": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "gs_wait": (C.c_int, [_vp]),
    "gs_last_error": (C.c_char_p, [_vp]),
    "gs_projection_forward": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 4 + [C.POINTER(gs_camera)] + [_vp] * 8),
    "gs_projection_backward": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 4 + [C.POINTER(gs_camera)] + [_vp] * 10),
    "gs_tile_bin": (C.c_int, [_vp, C.c_int] + [_vp] * 4),
    "gs_tile_bin_cut": (C.c_int, [_vp, C.c_int] + [_vp] * 5),
    "gs_tile_bin_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gs_tile_bin_views": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "gs_tile_bin_export": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gs_build_packed_tile_indices": (C.c_int, [_vp, C.c_uint32, _vp]),
    "gs_pack_gaussians": (C.c_int, [_vp, C.c_int] + [_vp] * 6),
    "gs_blend_forward": (C.c_int, [_vp, C.c_int] + [_vp] * 5),
    "gs_blend_backward": (C.c_int, [_vp, C.c_int] + [_vp] * 9),
    "gs_ssim_window": (C.c_int, [C.c_int, C.c_float, _vp]),
    "gs_ssim_forward": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 9),
    "gs_ssim_backward": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 11),
    "gs_render_forward": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 6 + [C.POINTER(gs_camera)] + [_vp] * 4),
    "gs_render_backward": (C.c_int, [_vp] + [_vp] * 9),
    "gs_render_backward_dp": (C.c_int, [_vp] + [_vp] * 8),
    "gs_render_backward_adam": (C.c_int, [_vp] * 7 + [C.c_longlong, _vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "gs_render_backward_dp_begin": (C.c_int, [_vp] * 5),
    "gs_render_backward_dp_finish": (C.c_int, [_vp] * 5),
    "gs_render_backward_dp_finish_geom": (C.c_int, [_vp] * 6),
    "gs_render_backward_dp_geom": (C.c_int, [_vp] * 10),
    "gs_sh_grad_from_views_adam_dir": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 9 + [C.c_longlong] + [C.c_float] * 6 + [_vp]),
    "gs_adam_step_add": (C.c_int, [_vp, C.c_longlong] + [_vp] * 4 + [C.c_int, _vp, _vp] + [C.c_float] * 4 + [_vp, C.c_longlong]),
    "gs_sh_grad_from_views": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 5),
    "gs_accum_grad_norm": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "gs_classify_gaussians": (C.c_int, [_vp, C.c_int, _vp, C.c_float, _vp, C.c_int, _vp, C.c_float, C.c_float,
                                        C.c_float, C.c_int, _vp, _vp]),
    "gs_densify_offsets": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "gs_build_densify_output_map": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "gs_densify_gather": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 15),
    "gs_densify_plan": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "gs_densify_plan_read": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(C.c_int)]),
    "gs_build_densify_output_map_planned": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "gs_densify_gather_planned": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 8 + [C.c_ulonglong] + [_vp] * 6),
    "gs_densify_gather_planned_packed": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 8 + [C.c_ulonglong, _vp, C.POINTER(C.c_int)]),
    "gs_densify_noise": (C.c_int, [_vp, C.c_ulonglong, C.c_int, _vp]),
    "gs_ply_write": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int] + [_vp] * 6),
    "gs_ply_probe": (C.c_int, [_vp, C.c_char_p, _vp, _vp, _vp]),
    "gs_ply_load": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int] + [_vp] * 6),
    "gs_ply_pack_rows": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 7),
    "gs_set_block_work_buffer": (C.c_int, [_vp, _vp]),
    "gs_view_hint_words": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gs_set_view_hints": (C.c_int, [_vp, _vp, C.c_int]),
    "gs_set_depth_cuts": (C.c_int, [_vp, C.c_int]),
    "gs_clear_depth_cuts": (C.c_int, [_vp, _vp, C.c_int]),
    "gs_forward_missed": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gs_cut_stats": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "gs_set_grad_norm_accum": (C.c_int, [_vp, _vp]),
    "gs_block_count": (C.c_int, [_vp, _vp]),
    "gs_copy_block_work": (C.c_int, [_vp, _vp]),
    "gs_dist_topk": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gs_sh_grad_from_views_adam": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 8 + [C.c_longlong] + [C.c_float] * 6),
    "gs_loss_target_cache_floats": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "gs_set_loss_target_cache": (C.c_int, [_vp, _vp, C.c_int]),
    "gs_loss_forward_backward": (C.c_int, [_vp] + [_vp] * 5 + [C.c_float, C.c_float] + [_vp] * 3),
    "gs_adam_step": (C.c_int, [_vp, C.c_longlong] + [_vp] * 4 + [C.c_int, _vp, _vp] + [C.c_float] * 4),
    "gs_profile_enable": (C.c_int, [_vp, C.c_uint]),
    "gs_profile_read": (C.c_int, [_vp, _vp, _vp]),
    "gs_copy_last_contrib": (C.c_int, [_vp, _vp]),
    "gs_last_stats": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "gs_ctx_set_tuning": (C.c_int, [_vp, C.c_int, C.c_longlong]),
    "gs_copy_overflow_flag": (C.c_int, [_vp, _vp]),
    "gs_set_update_gate": (C.c_int, [_vp, _vp]),
    "gs_set_overflow_rider": (C.c_int, [_vp, _vp]),
    "gs_set_gathered_gate": (C.c_int, [_vp, C.c_longlong, C.c_int, _vp]),
    "gs_set_gate_seen": (C.c_int, [_vp, _vp]),
    "gs_dp_cc_floats": (C.c_longlong, [C.c_int]),
    "gs_dp_check_replicas": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong]),
    "gs_dp_check_replicas_begin": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong]),
    "gs_dp_check_replicas_end": (C.c_int, [_vp]),
    "gs_dp_check_plan": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.c_int]),
}

_lib = None


def load():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -m gaussiansplattingmlx_amd.build` "
                          "(or __graft_entry__.build()). There is no CPU fallback.")
    # torch owns the device memory handed to the library, so both must share ONE HIP runtime: import torch
    # first so that its libamdhip64 is the copy already mapped when ours resolves its dependency.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# gs_tuning (include/gsplat.h)
TUNE_FWD_WAVES_PER_SIMD, TUNE_BWD_WAVES_PER_CU, TUNE_FWD_QUADRANTS, TUNE_OP_FWD_PPL, TUNE_OP_BWD_PPL, \
    TUNE_FWD_TRACE_BUFFER, TUNE_DEPTH_GRADIENT, TUNE_WIDE_TILE_SORT, TUNE_HOST_OVERFLOW_ERRORS, TUNE_SPLITTER_DEPTH_SORT, \
    TUNE_COLOUR_RIDERS, TUNE_FWD_QUEUES, TUNE_FWD_FOUR_WAVES, TUNE_FWD_FOLD_TEST_SCALE, TUNE_POISON_CHECKPOINTS, TUNE_RENDER_ONLY, TUNE_FWD_PAIR, TUNE_FWD_SLOW_SLOT, TUNE_TRIM_RECTS = range(19)


def exported_symbols():
    return list(_SIGS)


def make_camera(view, proj, camCenter, fovX, fovY, focalX, focalY) -> gs_camera:
    import numpy as np
    cam = gs_camera()
    v = np.asarray(view, dtype=np.float32).reshape(16)
    p = np.asarray(proj, dtype=np.float32).reshape(16)
    cc = np.asarray(camCenter, dtype=np.float32).reshape(3)
    for i in range(16):
        cam.view[i] = float(v[i])
        cam.proj[i] = float(p[i])
    for i in range(3):
        cam.cam_center[i] = float(cc[i])
    cam.fov_x, cam.fov_y, cam.focal_x, cam.focal_y = float(fovX), float(fovY), float(focalX), float(focalY)
    return cam
