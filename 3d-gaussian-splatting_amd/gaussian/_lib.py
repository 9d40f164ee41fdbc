"""ctypes loader for libgs_amd.so (the C ABI in include/gs_abi.h).

There is NO CPU fallback: if the shared library is missing or cannot be loaded, importing
``gaussian`` raises.  The library is built in-tree by ``gs_build.build()`` /
``__graft_entry__.build()``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_AMD_LIB: an experiment variant of the library (gs_build.build(outdir=...), tools/ab_variants.py); the product
# always loads the in-tree build
LIB_PATH = os.environ.get("GS_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libgs_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"libgs_amd.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` "
        "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the rasterizer.")

lib = C.CDLL(LIB_PATH)

vp, i64, i32, f32, ci, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int, C.c_size_t


class GsFrame(C.Structure):
    """Mirror of ``struct gs_frame`` (include/gs_abi.h)."""

    _fields_ = [
        ("N", i64), ("color_dim", i32), ("scale_activation", i32),
        ("pos", vp), ("quat", vp), ("scale", vp), ("opa", vp), ("rgb", vp),
        ("rot", f32 * 9), ("tran", f32 * 3),
        ("near_plane", f32), ("half_width", f32), ("half_height", f32),
        ("width", i32), ("height", i32),
        ("focal_x", f32), ("focal_y", f32),
        ("tile_length_x", f32), ("tile_length_y", f32), ("leftmost", f32), ("topmost", f32),
        ("thresh", f32),
        ("rays_o", f32 * 3), ("lefttop", f32 * 3), ("vec_dx", f32 * 3), ("vec_dy", f32 * 3),
        ("max_pairs", i64), ("workspace", vp), ("workspace_bytes", sz),
        ("image", vp), ("image_padded", vp),
        ("training", i32), ("sort_mode", i32), ("tile_culling_method", i32),
        ("async_", vp), ("flags", i32),  # ABI 3 (`async` is a Python keyword)
    ]


GS_FRAME_EMIT_SORTED_KEYS = 1
GS_FRAME_SLICE_SORT = 2
GS_FRAME_TABLE_BIN = 4
GS_FRAME_SERIAL_LONG_LISTS = 8
GS_FRAME_LONG_LISTS = 16
GS_FRAME_STRIP_BIN = 32
GS_FRAME_BWD_ROWS = 64
GS_FRAME_LONG_SORT = 128
GS_FRAME_OCCLUSION_CULL = 256
GS_FRAME_CULL_DILATE = 512
GS_FRAME_CULL_DILATE_NEAR = 1024


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


gs_last_error = _sig("gs_last_error", C.c_char_p)
gs_abi_version = _sig("gs_abi_version", ci)
gs_culling = _sig("gs_culling", ci)
gs_world2camera = _sig("gs_world2camera", ci, vp, vp, vp, vp, i64, vp)
gs_world2camera_backward = _sig("gs_world2camera_backward", ci, vp, vp, vp, i64, vp)
gs_jacobian = _sig("gs_jacobian", ci, vp, vp, i64, vp)
gs_global_culling = _sig("gs_global_culling", ci, vp, vp, vp, vp, vp, i64, f32, f32, f32, vp, vp, vp, vp)
gs_global_culling_backward = _sig("gs_global_culling_backward", ci, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                  vp, vp)
gs_calc_tile_list = _sig("gs_calc_tile_list", ci, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, f32, ci, f32, f32,
                         i32, i32, f32, f32, vp)
gs_gather_gaussians = _sig("gs_gather_gaussians", ci, vp, vp, vp, vp, i64, i64, i64, vp)
gs_draw = _sig("gs_draw", ci, vp, vp, vp, vp, vp, vp, i32, i32, i64, f32, f32, ci, ci, ci, vp, vp, vp, vp, ci, vp)
gs_draw_backward_workspace_bytes = _sig("gs_draw_backward_workspace_bytes", sz, i64, i32, i32)
gs_draw_backward = _sig("gs_draw_backward", ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, f32, f32,
                        ci, ci, ci, vp, vp, vp, vp, ci, vp, sz, vp)
gs_sort_pairs_tmp_bytes = _sig("gs_sort_pairs_tmp_bytes", sz, i64)
gs_sort_pairs = _sig("gs_sort_pairs", ci, vp, vp, vp, vp, vp, i64, ci, vp, sz, C.POINTER(ci), vp)
gs_sort_pairs_bits = _sig("gs_sort_pairs_bits", ci, vp, vp, vp, vp, vp, i64, ci, ci, vp, sz, C.POINTER(ci), vp)
gs_frame_workspace_bytes = _sig("gs_frame_workspace_bytes", sz, i64, i64, i32, i32, i32, i32)
gs_frame_forward = _sig("gs_frame_forward", ci, C.POINTER(GsFrame), vp)
gs_frame_backward = _sig("gs_frame_backward", ci, C.POINTER(GsFrame), vp, vp, vp, vp, vp, vp, vp)
gs_frame_backward_part = _sig("gs_frame_backward_part", ci, C.POINTER(GsFrame), vp, vp, vp, vp, vp, vp, i32, vp)
GS_BWD_RASTER, GS_BWD_GEOMETRY, GS_BWD_COLOR = 1, 2, 4
gs_frame_backward_slice = _sig("gs_frame_backward_slice", ci, C.POINTER(GsFrame), vp, vp, vp, vp, vp, i32, i64, i64, vp)
gs_frame_project_slices = _sig("gs_frame_project_slices", ci, C.POINTER(GsFrame), C.POINTER(i32), C.POINTER(i64))
gs_frame_forward_project = _sig("gs_frame_forward_project", ci, C.POINTER(GsFrame), i32, i32, vp)
gs_frame_forward_rest = _sig("gs_frame_forward_rest", ci, C.POINTER(GsFrame), vp)
gs_frame_forward_profile = _sig("gs_frame_forward_profile", ci, C.POINTER(GsFrame), C.POINTER(f32), vp)
gs_frame_backward_profile = _sig("gs_frame_backward_profile", ci, C.POINTER(GsFrame), vp, vp, vp, vp, vp, vp,
                                 C.POINTER(f32), vp)
gs_frame_stats_async = _sig("gs_frame_stats_async", ci, C.POINTER(GsFrame), vp, vp)
gs_frame_longest_list_async = _sig("gs_frame_longest_list_async", ci, C.POINTER(GsFrame), vp, vp)
gs_frame_stats_tagged_async = _sig("gs_frame_stats_tagged_async", ci, C.POINTER(GsFrame), C.c_uint32, vp, vp)
gs_frame_cull_fallback_async = _sig("gs_frame_cull_fallback_async", ci, C.POINTER(GsFrame), vp, vp)
gs_frame_is_occlusion_culled = _sig("gs_frame_is_occlusion_culled", ci, C.POINTER(GsFrame), C.POINTER(C.c_int32))
GS_STATS_TAGGED_N = 15
gs_frame_debug_views = _sig("gs_frame_debug_views", ci, C.POINTER(GsFrame), C.POINTER(vp), C.POINTER(vp),
                            C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp))

gs_frame_binning_variant = _sig("gs_frame_binning_variant", ci, C.POINTER(GsFrame))
gs_frame_debug_rects = _sig("gs_frame_debug_rects", ci, C.POINTER(GsFrame), C.POINTER(vp))
class GsAdamFused(C.Structure):
    """Mirror of ``struct gs_adam_fused`` (include/gs_abi.h): moments / learning rates in the order pos, quat, scale, opa, rgb."""

    _fields_ = [("exp_avg", vp * 5), ("exp_avg_sq", vp * 5), ("lr", C.c_float * 5),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", i64),
                ("grad_stat", vp), ("stat_mode", i32), ("skip_if_nonzero", vp)]


gs_frame_backward_adam = _sig("gs_frame_backward_adam", ci, C.POINTER(GsFrame), vp, C.POINTER(GsAdamFused), vp)
gs_frame_debug_tile_nproc = _sig("gs_frame_debug_tile_nproc", ci, C.POINTER(GsFrame), C.POINTER(vp))
gs_frame_debug_bwd_exec_rows = _sig("gs_frame_debug_bwd_exec_rows", ci, C.POINTER(GsFrame), C.POINTER(vp), C.POINTER(C.c_int32))

gs_adam_step = _sig("gs_adam_step", ci, vp, vp, vp, vp, i64, i32, C.POINTER(i64), C.POINTER(f32), f32, f32, f32, i64,
                    vp, i64, i64, i32, vp)
gs_adam_step_range = _sig("gs_adam_step_range", ci, vp, vp, vp, vp, i64, i64, i64, i32, C.POINTER(i64), C.POINTER(f32),
                          f32, f32, f32, i64, vp, i64, i64, i32, vp)
gs_adam_step_sharded = _sig("gs_adam_step_sharded", ci, vp, vp, vp, vp, i64, i64, i64, i64, i32, C.POINTER(i64),
                            C.POINTER(f32), f32, f32, f32, i64, vp, i64, i64, i32, vp, vp)
gs_adam_step_multi = _sig("gs_adam_step_multi", ci, vp, vp, vp, vp, i64, i32, C.POINTER(i64), C.POINTER(i64),
                          C.POINTER(i64), i32, C.POINTER(i64), C.POINTER(f32), f32, f32, f32, i64, vp, i64, i64, i32, vp,
                          f32, vp)
gs_frame_overflow_flag = _sig("gs_frame_overflow_flag", ci, C.POINTER(GsFrame), C.POINTER(vp))
gs_grad_stat_update = _sig("gs_grad_stat_update", ci, vp, vp, i64, i32, vp)
gs_frame_async_create = _sig("gs_frame_async_create", ci, C.POINTER(vp))
gs_frame_async_wait = _sig("gs_frame_async_wait", ci, vp, vp)
gs_frame_async_destroy = _sig("gs_frame_async_destroy", ci, vp)
gs_loss_workspace_bytes = _sig("gs_loss_workspace_bytes", sz, i32, i32)
gs_loss_l1_ssim = _sig("gs_loss_l1_ssim", ci, vp, vp, i32, i32, f32, vp, vp, vp, sz, vp)



class GsDensifyOpts(C.Structure):
    """Mirror of ``struct gs_densify_opts``."""

    _fields_ = [("taus", f32), ("delete_thresh", f32), ("grad_thresh", f32), ("clone_dt", f32),
                ("scale_activation", i32), ("grad_aggregation", i32), ("use_clone", i32), ("use_split", i32),
                ("color_dim", i32)]


gs_densify_workspace_bytes = _sig("gs_densify_workspace_bytes", sz, i64)
gs_densify_classify = _sig("gs_densify_classify", ci, vp, vp, vp, i64, C.POINTER(GsDensifyOpts), vp, vp, sz, vp)
gs_densify_apply = _sig("gs_densify_apply", ci, vp, vp, vp, vp, vp, vp, i64, C.POINTER(GsDensifyOpts), vp, vp, i64,
                        vp, vp, vp, vp, vp, i64, vp, vp, sz, vp)

# Every symbol include/gs_abi.h declares (checked by tests/test_abi.py without a GPU).
EXPORTS = [
    "gs_last_error", "gs_abi_version", "gs_culling", "gs_world2camera", "gs_world2camera_backward",
    "gs_jacobian", "gs_global_culling", "gs_global_culling_backward", "gs_calc_tile_list",
    "gs_gather_gaussians", "gs_draw", "gs_draw_backward_workspace_bytes", "gs_draw_backward",
    "gs_sort_pairs_tmp_bytes", "gs_sort_pairs", "gs_sort_pairs_bits", "gs_frame_workspace_bytes", "gs_frame_forward",
    "gs_frame_stats_async", "gs_frame_longest_list_async", "gs_frame_stats_tagged_async", "gs_frame_cull_fallback_async", "gs_frame_is_occlusion_culled", "gs_frame_debug_views", "gs_frame_debug_rects", "gs_frame_binning_variant", "gs_frame_debug_tile_nproc", "gs_frame_debug_bwd_exec_rows", "gs_frame_backward", "gs_frame_backward_adam", "gs_frame_forward_profile",
    "gs_frame_backward_part", "gs_frame_async_create", "gs_frame_async_wait", "gs_frame_async_destroy",
    "gs_frame_backward_slice", "gs_frame_project_slices", "gs_frame_forward_project", "gs_frame_forward_rest",
    "gs_adam_step_multi",
    "gs_frame_backward_profile", "gs_adam_step", "gs_adam_step_range", "gs_adam_step_sharded", "gs_frame_overflow_flag", "gs_grad_stat_update", "gs_loss_workspace_bytes", "gs_loss_l1_ssim",
    "gs_densify_workspace_bytes", "gs_densify_classify", "gs_densify_apply",
]


def check(rc: int, what: str):
    """Map the C return convention onto the reference's: a Python RuntimeError."""
    if rc != 0:
        msg = gs_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
