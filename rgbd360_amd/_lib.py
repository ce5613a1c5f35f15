"""ctypes binding of include/rgbd360_hip.h.  Loading fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGBD360_LIB") or os.path.join(_HERE, "lib", "librgbd360_hip.so")     # RGBD360_LIB: another build of the library (A/B runs)

# Every symbol include/rgbd360_hip.h declares (checked by tests/test_abi.py against the header text).
SYMBOLS = [
    "rgbd360_default_params", "rgbd360_create", "rgbd360_destroy", "rgbd360_last_error", "rgbd360_set_target",
    "rgbd360_set_source", "rgbd360_set_target_dev", "rgbd360_set_source_dev", "rgbd360_promote_source_to_target",
    "rgbd360_align360", "rgbd360_align360_begin", "rgbd360_align360_finish", "rgbd360_align360_batch", "rgbd360_align360_batch_dev", "rgbd360_level_dims", "rgbd360_get_plane", "rgbd360_get_lut", "rgbd360_eval", "rgbd360_eval_occ",
    "rgbd360_warp_indices", "rgbd360_gn_step", "rgbd360_forced_iters", "rgbd360_time_eval_kernel", "rgbd360_stream",
    "rgbd360_sync", "rgbd360_device_count", "rgbd360_sphere_cloud", "rgbd360_selftest_math", "rgbd360_selftest_libm", "rgbd360_set_index_arithmetic", "rgbd360_get_index_arithmetic", "rgbd360_time_solve_kernel", "rgbd360_normals", "rgbd360_distance_map",
    "rgbd360_plane_fit", "rgbd360_frame_planes", "rgbd360_frame_planes_dev", "rgbd360_load_frame_bin", "rgbd360_stitch_sphere",
    "rgbd360_set_camera", "rgbd360_align_pinhole", "rgbd360_eval_pinhole", "rgbd360_eval_pinhole_occ", "rgbd360_use_saliency", "rgbd360_warp_indices_pinhole",
    "rgbd360_pbmap_default_params", "rgbd360_register_planes", "rgbd360_bilateral_filter",
    "rgbd360_cloud_planes", "rgbd360_sensor_cloud", "rgbd360_sensor_planes", "rgbd360_sensor_planes_ex", "rgbd360_sensor_cloud_ex", "rgbd360_depth_model_load", "rgbd360_depth_model_free", "rgbd360_depth_model_info", "rgbd360_depth_model_undistort", "rgbd360_merge_planes", "rgbd360_group_planes", "rgbd360_pool_sensor_planes", "rgbd360_debug_set_schedule", "rgbd360_debug_set_sequence_route", "rgbd360_debug_knobs_enabled", "rgbd360_frame_planes_stage_timing", "rgbd360_frame_planes_stage_times", "rgbd360_planes_available", "rgbd360_set_plane_refinement", "rgbd360_plane_refinement_stats", "rgbd360_set_plane_color_image",
    "rgbd360_multi_create", "rgbd360_multi_destroy", "rgbd360_multi_last_error", "rgbd360_multi_n_gpus", "rgbd360_multi_uses_rccl", "rgbd360_multi_set_index_arithmetic",
    "rgbd360_shard_range", "rgbd360_gather_slot", "rgbd360_multi_align_sequence", "rgbd360_multi_load_sequence", "rgbd360_multi_align_resident",
    "rgbd360_align360_batch_multi", "rgbd360_time_eval_kernel_rotating", "rgbd360_forced_iters_batch",
    "rgbd360_rig_create", "rgbd360_rig_destroy", "rgbd360_rig_last_error", "rgbd360_rig_set_target", "rgbd360_rig_set_source",
    "rgbd360_rig_eval", "rgbd360_rig_align", "rgbd360_rig_use_saliency", "rgbd360_debug_solve_partials",
]


class Params(C.Structure):
    _fields_ = [("n_pyr", C.c_int), ("min_depth", C.c_float), ("max_depth", C.c_float), ("sigma_photo", C.c_float),
                ("sigma_depth", C.c_float), ("thres_sal_photo", C.c_float), ("thres_sal_depth", C.c_float),
                ("max_iters", C.c_int), ("tol_residual", C.c_float), ("tol_update", C.c_float), ("mask_seams", C.c_int),
                ("device", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int * 8), ("sso", C.c_float), ("err_final", C.c_double),
                ("rms_photo", C.c_double), ("rms_depth", C.c_double), ("hessian", C.c_float * 36),
                ("gradient", C.c_float * 6)]


class Plane(C.Structure):
    _fields_ = [("centroid", C.c_float * 3), ("normal", C.c_float * 3), ("d", C.c_float), ("curvature", C.c_float),
                ("count", C.c_int), ("root", C.c_int), ("area", C.c_float), ("elongation", C.c_float),
                ("ppal_dir", C.c_float * 3), ("area_moment", C.c_float), ("center_hull", C.c_float * 3), ("hull_points", C.c_int),
                ("color_count", C.c_int), ("color_nrgb", C.c_float * 3), ("color_dev", C.c_float * 3), ("intensity", C.c_float),
                ("hist_h", C.c_float * 74), ("hull_n", C.c_int), ("hull", (C.c_float * 3) * 64),
                ("color_mode_count", C.c_int), ("color_mode", C.c_float * 3), ("intensity_mode", C.c_float), ("color_concentration", C.c_float)]


class PbmapParams(C.Structure):
    _fields_ = [("dist_d", C.c_float), ("angle_deg", C.c_float), ("elongation_threshold", C.c_float),
                ("area_threshold", C.c_float), ("dist_threshold", C.c_float), ("angle_threshold_deg", C.c_float),
                ("height_threshold", C.c_float), ("cos_normal_threshold", C.c_float), ("min_planes_recognition", C.c_int),
                ("max_curvature_plane", C.c_float), ("min_area_plane", C.c_float), ("max_elongation_plane", C.c_float),
                ("up_axis", C.c_int), ("planar_normal_tol", C.c_float), ("max_conditioning", C.c_float),
                ("sigma_dist", C.c_float), ("sigma_normal", C.c_float), ("max_nodes", C.c_int),
                ("use_color", C.c_int), ("color_threshold", C.c_float), ("intensity_threshold", C.c_float), ("hue_threshold", C.c_float)]


_lib = None


def load() -> C.CDLL:
    """Returns the loaded HIP library or raises; never substitutes another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(rgbd360_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.c_void_p
    L.rgbd360_default_params.argtypes = [C.POINTER(Params)]
    L.rgbd360_default_params.restype = None
    L.rgbd360_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.rgbd360_destroy.argtypes = [vp]
    L.rgbd360_destroy.restype = None
    L.rgbd360_last_error.argtypes = [vp]
    L.rgbd360_last_error.restype = C.c_char_p
    for f in (L.rgbd360_set_target, L.rgbd360_set_source, L.rgbd360_set_target_dev, L.rgbd360_set_source_dev):
        f.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32]
    L.rgbd360_promote_source_to_target.argtypes = [vp]
    L.rgbd360_align360.argtypes = [vp, f32p, i32, i32, f32p, C.POINTER(Result)]
    L.rgbd360_align360_begin.argtypes = [vp, f32p, i32, i32]
    L.rgbd360_align360_finish.argtypes = [vp, f32p, C.POINTER(Result)]
    L.rgbd360_align360_batch.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32, f32p, i32, i32, i32, f32p, vp]
    L.rgbd360_align360_batch_dev.argtypes = L.rgbd360_align360_batch.argtypes
    L.rgbd360_level_dims.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_get_plane.argtypes = [vp, i32, i32, f32p]
    L.rgbd360_get_lut.argtypes = [vp, i32, f32p]
    L.rgbd360_eval.argtypes = [vp, i32, f32p, i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), vp, vp, vp, vp, vp, vp,
                               C.POINTER(C.c_longlong)]
    L.rgbd360_eval_occ.argtypes = [vp, i32, f32p, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), vp, vp, vp, vp, vp, vp,
                                   C.POINTER(C.c_longlong)]
    L.rgbd360_warp_indices.argtypes = [vp, i32, f32p, vp]
    L.rgbd360_set_camera.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    L.rgbd360_align_pinhole.argtypes = [vp, f32p, i32, i32, f32p, C.POINTER(Result)]
    L.rgbd360_eval_pinhole.argtypes = [vp, i32, f32p, i32, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_longlong)]
    L.rgbd360_eval_pinhole_occ.argtypes = [vp, i32, f32p, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_longlong)]
    L.rgbd360_use_saliency.argtypes = [vp, i32, C.c_float]
    L.rgbd360_warp_indices_pinhole.argtypes = [vp, i32, f32p, vp]
    L.rgbd360_gn_step.argtypes = [vp, f32p, f32p, C.c_float, f32p, f32p, f32p]
    L.rgbd360_forced_iters.argtypes = [vp, i32, f32p, i32, i32, f32p, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.rgbd360_time_eval_kernel.argtypes = [vp, i32, f32p, i32, i32, i32, C.POINTER(C.c_float)]
    L.rgbd360_stream.argtypes = [vp]
    L.rgbd360_stream.restype = vp
    L.rgbd360_sync.argtypes = [vp]
    L.rgbd360_device_count.argtypes = []
    L.rgbd360_time_solve_kernel.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float)]
    L.rgbd360_normals.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, i32, vp]
    L.rgbd360_distance_map.argtypes = [vp, vp, i32, i32, C.c_float, i32, vp]
    L.rgbd360_bilateral_filter.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, vp]
    L.rgbd360_sensor_cloud.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, C.c_float, C.c_float, vp]
    L.rgbd360_sensor_planes.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32,
                                        C.c_float, C.c_float, C.c_float, vp, vp, i32, C.POINTER(i32)]
    L.rgbd360_merge_planes.argtypes = [vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, i32, C.POINTER(i32)]
    L.rgbd360_sensor_cloud_ex.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, C.c_float, C.c_float, vp]
    L.rgbd360_sensor_planes_ex.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32,
                                           C.c_float, C.c_float, C.c_float, vp, vp, i32, C.POINTER(i32)]
    L.rgbd360_depth_model_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.rgbd360_depth_model_free.argtypes = [vp]
    L.rgbd360_depth_model_free.restype = None
    L.rgbd360_depth_model_info.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double)]
    L.rgbd360_depth_model_undistort.argtypes = [vp, vp, C.c_size_t, i32, i32]
    L.rgbd360_pool_sensor_planes.argtypes = [vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, i32, C.POINTER(i32)]
    L.rgbd360_pool_sensor_planes.restype = i32
    L.rgbd360_debug_set_schedule.argtypes = [vp, i32, i32]
    L.rgbd360_debug_set_sequence_route.argtypes = [vp, i32, i32]
    L.rgbd360_debug_knobs_enabled.argtypes = []
    L.rgbd360_frame_planes_stage_timing.argtypes = [vp, i32]
    L.rgbd360_frame_planes_stage_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.rgbd360_group_planes.argtypes = [vp, C.POINTER(i32), i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, i32, C.POINTER(i32)]
    L.rgbd360_cloud_planes.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_float, C.c_float, C.c_float, i32, vp,
                                       vp, i32, C.POINTER(i32)]
    L.rgbd360_plane_fit.argtypes = [vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, C.c_float, i32, vp, vp, i32, C.POINTER(i32)]
    L.rgbd360_frame_planes.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, C.c_float, C.c_float, i32, C.c_float, C.c_float, C.c_float,
                                       i32, vp, vp, vp, vp, i32, C.POINTER(i32)]
    L.rgbd360_frame_planes_dev.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, C.c_float, C.c_float, i32, C.c_float, C.c_float,
                                           C.c_float, i32, vp, i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.rgbd360_load_frame_bin.argtypes = [C.c_char_p, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_stitch_sphere.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_selftest_math.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.rgbd360_selftest_libm.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.rgbd360_set_index_arithmetic.argtypes = [vp, C.c_int]
    L.rgbd360_get_index_arithmetic.argtypes = [vp]
    L.rgbd360_sphere_cloud.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, f32p]
    L.rgbd360_pbmap_default_params.argtypes = [C.POINTER(PbmapParams), i32]
    L.rgbd360_pbmap_default_params.restype = None
    L.rgbd360_register_planes.argtypes = [vp, i32, vp, i32, i32, i32, C.POINTER(PbmapParams), vp, vp, vp, C.POINTER(i32),
                                          C.POINTER(C.c_float)]
    L.rgbd360_planes_available.argtypes = [vp]
    L.rgbd360_forced_iters_batch.argtypes = [vp, i32, vp, vp, vp, vp, C.c_size_t, C.c_size_t, i32, i32, i32, i32, f32p, i32, i32, f32p,
                                             C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rgbd360_debug_solve_partials.argtypes = [vp, i32, vp, i32, i32, vp, f32p, f32p]
    L.rgbd360_set_plane_refinement.argtypes = [vp, i32, C.c_float]
    L.rgbd360_set_plane_color_image.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32]
    L.rgbd360_plane_refinement_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_rig_create.argtypes = [C.POINTER(Params), i32, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(vp)]
    L.rgbd360_rig_destroy.argtypes = [vp]
    L.rgbd360_rig_destroy.restype = None
    L.rgbd360_rig_last_error.argtypes = [vp]
    L.rgbd360_rig_last_error.restype = C.c_char_p
    for f in (L.rgbd360_rig_set_target, L.rgbd360_rig_set_source):
        f.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32]
    L.rgbd360_rig_eval.argtypes = [vp, i32, f32p, i32, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_longlong)]
    L.rgbd360_rig_align.argtypes = [vp, f32p, i32, f32p, C.POINTER(Result)]
    L.rgbd360_rig_use_saliency.argtypes = [vp, i32, C.c_float]
    L.rgbd360_time_eval_kernel_rotating.argtypes = [vp, i32, i32, f32p, i32, i32, i32, C.POINTER(C.c_float)]
    L.rgbd360_multi_create.argtypes = [C.POINTER(Params), i32, vp, C.POINTER(vp)]
    L.rgbd360_multi_destroy.argtypes = [vp]
    L.rgbd360_multi_destroy.restype = None
    L.rgbd360_multi_last_error.argtypes = [vp]
    L.rgbd360_multi_last_error.restype = C.c_char_p
    L.rgbd360_multi_n_gpus.argtypes = [vp]
    L.rgbd360_multi_uses_rccl.argtypes = [vp]
    L.rgbd360_multi_set_index_arithmetic.argtypes = [vp, C.c_int]
    L.rgbd360_shard_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_shard_range.restype = None
    L.rgbd360_gather_slot.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.rgbd360_gather_slot.restype = None
    L.rgbd360_multi_align_sequence.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32, f32p, i32, i32, i32, f32p, vp]
    L.rgbd360_multi_load_sequence.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32]
    L.rgbd360_multi_align_resident.argtypes = [vp, f32p, i32, i32, i32, f32p, vp]
    L.rgbd360_align360_batch_multi.argtypes = [C.POINTER(Params), i32, vp, C.c_size_t, vp, C.c_size_t, i32, i32, i32, f32p, i32, i32, i32,
                                               i32, vp, f32p, vp]
    _lib = L
    return L
