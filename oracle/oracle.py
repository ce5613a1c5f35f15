"""ctypes harness for oracle/liboracle_photo_icp.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package rgbd360_amd never does (it fails loudly when its HIP library is missing instead).
PARITY UNPINNED (see photo_icp_ref.cpp header): the reference has no golden vectors for this path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_photo_icp.so")

PHOTO_CONSISTENCY, DEPTH_CONSISTENCY, PHOTO_DEPTH = 0, 1, 2
PLANES = {"gray_src": 0, "gray_trg": 1, "depth_src": 2, "depth_trg": 3, "gx": 4, "gy": 5, "dgx": 6, "dgy": 7}


class Params(C.Structure):
    _fields_ = [("n_pyr", C.c_int), ("min_depth", C.c_float), ("max_depth", C.c_float), ("sigma_photo", C.c_float),
                ("sigma_depth", C.c_float), ("thres_sal_photo", C.c_float), ("thres_sal_depth", C.c_float),
                ("max_iters", C.c_int), ("tol_residual", C.c_float), ("tol_update", C.c_float), ("mask_seams", C.c_int),
                ("math_mode", C.c_int), ("reduce_mode", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int * 8), ("sso", C.c_float), ("err_final", C.c_double),
                ("rms_photo", C.c_double), ("rms_depth", C.c_double), ("hessian", C.c_float * 36),
                ("gradient", C.c_float * 6)]


class Trace(C.Structure):
    _fields_ = [("level", C.c_int), ("it", C.c_int), ("accepted", C.c_int), ("error", C.c_double),
                ("new_error", C.c_double), ("pose", C.c_float * 16), ("update", C.c_float * 6), ("n_valid", C.c_long)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "photo_icp_ref.cpp"), os.path.join(_HERE, "frame360_ref.cpp")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_photo_icp.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(Params)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_modes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for f in (L.oracle_set_target, L.oracle_set_source):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.oracle_align360.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Result)]
        L.oracle_align360_occ.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(Result)]
        L.oracle_error_occ.restype = C.c_double
        L.oracle_error_occ.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_hessgrad_occ.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(C.c_long)]
        L.oracle_set_camera.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.oracle_align_pinhole.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Result)]
        L.oracle_error_pinhole.restype = C.c_double
        L.oracle_error_pinhole.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_hessgrad_pinhole.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_long)]
        L.oracle_get_lut_pinhole.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_align_pinhole_occ.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(Result)]
        L.oracle_use_saliency.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.oracle_salient_pixels.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_error_pinhole_salient.restype = C.c_double
        L.oracle_error_pinhole_salient.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_error_pinhole_occ.restype = C.c_double
        L.oracle_error_pinhole_occ.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_hessgrad_pinhole_occ.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.POINTER(C.c_long)]
        L.oracle_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_warp_indices_pinhole.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_trace_len.argtypes = [C.c_void_p]
        L.oracle_trace_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(Trace)]
        L.oracle_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_get_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_prepare_level.argtypes = [C.c_void_p, C.c_int]
        L.oracle_get_lut.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_error.restype = C.c_double
        L.oracle_error.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.oracle_hessgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_long)]
        L.oracle_gn_step.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_forced_iters.restype = C.c_double
        L.oracle_forced_iters.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_warp_indices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_asinf_poly.restype = C.c_float
        L.oracle_asinf_poly.argtypes = [C.c_float]
        L.oracle_atan2f_poly.restype = C.c_float
        L.oracle_atan2f_poly.argtypes = [C.c_float, C.c_float]
        L.oracle_round_half_away.restype = C.c_float
        L.oracle_round_half_away.argtypes = [C.c_float]
        L.oracle_weight_huber.restype = C.c_float
        L.oracle_weight_huber.argtypes = [C.c_float, C.c_float]
        L.oracle_rank6.argtypes = [C.c_void_p]
        L.oracle_inverse6.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_se3_pseudo_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_rig_create.restype = C.c_void_p
        L.oracle_rig_create.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.oracle_rig_destroy.argtypes = [C.c_void_p]
        L.oracle_rig_set_modes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_rig_use_saliency.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.oracle_rig_set_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.oracle_rig_error.restype = C.c_double
        L.oracle_rig_error.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_rig_hessgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_long)]
        L.oracle_rig_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_rig_trace_len.argtypes = [C.c_void_p]
        L.oracle_rig_trace_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def default_params(**kw) -> Params:
    p = Params()
    lib().oracle_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def pose_to_cm(T) -> np.ndarray:
    """4x4 (row-major numpy) -> 16 floats column-major (Eigen layout at the ABI)."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T.reshape(16))


def pose_from_cm(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32).reshape(4, 4).T.copy()


class Oracle:
    """Mirror of the RegisterPhotoICP call sequence (RPI.h:480-516, 4519) on the CPU restatement."""

    def __init__(self, **params):
        self.params = default_params(**params)
        self.h = C.c_void_p(lib().oracle_create(C.byref(self.params)))
        self.result = Result()

    def close(self):
        if self.h:
            lib().oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_modes(self, math_mode: int, reduce_mode: int):
        lib().oracle_set_modes(self.h, math_mode, reduce_mode)

    def _set(self, fn, rgb, depth):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        if depth.dtype == np.uint16:
            d, dt = np.ascontiguousarray(depth), 0
        else:
            d, dt = np.ascontiguousarray(depth, dtype=np.float32), 1
        rows, cols = d.shape
        fn(self.h, _ptr(rgb), rgb.strides[0], _ptr(d), d.strides[0], dt, rows, cols)

    def set_target(self, rgb, depth):
        self._set(lib().oracle_set_target, rgb, depth)

    def set_source(self, rgb, depth):
        self._set(lib().oracle_set_source, rgb, depth)

    def align360(self, guess=None, method=PHOTO_CONSISTENCY, occlusion=0):
        g = pose_to_cm(np.eye(4) if guess is None else guess)
        out = np.zeros(16, dtype=np.float32)
        if occlusion:
            st = lib().oracle_align360_occ(self.h, _ptr(g), method, occlusion, _ptr(out), C.byref(self.result))
        else:
            st = lib().oracle_align360(self.h, _ptr(g), method, _ptr(out), C.byref(self.result))
        return st, pose_from_cm(out)

    # ---- pinhole single-sensor path (RPI.h:4254-4512)
    def set_camera(self, fx, fy, ox, oy):
        lib().oracle_set_camera(self.h, fx, fy, ox, oy)

    def align_pinhole(self, guess=None, method=PHOTO_DEPTH, occlusion=0):
        g = pose_to_cm(np.eye(4) if guess is None else guess)
        out = np.zeros(16, dtype=np.float32)
        st = lib().oracle_align_pinhole_occ(self.h, _ptr(g), method, occlusion, _ptr(out), C.byref(self.result))
        return st, pose_from_cm(out)

    def use_saliency(self, on=True, thres_saliency=0.01):
        """useSaliency(bool) RPI.h:266: the pinhole error pass runs over vSalientPixels only."""
        lib().oracle_use_saliency(self.h, int(bool(on)), thres_saliency)

    def salient_pixels(self, level) -> np.ndarray:
        n = lib().oracle_salient_pixels(self.h, level, None)
        out = np.empty(n, np.int32)
        lib().oracle_salient_pixels(self.h, level, _ptr(out))
        return out

    def error_pinhole_salient(self, level, pose, method):
        sums = np.zeros(4, np.float64)
        e = lib().oracle_error_pinhole_salient(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(sums))
        return e, sums[0], sums[1], int(sums[2]), int(sums[3])

    def error_pinhole_occ(self, level, pose, method, occlusion):
        """errorPhotoICP_Occ1 / _Occ2 -> (avPhoto + avDepth, sum photo, sum depth, n photo, n depth)."""
        sums = np.zeros(4, np.float64)
        e = lib().oracle_error_pinhole_occ(self.h, level, _ptr(pose_to_cm(pose)), method, occlusion, _ptr(sums))
        return e, sums[0], sums[1], int(sums[2]), int(sums[3])

    def hessgrad_pinhole_occ(self, level, pose, method, occlusion):
        """calcHessGrad_Occ1 / _Occ2 -> (H f32, g f32, H f64, g f64, numVisiblePixels)."""
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        nv = C.c_long()
        lib().oracle_hessgrad_pinhole_occ(self.h, level, _ptr(pose_to_cm(pose)), method, occlusion, _ptr(H), _ptr(g), _ptr(Hd), _ptr(gd),
                                          C.byref(nv))
        return H.reshape(6, 6).T.copy(), g, Hd.reshape(6, 6).T.copy(), gd, nv.value

    def error_pinhole(self, level, pose, method):
        sums = np.zeros(4, np.float64)
        e = lib().oracle_error_pinhole(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(sums))
        return e, sums[0], sums[1], int(sums[2]), int(sums[3])

    def hessgrad_pinhole(self, level, pose, method):
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        nv = C.c_long()
        lib().oracle_hessgrad_pinhole(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(H), _ptr(g), _ptr(Hd), _ptr(gd),
                                      C.byref(nv))
        return H.reshape(6, 6).T.copy(), g, Hd.reshape(6, 6).T.copy(), gd, nv.value

    def lut_pinhole(self, level) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 3), dtype=np.float32)
        lib().oracle_get_lut_pinhole(self.h, level, _ptr(out))
        return out

    def warp_indices_pinhole(self, level, pose) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 2), dtype=np.int32)
        lib().oracle_warp_indices_pinhole(self.h, level, _ptr(pose_to_cm(pose)), _ptr(out))
        return out

    def error_occ(self, level, pose, method, occlusion):
        """Occlusion-mode error pass -> (avPhoto + avDepth, sum photo, sum depth, n photo, n depth)."""
        sums = np.zeros(4, np.float64)
        e = lib().oracle_error_occ(self.h, level, _ptr(pose_to_cm(pose)), method, occlusion, _ptr(sums))
        return e, sums[0], sums[1], int(sums[2]), int(sums[3])

    def hessgrad_occ(self, level, pose, method, occlusion):
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        nv = C.c_long()
        lib().oracle_hessgrad_occ(self.h, level, _ptr(pose_to_cm(pose)), method, occlusion, _ptr(H), _ptr(g), _ptr(Hd),
                                  _ptr(gd), C.byref(nv))
        return H.reshape(6, 6).T.copy(), g, Hd.reshape(6, 6).T.copy(), gd, nv.value

    def trace(self):
        n = lib().oracle_trace_len(self.h)
        out = []
        for i in range(n):
            t = Trace()
            lib().oracle_trace_get(self.h, i, C.byref(t))
            out.append(dict(level=t.level, it=t.it, accepted=t.accepted, error=t.error, new_error=t.new_error,
                            pose=pose_from_cm(list(t.pose)), update=np.array(list(t.update), dtype=np.float32),
                            n_valid=t.n_valid))
        return out

    def level_dims(self, level):
        r, c = C.c_int(), C.c_int()
        if lib().oracle_level_dims(self.h, level, C.byref(r), C.byref(c)) != 0:
            raise IndexError(level)
        return r.value, c.value

    def plane(self, which: str, level: int) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r, c), dtype=np.float32)
        if lib().oracle_get_plane(self.h, PLANES[which], level, _ptr(out)) != 0:
            raise KeyError(which)
        return out

    def prepare_level(self, level):
        lib().oracle_prepare_level(self.h, level)

    def lut(self, level) -> np.ndarray:
        self.prepare_level(level)
        r, c = self.level_dims(level)
        out = np.empty((r * c, 3), dtype=np.float32)
        lib().oracle_get_lut(self.h, _ptr(out))
        return out

    def error(self, level, pose, method):
        e2, n = C.c_double(), C.c_long()
        rms = lib().oracle_error(self.h, level, _ptr(pose_to_cm(pose)), method, C.byref(e2), C.byref(n))
        return rms, e2.value, n.value

    def hessgrad(self, level, pose, method):
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        nv = C.c_long()
        lib().oracle_hessgrad(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(H), _ptr(g), _ptr(Hd), _ptr(gd),
                              C.byref(nv))
        return H.reshape(6, 6).T.copy(), g, Hd.reshape(6, 6).T.copy(), gd, nv.value

    def forced_iters(self, level, pose, method, n_iters):
        out = np.zeros(16, np.float32)
        e = lib().oracle_forced_iters(self.h, level, _ptr(pose_to_cm(pose)), method, n_iters, _ptr(out))
        return e, pose_from_cm(out)

    def warp_indices(self, level, pose) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 2), dtype=np.int32)
        lib().oracle_warp_indices(self.h, level, _ptr(pose_to_cm(pose)), _ptr(out))
        return out


def se3_exp(v) -> np.ndarray:
    """CPose3D::exp(v, pseudo_exponential=false) restatement: 4x4 float64."""
    out = np.zeros(16, np.float64)
    lib().oracle_se3_exp(_ptr(np.ascontiguousarray(v, np.float64)), _ptr(out))
    return out.reshape(4, 4).T.copy()


def gn_step(H, g, lam, pose):
    Hc = np.ascontiguousarray(np.asarray(H, np.float32).reshape(6, 6).T.reshape(36))
    gc = np.ascontiguousarray(np.asarray(g, np.float32))
    out = np.zeros(16, np.float32)
    upd = np.zeros(6, np.float32)
    st = lib().oracle_gn_step(_ptr(Hc), _ptr(gc), float(lam), _ptr(pose_to_cm(pose)), _ptr(out), _ptr(upd))
    return st, pose_from_cm(out), upd


def sphere_cloud(depth: np.ndarray, convention: int) -> np.ndarray:
    d = np.ascontiguousarray(depth)
    dt = 0 if d.dtype == np.uint16 else 1
    if dt == 1:
        d = np.ascontiguousarray(d, dtype=np.float32)
    out = np.empty((d.shape[0] * d.shape[1], 3), dtype=np.float32)
    f = lib().oracle_sphere_cloud
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f.restype = None
    f(_ptr(d), d.strides[0], dt, d.shape[0], d.shape[1], convention, _ptr(out))
    return out


class OraclePlane(C.Structure):
    _fields_ = [("centroid", C.c_float * 3), ("normal", C.c_float * 3), ("d", C.c_float), ("curvature", C.c_float),
                ("count", C.c_int), ("root", C.c_int), ("area", C.c_float), ("elongation", C.c_float),
                ("ppal_dir", C.c_float * 3)]


def f360_distance_map(xyz, rows, cols, max_depth_change_factor=0.05, depth_mode=1):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
    out = np.empty((rows, cols), np.float32)
    f = lib().oracle_f360_distance_map
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    f.restype = None
    f(_ptr(xyz), rows, cols, max_depth_change_factor, depth_mode, _ptr(out))
    return out


def f360_normals(xyz, rows, cols, max_depth_change_factor=0.05, normal_smoothing_size=8.0, depth_mode=1):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
    out = np.empty_like(xyz)
    win = np.empty(rows * cols, np.int32)
    f = lib().oracle_f360_normals
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    f(_ptr(xyz), rows, cols, max_depth_change_factor, normal_smoothing_size, depth_mode, _ptr(out), _ptr(win))
    return out, win.reshape(rows, cols)


def f360_plane_segment(xyz, normals, rows, cols, min_inliers=40, angular_threshold=0.05, distance_threshold=0.05,
                       max_curvature=0.001, depth_mode=1, max_planes=256):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
    normals = np.ascontiguousarray(normals, np.float32).reshape(rows * cols, 3)
    labels = np.empty(rows * cols, np.int32)
    arr = (OraclePlane * max_planes)()
    f = lib().oracle_f360_plane_segment
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p,
                  C.c_void_p, C.c_int]
    f.restype = C.c_int
    n = f(_ptr(xyz), _ptr(normals), rows, cols, min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode,
          _ptr(labels), C.cast(arr, C.c_void_p), max_planes)
    planes = [dict(centroid=np.array(list(arr[i].centroid), np.float32), normal=np.array(list(arr[i].normal), np.float32),
                   d=float(arr[i].d), curvature=float(arr[i].curvature), count=int(arr[i].count), root=int(arr[i].root),
                   area=float(arr[i].area), elongation=float(arr[i].elongation),
                   ppal_dir=np.array(list(arr[i].ppal_dir), np.float32))
              for i in range(n)]
    return labels.reshape(rows, cols), planes


def f360_plane_colour_mode(labels, rgb, planes, step=1):
    """The dominant colour of planar regions: mrpt::pbmap::Plane::calcMainColor2 (Frame360.h:1046; MRPT is third-party, restated from its
    published source) as the library defines it in integer arithmetic (frame360_kernels.h k_f360_colour / k_f360_colour_mode):
    the region's pixels with R + G + B > 0 on the grid rows r % sr == 0, columns c % sc == 0 (sr = isqrt(step), sc = step // sr,
    step = max(count // 2000, 1), count = the region's inliers), normalised colour q = (C << 16) // S; then
    getMultiDimMeanShift_color: mean = floor(sum q / n); threshold^2 = sum_k (floor(sum q_k^2 / N) - mean_k^2); while 2 n > N and the
    mean moved by more than 0.001: drop the samples farther than the threshold for good, take the mean of the rest.
    Returns a list of dicts color_mode_count / color_mode / intensity_mode / color_concentration / kept / iterations."""
    lab = np.asarray(labels, np.int64)
    rows, cols = lab.shape
    rgb = np.asarray(rgb, np.uint8)
    rr, cc = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    out = []
    for p in planes:
        m = lab == int(p["root"])
        count = int(m.sum())
        d = dict(color_mode_count=0, color_mode=np.zeros(3, np.float32), intensity_mode=0.0, color_concentration=0.0, kept=0, iterations=0)
        st = max(count // 2000, 1)
        sr = 1
        while (sr + 1) * (sr + 1) <= st:
            sr += 1
        sc = st // sr
        sel = m & (rr % sr == 0) & (cc % sc == 0)
        px = rgb[rr[sel] * step + step // 2, cc[sel] * step + step // 2].astype(np.int64)
        S = px.sum(1)
        px, S = px[S > 0], S[S > 0]
        N = len(S)
        if N == 0 or N > 4096:      # (more than a region slot holds: no dominant colour, by the device's rule -- k_f360_colour_mode)
            d["color_mode_count"] = 0
            out.append(d)
            continue
        q = (px << 16) // S[:, None]
        mean = q.sum(0) // N
        thr2 = int(sum(max(int((q[:, k] * q[:, k]).sum() // N) - int(mean[k]) ** 2, 0) for k in range(3)))
        alive = np.ones(N, bool)
        n_alive, shift2, iters, sumS = N, None, 0, int(S.sum())
        while 2 * n_alive > N and (shift2 is None or shift2 > 4294) and iters < 64:
            d2 = ((q - mean) ** 2).sum(1)
            alive &= ~(d2 > thr2)
            iters += 1
            left = int(alive.sum())
            if left == 0:
                n_alive = 0
                break
            new = q[alive].sum(0) // left
            shift2 = int(((new - mean) ** 2).sum())
            mean, n_alive, sumS = new, left, int(S[alive].sum())
        if iters == 0 or n_alive == 0:
            sumS = int(S.sum())
            if n_alive == 0:
                n_alive = N
        d.update(color_mode_count=N, color_mode=(mean / 65536.0).astype(np.float32), intensity_mode=float(np.float32(sumS / n_alive)),
                 color_concentration=float(np.float32(n_alive / N)), kept=n_alive, iterations=iters)
        out.append(d)
    return out


def f360_plane_colour(labels, rgb, planes, step=1):
    """Colour descriptors of planar regions (oracle/frame360_ref.cpp oracle_f360_plane_colour; the roles of
    mrpt::pbmap::Plane::calcMainColor / calcPlaneHistH, Frame360.h:1045-1046).  labels: rows x cols root indices; rgb: the colour
    image (cloud pixel (r, c) <- image pixel (r step + step // 2, c step + step // 2)); planes: dicts with "root".
    Returns (raw uint64 sums [n, 82], list of dicts color_count / color_nrgb / color_dev / intensity / hist_h)."""
    lab = np.ascontiguousarray(np.asarray(labels, np.int32))
    rows, cols = lab.shape
    rgb = np.ascontiguousarray(rgb, np.uint8)
    roots = np.ascontiguousarray([int(p["root"]) for p in planes], np.int32)
    out = np.zeros((max(len(planes), 1), 82), np.uint64)
    f = lib().oracle_f360_plane_colour
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    f.restype = None
    f(_ptr(lab), rows, cols, _ptr(rgb), rgb.strides[0], int(step), _ptr(roots), len(planes), _ptr(out))
    desc = []
    for k in range(len(planes)):
        w = out[k].astype(np.float64)
        n, total = w[7], w[8:].sum()
        d = dict(color_count=int(out[k, 7]), color_nrgb=np.zeros(3, np.float32), color_dev=np.zeros(3, np.float32), intensity=0.0,
                 hist_h=np.zeros(74, np.float32))
        if n > 0:
            m = w[0:3] / n / 65536.0
            var = w[3:6] / n / 65536.0 ** 2 - m * m
            d["color_nrgb"] = m.astype(np.float32)
            d["color_dev"] = np.sqrt(np.maximum(var, 0.0)).astype(np.float32)
            d["intensity"] = float(np.float32(w[6] / n))
        if total > 0:
            d["hist_h"] = (w[8:] / total).astype(np.float32)
        desc.append(d)
    return out[:len(planes)], desc


def f360_hull_stats(xyz, labels, plane):
    """EXACT convex hull of a planar region in its own plane -- the checker of the device's 64-direction hull stage (the roles of
    mrpt::pbmap::Plane::calcConvexHull / computeMassCenterAndArea, Frame360.h:1025-1031; MRPT itself is not in the reference tree):
    every pixel labelled plane["root"] is projected onto the plane through plane["centroid"] with normal plane["normal"], the hull
    comes from scipy (Qhull).  Returns (area, mass centre of the hull polygon as a 3-D point, number of hull vertices)."""
    from scipy.spatial import ConvexHull
    lab = np.asarray(labels).reshape(-1)
    pts = np.asarray(xyz, np.float64).reshape(-1, 3)[lab == int(plane["root"])]
    n = np.asarray(plane["normal"], np.float64)
    n = n / np.linalg.norm(n)
    e1 = np.cross(n, [1.0, 0.0, 0.0] if abs(n[0]) < 0.9 else [0.0, 1.0, 0.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)
    c = np.asarray(plane["centroid"], np.float64)
    uv = np.stack([(pts - c) @ e1, (pts - c) @ e2], axis=1)
    hull = ConvexHull(uv)
    poly = uv[hull.vertices]                     # counter-clockwise in 2-D
    x, y = poly[:, 0], poly[:, 1]
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    cr = x * yn - xn * y
    a2 = cr.sum()
    cu, cv = ((x + xn) * cr).sum() / (3 * a2), ((y + yn) * cr).sum() / (3 * a2)
    return abs(a2) / 2, c + cu * e1 + cv * e2, len(hull.vertices)


def f360_plane_refine(xyz, rows, cols, labels, planes, distance_threshold=0.02):
    """The refine half of segmentAndRefine (oracle/frame360_ref.cpp): returns (refined labels, planes with grown inlier sets,
    number of relabelled pixels).  planes: the dicts f360_plane_segment (or the device) returned."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
    lab = np.ascontiguousarray(np.asarray(labels, np.int32).reshape(rows * cols)).copy()
    arr = (OraclePlane * max(len(planes), 1))()
    for i, p in enumerate(planes):
        for k in range(3):
            arr[i].centroid[k] = float(p["centroid"][k]); arr[i].normal[k] = float(p["normal"][k]); arr[i].ppal_dir[k] = float(p["ppal_dir"][k])
        arr[i].d = float(p["d"]); arr[i].curvature = float(p["curvature"]); arr[i].count = int(p["count"]); arr[i].root = int(p["root"])
        arr[i].area = float(p["area"]); arr[i].elongation = float(p["elongation"])
    f = lib().oracle_f360_plane_refine
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    f.restype = C.c_int
    changed = f(_ptr(xyz), rows, cols, _ptr(lab), C.cast(arr, C.c_void_p), len(planes), distance_threshold)
    out = [dict(centroid=np.array(list(arr[i].centroid), np.float32), normal=np.array(list(arr[i].normal), np.float32),
                d=float(arr[i].d), curvature=float(arr[i].curvature), count=int(arr[i].count), root=int(arr[i].root),
                area=float(arr[i].area), elongation=float(arr[i].elongation), ppal_dir=np.array(list(arr[i].ppal_dir), np.float32))
           for i in range(len(planes))]
    return lab.reshape(rows, cols), out, changed


def sensor_cloud(depth_mm, step=2, min_depth=0.3, max_depth=10.0):
    """CloudRGBD::getPointCloud + DownsampleRGBD::downsamplePointCloud restated (oracle/frame360_ref.cpp); depth uint16 mm."""
    d = np.ascontiguousarray(depth_mm, np.uint16)
    rows, cols = d.shape
    out = np.empty((rows // step) * (cols // step) * 3, np.float32)
    f = lib().oracle_sensor_cloud
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    f.restype = None
    f(_ptr(d), d.strides[0], rows, cols, step, min_depth, max_depth, _ptr(out))
    return out.reshape(rows // step, cols // step, 3)


def fast_bilateral(xyz, rows, cols, sigma_s=10.0, sigma_r=0.05):
    """pcl::FastBilateralFilter restated (oracle/frame360_ref.cpp): organised cloud in, cloud with filtered z out."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
    out = np.empty_like(xyz)
    f = lib().oracle_fast_bilateral
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    f.restype = None
    f(_ptr(xyz), rows, cols, sigma_s, sigma_r, _ptr(out))
    return out


def stitch_sphere(rgb8, depth8, Rt_inv, K=(262.5, 262.5, 159.5, 119.5)):
    """Frame360::stitchSphericalImage restated (oracle/frame360_ref.cpp).  Rt_inv: [8,4,4] row-major numpy."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    depth8 = np.ascontiguousarray(depth8, np.uint16)
    _, rows, cols, _ = rgb8.shape
    W = rows * 8
    H = int(W * 0.5 * 60.0 / 180)
    M = np.ascontiguousarray(np.asarray(Rt_inv, np.float32).transpose(0, 2, 1).reshape(8 * 16))
    Kc = np.asarray(K, np.float32)
    out_rgb = np.empty((H, W, 3), np.uint8)
    out_d = np.empty((H, W), np.uint16)
    rp = (C.c_void_p * 8)(*[rgb8[s].ctypes.data for s in range(8)])
    dp = (C.c_void_p * 8)(*[depth8[s].ctypes.data for s in range(8)])
    f = lib().oracle_stitch_sphere
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = None
    f(rp, dp, rows, cols, _ptr(M), _ptr(Kc), _ptr(out_rgb), _ptr(out_d))
    return out_rgb, out_d


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))


class RigOracle:
    """RegisterRGBD360::RegisterDensePhotoICP (RegisterRGBD360.h:344-520) on the CPU restatement, with the three defects of the
    reference fixed as documented in photo_icp_ref.cpp (new_error at the candidate pose, jacobianRt_z, transformed depth).
    Rt: list of 4x4 sensor -> rig poses; K = (fx, fy, ox, oy) of level 0."""

    def __init__(self, Rt, K, **params):
        params.setdefault("mask_seams", 0)
        self.params = default_params(**params)
        self.n = len(Rt)
        rt = np.ascontiguousarray(np.stack([pose_to_cm(T) for T in Rt]))
        self.h = C.c_void_p(lib().oracle_rig_create(C.byref(self.params), self.n, _ptr(rt), *[float(k) for k in K]))
        self.iters = []
        self.hessian = None

    def close(self):
        if self.h:
            lib().oracle_rig_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_modes(self, math_mode: int, reduce_mode: int):
        lib().oracle_rig_set_modes(self.h, math_mode, reduce_mode)

    def use_saliency(self, on=True, thres_saliency=0.01):
        """useSaliency(bool) on the per-sensor objects: both passes run over vSalientPixels only (RPI.h:4930-5003, 5121-5262)."""
        lib().oracle_rig_use_saliency(self.h, int(bool(on)), thres_saliency)

    def set_frame(self, sensor: int, target: bool, rgb, depth):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        depth = np.ascontiguousarray(depth)
        dt = 0 if depth.dtype == np.uint16 else 1
        lib().oracle_rig_set_frame(self.h, sensor, 1 if target else 0, _ptr(rgb), rgb.strides[0], _ptr(depth), depth.strides[0], dt,
                                   rgb.shape[0], rgb.shape[1])

    def error(self, level, pose, method):
        sums = np.zeros(4, np.float64)
        e = lib().oracle_rig_error(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(sums))
        return e, sums

    def hessgrad(self, level, pose, method):
        H, g = np.zeros(36, np.float32), np.zeros(6, np.float32)
        Hd, gd = np.zeros(36, np.float64), np.zeros(6, np.float64)
        n = C.c_long()
        lib().oracle_rig_hessgrad(self.h, level, _ptr(pose_to_cm(pose)), method, _ptr(H), _ptr(g), _ptr(Hd), _ptr(gd), C.byref(n))
        return H.reshape(6, 6).T.copy(), g, Hd.reshape(6, 6).T.copy(), gd, n.value

    def align(self, guess, method):
        out, H = np.zeros(16, np.float32), np.zeros(36, np.float32)
        it = np.zeros(8, np.int32)
        st = lib().oracle_rig_align(self.h, _ptr(pose_to_cm(guess)), method, _ptr(out), _ptr(H), _ptr(it))
        self.iters = [int(x) for x in it[: self.params.n_pyr]]
        self.hessian = H.reshape(6, 6).T.copy()
        return st, pose_from_cm(out)

    def trace(self):
        out = []
        for i in range(lib().oracle_rig_trace_len(self.h)):
            lv, it, acc = C.c_int(), C.c_int(), C.c_int()
            e, ne = C.c_double(), C.c_double()
            lib().oracle_rig_trace_get(self.h, i, C.byref(lv), C.byref(it), C.byref(acc), C.byref(e), C.byref(ne))
            out.append((lv.value, it.value, acc.value, e.value, ne.value))
        return out


# ---- the sensors' intrinsic depth model (CLAMS DiscreteDepthDistortionModel; Frame360::undistort, Frame360.h:293-311, 1084-1097) ----------
# Independent numpy restatement of the reference's vendored code (OpenNI2_Grabber/third_party/CLAMS/discrete_depth_distortion_model.cpp:
# deserialize :82-92, :259-282; index :38-41; interpolatedUndistort :48-69; undistort :175-186; downsampleParams :313-320) and of its
# serialisation helpers (include/eigen_extensions/eigen_extensions.h:87-113).  Test infrastructure only.
def depth_model_read(path, downsample=1):
    import struct
    b = open(path, "rb").read()
    nl = b.index(b"\n")
    assert b[:nl] == b"DiscreteDepthDistortionModel v01"
    off = nl + 1
    width, height, bin_w, bin_h = struct.unpack_from("<iiii", b, off); off += 16
    bin_depth, = struct.unpack_from("<d", b, off); off += 8
    nbx, nby = struct.unpack_from("<ii", b, off); off += 8
    counts, mults, depths, nbins = [], [], [], []
    for _ in range(nbx * nby):
        _max_dist, = struct.unpack_from("<d", b, off); off += 8
        nb, = struct.unpack_from("<i", b, off); off += 4
        bd, = struct.unpack_from("<d", b, off); off += 8
        vecs = []
        for _k in range(4):
            by, r, c = struct.unpack_from("<iii", b, off); off += 12
            assert by == 4
            vecs.append(np.frombuffer(b, np.float32, r * c, off).copy()); off += 4 * r * c
        assert len(vecs[0]) == nb == len(vecs[3])
        counts.append(vecs[0]); mults.append(vecs[3]); depths.append(bd); nbins.append(nb)
    assert off == len(b)
    assert bin_w % downsample == 0 and bin_h % downsample == 0
    return dict(width=width // downsample, height=height // downsample, bin_width=bin_w // downsample, bin_height=bin_h // downsample, bin_depth=bin_depth,
                num_bins_x=nbx, num_bins_y=nby, counts=counts, multipliers=mults, frustum_bin_depth=depths, num_bins=nbins)


def depth_model_undistort(model, depth_m):
    """DiscreteDepthDistortionModel::undistort on a float32 image in metres (0 = no measurement): returns the corrected copy."""
    z = np.ascontiguousarray(depth_m, np.float32).copy()
    H, W = z.shape
    assert (H, W) == (model["height"], model["width"])
    for v in range(H):
        yb = min(v // model["bin_height"], model["num_bins_y"] - 1)
        for u in range(W):
            zz = z[v, u]
            if not (zz > 0):
                continue
            f = yb * model["num_bins_x"] + min(u // model["bin_width"], model["num_bins_x"] - 1)
            bd, nb, cnt, mul = model["frustum_bin_depth"][f], model["num_bins"][f], model["counts"][f], model["multipliers"][f]
            idx = min(nb - 1, int(np.floor(np.float64(zz) / bd)))
            start = np.float32(bd * idx)
            idx1 = idx if np.float64(np.float32(zz - start)) < bd / 2 else idx + 1
            idx0 = idx1 - 1
            if idx0 < 0 or idx1 >= nb or cnt[idx0] < 50 or cnt[idx1] < 50:
                z[v, u] = np.float32(zz * mul[idx])
            else:
                z0 = (idx0 + 1) * bd - bd * 0.5
                c1 = (np.float64(zz) - z0) / bd
                z[v, u] = np.float32(np.float64(zz) * ((1.0 - c1) * np.float64(mul[idx0]) + c1 * np.float64(mul[idx1])))
    return z
