"""Dense registration of two frames of the 8-sensor rig: host-side mirror of RegisterRGBD360::RegisterDensePhotoICP
(RegisterRGBD360.h:344-520) over the C ABI (rgbd360_rig_*, csrc/rig_dense.h).  The reference function is broken as written (it
never accepts a step and reads an uninitialised Jacobian row); the library implements it with the three documented fixes
(include/rgbd360_hip.h).  All arithmetic runs in the HIP library; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .register import Rgbd360Error, _ptr, pose_from_cm, pose_to_cm


class RegisterDensePhotoICP:
    PHOTO_CONSISTENCY, DEPTH_CONSISTENCY, PHOTO_DEPTH = 0, 1, 2

    def __init__(self, Rt, K, n_pyr: int = 4, device: int = 0, **params):
        """Rt: the sensors' 4x4 sensor -> rig poses (Calib360::Rt_); K = (fx, fy, ox, oy) of the full-resolution sensor image."""
        self._L = _lib.load()
        self._p = _lib.Params()
        self._L.rgbd360_default_params(C.byref(self._p))
        self._p.n_pyr = int(n_pyr)
        self._p.device = int(device)
        for k, v in params.items():
            setattr(self._p, k, v)
        self.n_sensors = len(Rt)
        rt = np.ascontiguousarray(np.stack([pose_to_cm(T) for T in Rt]))
        h = C.c_void_p()
        rc = self._L.rgbd360_rig_create(C.byref(self._p), self.n_sensors, _ptr(rt), *[float(k) for k in K], C.byref(h))
        if rc != 0:
            raise Rgbd360Error(f"rgbd360_rig_create failed ({rc}): no usable HIP device or bad arguments; there is no CPU fallback")
        self._h = h
        self._res = _lib.Result()
        self._pose = np.eye(4, dtype=np.float32)
        self.num_iterations = []
        self.status = 0

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.rgbd360_rig_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=(0,)):
        if rc not in allow:
            raise Rgbd360Error(f"{self._L.rgbd360_rig_last_error(self._h).decode()} ({rc})")
        return rc

    def _set(self, fn, frame):
        if len(frame) != self.n_sensors:
            raise Rgbd360Error("a frame must hold one (rgb, depth) pair per sensor")
        rgbs = [np.ascontiguousarray(f[0], np.uint8) for f in frame]
        deps = [np.ascontiguousarray(f[1]) for f in frame]
        shape, dtype = deps[0].shape, deps[0].dtype
        if dtype not in (np.uint16, np.float32):
            raise Rgbd360Error("depth must be uint16 millimetres or float32 metres")
        for r, d in zip(rgbs, deps):
            if r.shape != shape + (3,) or d.shape != shape or d.dtype != dtype:
                raise Rgbd360Error("all sensor images of a frame must share one size and depth type")
        rp = (C.c_void_p * self.n_sensors)(*[r.ctypes.data for r in rgbs])
        dp = (C.c_void_p * self.n_sensors)(*[d.ctypes.data for d in deps])
        self._check(fn(self._h, rp, shape[1] * 3, dp, shape[1] * dtype.itemsize, 0 if dtype == np.uint16 else 1, shape[0], shape[1]))

    def setTargetFrame(self, frame1):       # frame1->frameRGBD_[s] (RegisterRGBD360.h:376)
        self._set(self._L.rgbd360_rig_set_target, frame1)

    def setSourceFrame(self, frame2):       # frame2->frameRGBD_[s] (RegisterRGBD360.h:375)
        self._set(self._L.rgbd360_rig_set_source, frame2)

    def useSaliency(self, flag: bool, thresSaliency: float = 0.01):
        """useSaliency(bool) on the per-sensor RegisterPhotoICP objects (RPI.h:266): both passes run over vSalientPixels only."""
        self._check(self._L.rgbd360_rig_use_saliency(self._h, int(bool(flag)), float(thresSaliency)))

    def eval(self, level: int, pose, method: int):
        e2, ns = np.zeros(2, np.float64), np.zeros(2, np.int64)
        H, g = np.zeros(36, np.float32), np.zeros(6, np.float32)
        H64, g64 = np.zeros(36, np.float64), np.zeros(6, np.float64)
        nr = C.c_longlong()
        self._check(self._L.rgbd360_rig_eval(self._h, level, _ptr(pose_to_cm(pose)), method, _ptr(e2), _ptr(ns), _ptr(H), _ptr(g), _ptr(H64),
                                             _ptr(g64), C.byref(nr)))
        return dict(err2=float(e2.sum()), err2_split=e2, n_split=ns, H=H.reshape(6, 6).T.copy(), g=g, H64=H64.reshape(6, 6).T.copy(), g64=g64,
                    n_rows=nr.value)

    def align(self, pose_estim=None, method: int = 0) -> bool:
        """RegisterDensePhotoICP(frame1, frame2, pose_estim, method): True unless the problem is ill-posed; getPose() / getInfoMat()."""
        g = np.eye(4) if pose_estim is None else pose_estim
        out = np.zeros(16, np.float32)
        rc = self._L.rgbd360_rig_align(self._h, _ptr(pose_to_cm(g)), int(method), _ptr(out), C.byref(self._res))
        self._check(rc, allow=(0, 1))
        self._pose = pose_from_cm(out)
        self.status = rc
        self.num_iterations = [int(self._res.iters[l]) for l in range(self._p.n_pyr)]
        return rc == 0

    def getPose(self) -> np.ndarray:           # rigidTransf
        return self._pose.copy()

    def getInfoMat(self) -> np.ndarray:        # informationM = Hessian (RegisterRGBD360.h:511)
        return np.asarray(list(self._res.hessian), dtype=np.float32).reshape(6, 6).T.copy()

    @property
    def error(self) -> float:
        return float(self._res.err_final)
