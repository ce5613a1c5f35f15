"""Host-side mirror of the reference's RegisterPhotoICP interface for the spherical path, over the C ABI.

Method names, argument meaning and error behaviour follow include/RegisterPhotoICP.h ("RPI.h") of
EduFdez/rgbd360 so that callers such as OdometryRGBD360.cpp:189-193 read the same:

    align360 = RegisterPhotoICP(); align360.setNumPyr(5); align360.useSaliency(False)
    align360.setTargetFrame(rgb1, depth1); align360.setSourceFrame(rgb2, depth2)
    align360.alignFrames360(guess, RegisterPhotoICP.PHOTO_DEPTH)
    pose = align360.getOptimalPose()

All arithmetic runs in the HIP library (rgbd360_amd/lib/librgbd360_hip.so); there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def pose_to_cm(T) -> np.ndarray:
    """4x4 numpy (row-major) -> 16 floats column-major, Eigen's layout at the ABI."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T.reshape(16))


def pose_from_cm(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32).reshape(4, 4).T.copy()


class Rgbd360Error(RuntimeError):
    pass


class RegisterPhotoICP:
    # costFuncType, RPI.h:194
    PHOTO_CONSISTENCY, DEPTH_CONSISTENCY, PHOTO_DEPTH = 0, 1, 2

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._p = _lib.Params()
        self._L.rgbd360_default_params(C.byref(self._p))   # RPI.h:201-221
        self._p.device = device
        self._h = None
        self._res = _lib.Result()
        self._pose = np.eye(4, dtype=np.float32)
        # public fields of the reference (RPI.h:177-189)
        self.SSO = 0.0
        self.avResidual = 0.0
        self.avPhotoResidual = 0.0
        self.avDepthResidual = 0.0
        self.num_iterations = []
        self.status = 0

    # ---- setters (RPI.h:224-269); they must precede the first frame, like the reference's constructor-time state
    def _dirty(self):
        if self._h is not None:
            self._L.rgbd360_destroy(self._h)
            self._h = None

    @property
    def nPyrLevels(self) -> int:
        return self._p.n_pyr

    def setNumPyr(self, Npyr: int):
        self._p.n_pyr = int(Npyr)
        self._dirty()

    def setMinDepth(self, minD: float):
        self._p.min_depth = float(minD)
        self._dirty()

    def setMaxDepth(self, maxD: float):
        self._p.max_depth = float(maxD)
        self._dirty()

    def setGrayVariance(self, stdDev: float):     # sets the std-dev despite its name (RPI.h:242-245)
        self._p.sigma_photo = float(stdDev)
        self._dirty()

    def setDepthVariance(self, stdDev: float):    # RPI.h:248-251
        self._p.sigma_depth = float(stdDev)
        self._dirty()

    def useSaliency(self, flag: bool, thresSaliency: float = 0.01):
        # bUseSalientPixels selects calcGradientXY_saliency's index list (RPI.h:401-425), which only the pinhole error pass reads
        # (RPI.h:590-690); the spherical passes' saliency branch is commented out (RPI.h:2568-2642): no effect there.
        self._use_saliency = (bool(flag), float(thresSaliency))
        if self._h is not None:
            self._check(self._L.rgbd360_use_saliency(self._h, int(bool(flag)), float(thresSaliency)))

    def setVisualization(self, viz: bool):
        if viz:
            raise Rgbd360Error("visualisation windows are not part of the MI355X path")

    def setMaskSeams(self, flag: bool):
        self._p.mask_seams = 1 if flag else 0
        self._dirty()

    # ---- lifecycle
    def _ctx(self):
        if self._h is None:
            h = C.c_void_p()
            rc = self._L.rgbd360_create(C.byref(self._p), C.byref(h))
            if rc != 0:
                raise Rgbd360Error(f"rgbd360_create failed ({rc}): no usable HIP device; there is no CPU fallback")
            self._h = h
            if getattr(self, "_cam", None) is not None:          # the camera matrix survives the setters that recreate the context
                self._check(self._L.rgbd360_set_camera(self._h, *self._cam))
            if getattr(self, "_use_saliency", None):
                self._check(self._L.rgbd360_use_saliency(self._h, int(self._use_saliency[0]), self._use_saliency[1]))
            if getattr(self, "_index_arithmetic", 0):
                self._check(self._L.rgbd360_set_index_arithmetic(self._h, int(self._index_arithmetic)))
        return self._h

    def close(self):
        self._dirty()

    def __del__(self):
        try:
            self._dirty()
        except Exception:
            pass

    def _check(self, rc: int, allow=(0,)):
        if rc not in allow:
            msg = self._L.rgbd360_last_error(self._h).decode() if self._h else ""
            raise Rgbd360Error(f"rgbd360 call failed ({rc}): {msg}")
        return rc

    # ---- frames (RPI.h:480-516)
    def _set(self, fn, imgRGB, imgDepth):
        rgb = np.asarray(imgRGB)
        if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
            raise Rgbd360Error("imgRGB must be HxWx3 uint8 (CV_8UC3)")
        if rgb.strides[2] != 1 or rgb.strides[1] != 3:
            rgb = np.ascontiguousarray(rgb)
        d = np.asarray(imgDepth)
        if d.dtype == np.uint16:
            dt = 0
        elif d.dtype == np.float32:
            dt = 1
        else:
            raise Rgbd360Error("imgDepth must be uint16 millimetres (CV_16UC1) or float32 metres (CV_32FC1)")
        if d.ndim != 2 or d.shape != rgb.shape[:2]:
            raise Rgbd360Error("imgDepth must be HxW and match imgRGB")
        if d.strides[1] != d.itemsize:
            d = np.ascontiguousarray(d)
        self._check(fn(self._ctx(), _ptr(rgb), rgb.strides[0], _ptr(d), d.strides[0], dt, d.shape[0], d.shape[1]))

    def setTargetFrame(self, imgRGB, imgDepth):
        self._set(self._L.rgbd360_set_target, imgRGB, imgDepth)

    def setSourceFrame(self, imgRGB, imgDepth):
        self._set(self._L.rgbd360_set_source, imgRGB, imgDepth)

    def setTargetFrameDev(self, rgb_ptr: int, rgb_step: int, depth_ptr: int, depth_step: int, depth_type: int, rows: int,
                          cols: int):
        """Images already in HBM on this context's device (raw device pointers)."""
        self._check(self._L.rgbd360_set_target_dev(self._ctx(), C.c_void_p(rgb_ptr), rgb_step, C.c_void_p(depth_ptr),
                                                   depth_step, depth_type, rows, cols))

    def setSourceFrameDev(self, rgb_ptr: int, rgb_step: int, depth_ptr: int, depth_step: int, depth_type: int, rows: int,
                          cols: int):
        self._check(self._L.rgbd360_set_source_dev(self._ctx(), C.c_void_p(rgb_ptr), rgb_step, C.c_void_p(depth_ptr),
                                                   depth_step, depth_type, rows, cols))

    def promoteSourceToTarget(self):
        self._check(self._L.rgbd360_promote_source_to_target(self._ctx()))

    # ---- alignment (RPI.h:4519-4784)
    def alignFrames360(self, pose_guess=None, method: int = 0, occlusion: int = 0):
        g = pose_to_cm(np.eye(4) if pose_guess is None else pose_guess)
        out = np.zeros(16, dtype=np.float32)
        rc = self._L.rgbd360_align360(self._ctx(), _ptr(g), int(method), int(occlusion), _ptr(out), C.byref(self._res))
        return self._take_result(rc, out)

    def alignFrames360_begin(self, pose_guess=None, method: int = 0, occlusion: int = 0):
        """Enqueue the alignment and return at once (several contexts can then be in flight on one GPU)."""
        g = pose_to_cm(np.eye(4) if pose_guess is None else pose_guess)
        self._check(self._L.rgbd360_align360_begin(self._ctx(), _ptr(g), int(method), int(occlusion)))

    def alignFrames360_finish(self):
        out = np.zeros(16, dtype=np.float32)
        rc = self._L.rgbd360_align360_finish(self._ctx(), _ptr(out), C.byref(self._res))
        return self._take_result(rc, out)

    def _take_result(self, rc, out):
        self._check(rc, allow=(0, 1, 2))
        self.status = rc
        self._pose = pose_from_cm(out)
        r = self._res
        self.SSO = float(r.sso)
        self.avResidual = 0.0 if rc == 1 else float(r.err_final)     # ill-posed: avResidual = 0 (RPI.h:4688)
        self.avPhotoResidual = float(r.rms_photo)
        self.avDepthResidual = float(r.rms_depth)
        self.num_iterations = [int(r.iters[l]) for l in range(self._p.n_pyr)]
        return rc

    # ---- pinhole single-sensor path (RPI.h:254-257, 4254-4512)
    def setCameraMatrix(self, camMat):
        """3x3 intrinsic matrix (or (fx, fy, ox, oy)) of the full-resolution sensor image."""
        K = np.asarray(camMat, np.float64)
        fx, fy, ox, oy = (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) if K.shape == (3, 3) else K.reshape(4)
        self._cam = (float(fx), float(fy), float(ox), float(oy))
        if self._h is not None:
            self._check(self._L.rgbd360_set_camera(self._h, *self._cam))

    def alignFrames(self, pose_guess=None, method: int = 0, occlusion: int = 0):
        g = pose_to_cm(np.eye(4) if pose_guess is None else pose_guess)
        out = np.zeros(16, dtype=np.float32)
        rc = self._L.rgbd360_align_pinhole(self._ctx(), _ptr(g), int(method), int(occlusion), _ptr(out), C.byref(self._res))
        self._check(rc, allow=(0, 1, 2))
        self.status = rc
        self._pose = pose_from_cm(out)
        r = self._res
        self.SSO = float(r.sso)          # only calcHessGrad_Occ2 sets it on this path (RPI.h:2016)
        self.avResidual = float(r.err_final)
        self.avPhotoResidual = float(r.rms_photo)
        self.avDepthResidual = float(r.rms_depth)
        self.num_iterations = [int(r.iters[l]) for l in range(self._p.n_pyr)]
        return rc

    def eval_pinhole(self, level: int, pose, method: int, occlusion: int = 0):
        p = pose_to_cm(pose)
        nrows = C.c_longlong()
        e2s = np.zeros(2, np.float64)
        ns = np.zeros(2, np.int64)
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        self._check(self._L.rgbd360_eval_pinhole_occ(self._ctx(), level, _ptr(p), method, int(occlusion), _ptr(e2s), _ptr(ns), _ptr(H), _ptr(g),
                                                     _ptr(Hd), _ptr(gd), C.byref(nrows)))
        return dict(err2_split=e2s, n_split=ns, H=H.reshape(6, 6).T.copy(), g=g, H64=Hd.reshape(6, 6).T.copy(), g64=gd,
                    n_rows=nrows.value)

    def warp_indices_pinhole(self, level: int, pose) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 2), dtype=np.int32)
        self._check(self._L.rgbd360_warp_indices_pinhole(self._ctx(), level, _ptr(pose_to_cm(pose)), _ptr(out)))
        return out

    def alignSequence(self, frames, method: int = 0, occlusion: int = 0, pose_guess=None, n_inflight: int = 32):
        """rgbd360_align360_batch: the len(frames)-1 consecutive pairs of a sequence (frame j = target, j+1 = source) on this
        context's GPU, n_inflight pairs in flight (slots of the lock-step sequence engine).  frames: list of (rgb HxWx3 uint8, depth HxW uint16 mm | float32 m).
        Returns (poses [n,4,4] float32, status [n] int32, iters [n, n_pyr] int32)."""
        from ._lib import Result
        n = len(frames) - 1
        poses = np.zeros((max(n, 0), 4, 4), np.float32)
        status = np.zeros(max(n, 0), np.int32)
        iters = np.zeros((max(n, 0), self._p.n_pyr), np.int32)
        if n <= 0:
            return poses, status, iters
        rgbs = [np.ascontiguousarray(f[0], np.uint8) for f in frames]
        deps = [np.ascontiguousarray(f[1]) for f in frames]
        shape, dtype = deps[0].shape, deps[0].dtype
        if dtype not in (np.uint16, np.float32):
            raise Rgbd360Error("imgDepth must be uint16 millimetres (CV_16UC1) or float32 metres (CV_32FC1)")
        for r, d in zip(rgbs, deps):
            if r.shape != shape + (3,) or d.shape != shape or d.dtype != dtype:
                raise Rgbd360Error("all frames of a sequence must share one size and depth type")
        rgb_ptrs = (C.c_void_p * len(frames))(*[r.ctypes.data for r in rgbs])
        dep_ptrs = (C.c_void_p * len(frames))(*[d.ctypes.data for d in deps])
        out = np.zeros(n * 16, np.float32)
        res = (Result * n)()
        g = None if pose_guess is None else _ptr(pose_to_cm(pose_guess))
        self._check(self._L.rgbd360_align360_batch(self._ctx(), len(frames), rgb_ptrs, shape[1] * 3, dep_ptrs,
                                                   shape[1] * dtype.itemsize, 0 if dtype == np.uint16 else 1, shape[0], shape[1],
                                                   g, int(method), int(occlusion), int(n_inflight), _ptr(out), res))
        for j in range(n):
            poses[j] = pose_from_cm(out[16 * j:16 * j + 16])
            status[j] = res[j].status
            iters[j] = [int(res[j].iters[l]) for l in range(self._p.n_pyr)]
        return poses, status, iters

    def alignSequenceDev(self, rgb_ptrs, depth_ptrs, rows: int, cols: int, depth_type: int, method: int = 0, occlusion: int = 0,
                         pose_guess=None, n_inflight: int = 32, rgb_step: int = 0, depth_step: int = 0):
        """rgbd360_align360_batch_dev: like alignSequence with every frame already in HBM on this context's device.
        rgb_ptrs / depth_ptrs: raw device pointers (e.g. torch_tensor.data_ptr()), row-major, steps in bytes (0 = packed)."""
        from ._lib import Result
        n = len(rgb_ptrs) - 1
        poses = np.zeros((max(n, 0), 4, 4), np.float32)
        status = np.zeros(max(n, 0), np.int32)
        iters = np.zeros((max(n, 0), self._p.n_pyr), np.int32)
        if n <= 0:
            return poses, status, iters
        if len(depth_ptrs) != len(rgb_ptrs):
            raise Rgbd360Error("rgb_ptrs and depth_ptrs must have the same length")
        rp = (C.c_void_p * len(rgb_ptrs))(*[int(x) for x in rgb_ptrs])
        dp = (C.c_void_p * len(depth_ptrs))(*[int(x) for x in depth_ptrs])
        out = np.zeros(n * 16, np.float32)
        res = (Result * n)()
        g = None if pose_guess is None else _ptr(pose_to_cm(pose_guess))
        self._check(self._L.rgbd360_align360_batch_dev(self._ctx(), len(rgb_ptrs), rp, rgb_step or cols * 3, dp,
                                                       depth_step or cols * (2 if depth_type == 0 else 4), int(depth_type), rows, cols,
                                                       g, int(method), int(occlusion), int(n_inflight), _ptr(out), res))
        for j in range(n):
            poses[j] = pose_from_cm(out[16 * j:16 * j + 16])
            status[j] = res[j].status
            iters[j] = [int(res[j].iters[l]) for l in range(self._p.n_pyr)]
        return poses, status, iters

    def getOptimalPose(self) -> np.ndarray:       # RPI.h:273
        return self._pose.copy()

    def getHessian(self) -> np.ndarray:           # RPI.h:279
        return np.asarray(list(self._res.hessian), dtype=np.float32).reshape(6, 6).T.copy()

    def getGradient(self) -> np.ndarray:          # RPI.h:285
        return np.asarray(list(self._res.gradient), dtype=np.float32)

    def debug_set_schedule(self, fused_solve: bool = True, fused_occ: bool = True):
        """rgbd360_debug_set_schedule (test hook): the two-launch / three-launch schedules the parity tests compare with the fused ones."""
        self._check(self._L.rgbd360_debug_set_schedule(self._ctx(), int(fused_solve), int(fused_occ)))

    def debug_set_sequence_route(self, contexts: bool, max_contexts: int = 0):
        """rgbd360_debug_set_sequence_route (test hook): alignSequence over the per-context route instead of the lock-step engines."""
        self._check(self._L.rgbd360_debug_set_sequence_route(self._ctx(), 1 if contexts else 0, int(max_contexts)))

    def calcEntropy(self) -> float:               # RPI.h:4789-4797 (OdometryRGBD360.cpp:207)
        """0.5 (6 (1 + ln 2 pi) + ln det H^-1) of the last alignment's Hessian; float64 log-determinant (the reference's float
        determinant underflows for full-size Hessians), NaN when H is singular."""
        import math
        sign, logdet = np.linalg.slogdet(self.getHessian().astype(np.float64))
        if sign <= 0:
            return float("nan")
        return float(0.5 * (6.0 * (1.0 + math.log(2 * 3.14159265359)) - logdet))      # PI: Miscellaneous.h:44

    # ---- stage-level access (tests / measurement)
    def level_dims(self, level: int):
        r, c = C.c_int(), C.c_int()
        self._check(self._L.rgbd360_level_dims(self._ctx(), level, C.byref(r), C.byref(c)))
        return r.value, c.value

    _PLANES = {"gray_src": 0, "gray_trg": 1, "depth_src": 2, "depth_trg": 3, "gx": 4, "gy": 5, "dgx": 6, "dgy": 7}

    def plane(self, which: str, level: int) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r, c), dtype=np.float32)
        self._check(self._L.rgbd360_get_plane(self._ctx(), self._PLANES[which], level, _ptr(out)))
        return out

    def lut(self, level: int) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 3), dtype=np.float32)
        self._check(self._L.rgbd360_get_lut(self._ctx(), level, _ptr(out)))
        return out

    def eval(self, level: int, pose, method: int, occlusion: int = 0):
        p = pose_to_cm(pose)
        e2, nv, nvis = C.c_double(), C.c_longlong(), C.c_longlong()
        e2s = np.zeros(2, np.float64)
        ns = np.zeros(2, np.int64)
        H = np.zeros(36, np.float32)
        g = np.zeros(6, np.float32)
        Hd = np.zeros(36, np.float64)
        gd = np.zeros(6, np.float64)
        self._check(self._L.rgbd360_eval_occ(self._ctx(), level, _ptr(p), method, int(occlusion), C.byref(e2), C.byref(nv),
                                             _ptr(e2s), _ptr(ns), _ptr(H), _ptr(g), _ptr(Hd), _ptr(gd), C.byref(nvis)))
        return dict(err2=e2.value, n_valid=nv.value, err2_split=e2s, n_split=ns, H=H.reshape(6, 6).T.copy(), g=g,
                    H64=Hd.reshape(6, 6).T.copy(), g64=gd, n_visible=nvis.value)

    def warp_indices(self, level: int, pose) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r * c, 2), dtype=np.int32)
        self._check(self._L.rgbd360_warp_indices(self._ctx(), level, _ptr(pose_to_cm(pose)), _ptr(out)))
        return out

    def gn_step(self, H, g, lam: float, pose):
        Hc = np.ascontiguousarray(np.asarray(H, np.float32).reshape(6, 6).T.reshape(36))
        gc = np.ascontiguousarray(np.asarray(g, np.float32))
        out = np.zeros(16, np.float32)
        upd = np.zeros(6, np.float32)
        rc = self._L.rgbd360_gn_step(self._ctx(), _ptr(Hc), _ptr(gc), float(lam), _ptr(pose_to_cm(pose)), _ptr(out),
                                     _ptr(upd))
        self._check(rc, allow=(0, 1))
        return rc, pose_from_cm(out), upd

    def forced_iters(self, level: int, pose0, method: int, n_iters: int, want_elapsed: bool = True):
        """want_elapsed=False: no HIP events around the iterations (elapsed_ms is None); the caller times the call itself."""
        out = np.zeros(16, np.float32)
        rms, ms = C.c_double(), C.c_float()
        rc = self._L.rgbd360_forced_iters(self._ctx(), level, _ptr(pose_to_cm(pose0)), method, n_iters, _ptr(out),
                                          C.byref(rms), C.byref(ms) if want_elapsed else None)
        self._check(rc, allow=(0, 1, 2))
        return dict(status=rc, pose=pose_from_cm(out), rms=rms.value, elapsed_ms=ms.value if want_elapsed else None)

    def debug_solve_partials(self, level: int, row, method: int = 0, fused: bool = True):
        """rgbd360_debug_solve_partials: one solve on a hand-made partial table (row: 32 float64 -- 21 upper-triangle terms of H, 6 of
        g, err2 photo / depth, n photo / depth / visible) at the identity pose, through k_solve (fused=False) or the fused launch
        (fused=2: the row late in the table and too small a host bound on the pending rows, see rgbd360_hip_diag.h)."""
        row = np.ascontiguousarray(row, np.float64)
        assert row.shape == (32,)
        out_i = np.zeros(6, np.int32)
        cand, upd = np.zeros(16, np.float32), np.zeros(6, np.float32)
        self._check(self._L.rgbd360_debug_solve_partials(self._ctx(), level, _ptr(row), method, int(fused), _ptr(out_i), _ptr(cand), _ptr(upd)))
        return dict(status=int(out_i[0]), done=int(out_i[1]), level_active=int(out_i[2]), it=int(out_i[3]), n_evals=int(out_i[4]),
                    pend_nb=int(out_i[5]), cand=pose_from_cm(cand), update=upd)

    def forced_iters_call(self, level: int, pose0, method: int, n_iters: int):
        """A closure that runs rgbd360_forced_iters with every argument converted ONCE (for timed loops: the numpy / ctypes
        conversions of forced_iters cost 10-15 us per call, as much as a Gauss-Newton iteration).  call() -> status;
        call.pose_cm is the float32[16] column-major pose the last call wrote (pose_from_cm turns it into 4x4)."""
        fn, ctx = self._L.rgbd360_forced_iters, self._ctx()
        p0 = pose_to_cm(pose0)
        out = np.zeros(16, np.float32)
        rms = C.c_double()
        a_p0, a_out, a_rms = _ptr(p0), _ptr(out), C.byref(rms)

        def call():
            rc = fn(ctx, level, a_p0, method, n_iters, a_out, a_rms, None)
            if rc < 0:
                self._check(rc)
            return rc
        call.pose_cm, call._keep = out, (p0, rms)
        return call

    def forced_iters_batch(self, n_pairs: int, trg, src, level: int, pose0, method: int, n_iters: int):
        """rgbd360_forced_iters_batch: n_pairs copies of the pair (trg, src) = ((rgb, depth), (rgb, depth)) iterate in lock step."""
        rgbT, dT = np.ascontiguousarray(trg[0], np.uint8), np.ascontiguousarray(trg[1])
        rgbS, dS = np.ascontiguousarray(src[0], np.uint8), np.ascontiguousarray(src[1])
        if dT.dtype not in (np.uint16, np.float32) or dS.dtype != dT.dtype or rgbS.shape != rgbT.shape or dS.shape != dT.shape:
            raise Rgbd360Error("both frames must share size and depth type (uint16 mm or float32 m)")
        out = np.zeros(16 * n_pairs, np.float32)
        ms, pus = C.c_float(), C.c_float()
        rc = self._L.rgbd360_forced_iters_batch(self._ctx(), int(n_pairs), _ptr(rgbT), _ptr(dT), _ptr(rgbS), _ptr(dS), rgbT.strides[0], dT.strides[0],
                                                0 if dT.dtype == np.uint16 else 1, dT.shape[0], dT.shape[1], level, _ptr(pose_to_cm(pose0)), method,
                                                n_iters, _ptr(out), C.byref(ms), C.byref(pus))
        self._check(rc, allow=(0, 1, 2))
        return dict(status=rc, poses=np.stack([pose_from_cm(out[16 * k:16 * k + 16]) for k in range(n_pairs)]), elapsed_ms=ms.value,
                    pass_avg_us=pus.value)

    def time_eval_kernel(self, level: int, pose, method: int, want_hg: bool = True, reps: int = 20) -> float:
        us = C.c_float()
        self._check(self._L.rgbd360_time_eval_kernel(self._ctx(), level, _ptr(pose_to_cm(pose)), method, int(want_hg), reps,
                                                     C.byref(us)))
        return float(us.value)

    @staticmethod
    def time_eval_kernel_rotating(regs, level: int, pose, method: int, want_hg: bool = True, reps: int = 20) -> float:
        """Average launch duration with the launches rotating over the contexts `regs` (one device, one pair each): HBM-fed once
        their working sets together exceed the Infinity Cache (rgbd360_hip_diag.h)."""
        arr = (C.c_void_p * len(regs))(*[r._ctx() for r in regs])
        us = C.c_float()
        regs[0]._check(regs[0]._L.rgbd360_time_eval_kernel_rotating(arr, len(regs), level, _ptr(pose_to_cm(pose)), method, int(want_hg),
                                                                    reps, C.byref(us)))
        return float(us.value)

    def time_solve_kernel(self, level: int, mode: int = 0, reps: int = 20) -> float:
        us = C.c_float()
        self._check(self._L.rgbd360_time_solve_kernel(self._ctx(), level, mode, reps, C.byref(us)))
        return float(us.value)

    def set_index_arithmetic(self, mode: int):
        """rgbd360_set_index_arithmetic: 0 = the device definition of the spherical warp (default), 1 = the reference's libm arithmetic
        (asinf / atan2f / roundf restated operation for operation: target indices bit-equal to the reference's own)."""
        self._index_arithmetic = int(mode)
        if self._h is not None:
            self._check(self._L.rgbd360_set_index_arithmetic(self._h, int(mode)))

    def selftest_libm(self, first_bits: int, count: int):
        """(asinf, atanf, roundf, atan2f) mismatches of csrc/libm_f32.h on the device against this process's C library."""
        out = np.zeros(4, dtype=np.uint64)
        self._check(self._L.rgbd360_selftest_libm(self._ctx(), first_bits, count, _ptr(out)))
        return tuple(int(v) for v in out)

    def selftest_math(self, first_bits: int, count: int):
        out = np.zeros(3, dtype=np.uint64)
        self._check(self._L.rgbd360_selftest_math(self._ctx(), first_bits, count, _ptr(out)))
        return int(out[0]), int(out[1]), int(out[2])

    def sync(self):
        self._check(self._L.rgbd360_sync(self._ctx()))

    def sphere_cloud(self, depth, convention: int = 2) -> np.ndarray:
        d = np.ascontiguousarray(depth)
        dt = 0 if d.dtype == np.uint16 else 1
        if dt == 1:
            d = np.ascontiguousarray(d, dtype=np.float32)
        out = np.empty((d.shape[0] * d.shape[1], 3), dtype=np.float32)
        self._check(self._L.rgbd360_sphere_cloud(self._ctx(), _ptr(d), d.strides[0], dt, d.shape[0], d.shape[1], convention,
                                                 _ptr(out)))
        return out


class DepthModel:
    """rgbd360_depth_model_*: one sensor's intrinsic depth model (clams::DiscreteDepthDistortionModel as Calib360::loadIntrinsicCalibration
    loads it, Calib360.h:104-119) and Frame360::undistort's use of it (Frame360.h:293-311).  Host only: no device context."""

    def __init__(self, path: str, downsample: int = 2):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgbd360_depth_model_load(str(path).encode(), int(downsample), C.byref(h))
        if rc != 0:
            raise Rgbd360Error("rgbd360_depth_model_load(%s): %s" % (path, {1: "cannot open", 2: "not a model file, truncated, or bins not divisible"}.get(rc, "bad arguments")))
        self._h = h
        dims = (C.c_int32 * 6)()
        bd = C.c_double()
        self._L.rgbd360_depth_model_info(self._h, dims, C.byref(bd))
        self.width, self.height, self.bin_width, self.bin_height, self.num_bins_x, self.num_bins_y = [int(x) for x in dims]
        self.bin_depth = float(bd.value)

    def undistort(self, depth_m):
        """The corrected copy of a float32 depth image in metres (0 = no measurement)."""
        z = np.ascontiguousarray(depth_m, np.float32).copy()
        if self._L.rgbd360_depth_model_undistort(self._h, _ptr(z), z.strides[0], z.shape[0], z.shape[1]) != 0:
            raise Rgbd360Error("rgbd360_depth_model_undistort: the image is not of the model's size (%d x %d)" % (self.height, self.width))
        return z

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgbd360_depth_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _planes_to_dicts(arr, n):
    return [dict(centroid=np.array(list(arr[i].centroid), np.float32), normal=np.array(list(arr[i].normal), np.float32),
                 d=float(arr[i].d), curvature=float(arr[i].curvature), count=int(arr[i].count), root=int(arr[i].root),
                 area=float(arr[i].area), elongation=float(arr[i].elongation),
                 ppal_dir=np.array(list(arr[i].ppal_dir), np.float32), area_moment=float(arr[i].area_moment),
                 center_hull=np.array(list(arr[i].center_hull), np.float32), hull_points=int(arr[i].hull_points),
                 color_count=int(arr[i].color_count), color_nrgb=np.array(list(arr[i].color_nrgb), np.float32),
                 color_dev=np.array(list(arr[i].color_dev), np.float32), intensity=float(arr[i].intensity),
                 hist_h=np.array(list(arr[i].hist_h), np.float32),
                 hull=np.array([list(arr[i].hull[v]) for v in range(max(0, min(int(arr[i].hull_n), 64)))], np.float32).reshape(-1, 3),
                 color_mode_count=int(arr[i].color_mode_count), color_mode=np.array(list(arr[i].color_mode), np.float32),
                 intensity_mode=float(arr[i].intensity_mode), color_concentration=float(arr[i].color_concentration))
            for i in range(n)]


class Frame360Stages:
    """The per-pixel Frame360 stages on the device: buildSphereCloud (Frame360.h:555-612), the PCL normal map and planar
    region extraction of getPlanesSensor / getPlanesStereo (Frame360.h:949-996, Frame360_stereo.h:854-900).  Defaults are
    the reference's settings for the spherical (stereo) variant."""

    def __init__(self, reg: RegisterPhotoICP | None = None):
        self._reg = reg or RegisterPhotoICP()
        self._L = self._reg._L

    def set_refinement(self, enabled: bool, distance_threshold: float = 0.02):
        """rgbd360_set_plane_refinement: segmentAndRefine's refinement (Frame360.h:977) for the later plane calls of this context."""
        self._refine = (bool(enabled), float(distance_threshold))
        self._reg._check(self._L.rgbd360_set_plane_refinement(self._reg._ctx(), int(enabled), float(distance_threshold)))

    def set_color_image(self, rgb, step: int = 1):
        """rgbd360_set_plane_color_image: the colour image (H x W x 3 uint8, host) that goes with the clouds of the later plane calls
        of this context; cloud pixel (r, c) takes image pixel (r step + step // 2, c step + step // 2).  None clears: planes then
        come back without colour (color_count 0).  Frame360.h:1045-1046 (calcPlaneHistH / calcMainColor2)."""
        if rgb is None:
            self._reg._check(self._L.rgbd360_set_plane_color_image(self._reg._ctx(), None, 0, 0, 0, 1, 0))
            return
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self._reg._check(self._L.rgbd360_set_plane_color_image(self._reg._ctx(), _ptr(rgb), rgb.strides[0], rgb.shape[0], rgb.shape[1], int(step), 0))

    def refinement_stats(self):
        a, b = C.c_int(), C.c_int()
        self._reg._check(self._L.rgbd360_plane_refinement_stats(self._reg._ctx(), C.byref(a), C.byref(b)))
        return dict(pixels_relabelled=a.value, sweeps=b.value)

    def normals(self, xyz, rows, cols, max_depth_change_factor=0.05, normal_smoothing_size=8.0, depth_mode=1):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
        out = np.empty_like(xyz)
        self._reg._check(self._L.rgbd360_normals(self._reg._ctx(), _ptr(xyz), rows, cols, max_depth_change_factor,
                                                 normal_smoothing_size, depth_mode, _ptr(out)))
        return out

    def bilateral_filter(self, xyz, rows, cols, sigma_s=10.0, sigma_r=0.05):
        """rgbd360_bilateral_filter: pcl::FastBilateralFilter as Frame360.h:493-499 configures it; returns the cloud with filtered z."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
        out = np.empty_like(xyz)
        self._reg._check(self._L.rgbd360_bilateral_filter(self._reg._ctx(), _ptr(xyz), rows, cols, sigma_s, sigma_r, _ptr(out)))
        return out

    def sensor_cloud(self, depth_mm, step=2, min_depth=0.3, max_depth=10.0):
        """rgbd360_sensor_cloud: CloudRGBD::getPointCloud + DownsampleRGBD::downsamplePointCloud of one sensor's uint16 mm depth image."""
        d = np.asarray(depth_mm)
        if d.dtype == np.float32:            # metres: the image Frame360::undistort leaves (rgbd360_sensor_cloud_ex, depth_type 1)
            d = np.ascontiguousarray(d)
            rows, cols = d.shape
            out = np.empty(((rows // step) * (cols // step), 3), np.float32)
            self._reg._check(self._L.rgbd360_sensor_cloud_ex(self._reg._ctx(), _ptr(d), d.strides[0], 1, rows, cols, step, min_depth, max_depth, _ptr(out)))
            return out.reshape(rows // step, cols // step, 3)
        if d.dtype != np.uint16 or d.strides[1] != 2:
            d = np.ascontiguousarray(d, np.uint16)
        rows, cols = d.shape
        out = np.empty(((rows // step) * (cols // step), 3), np.float32)
        self._reg._check(self._L.rgbd360_sensor_cloud(self._reg._ctx(), _ptr(d), d.strides[0], rows, cols, step, min_depth, max_depth, _ptr(out)))
        return out.reshape(rows // step, cols // step, 3)

    def sensor_planes(self, depth_mm, step=2, min_depth=0.3, max_depth=10.0, sigma_s=10.0, sigma_r=0.05, max_depth_change_factor=0.02,
                      normal_smoothing_size=8.0, min_inliers=80, angular_threshold=0.0398, distance_threshold=0.02, max_curvature=0.001,
                      Rt=None, max_planes=512):
        """rgbd360_sensor_planes: sensor_cloud + cloud_planes in one call (the cloud never leaves the device)."""
        d = np.asarray(depth_mm)
        if d.dtype != np.uint16 or d.strides[1] != 2:
            d = np.ascontiguousarray(d, np.uint16)
        rows, cols = d.shape
        arr = (_lib.Plane * max_planes)()
        n = C.c_int()
        rt = None if Rt is None else np.ascontiguousarray(np.asarray(Rt, np.float32).T.reshape(16))
        self._reg._check(self._L.rgbd360_sensor_planes(self._reg._ctx(), _ptr(d), d.strides[0], rows, cols, step, min_depth, max_depth, sigma_s,
                                                       sigma_r, max_depth_change_factor, normal_smoothing_size, min_inliers, angular_threshold,
                                                       distance_threshold, max_curvature, None if rt is None else _ptr(rt),
                                                       C.cast(arr, C.c_void_p), max_planes, C.byref(n)))
        return _planes_to_dicts(arr, n.value)

    def cloud_planes(self, xyz, rows, cols, sigma_s=10.0, sigma_r=0.05, max_depth_change_factor=0.02, normal_smoothing_size=8.0,
                     min_inliers=80, angular_threshold=0.0398, distance_threshold=0.02, max_curvature=0.001, depth_mode=0, Rt=None,
                     max_planes=512):
        """rgbd360_cloud_planes: one sensor cloud -> (bilateral filter) -> normal map -> planar regions -> planes in the rig frame
        (defaults = Frame360.h:493-499, 949-977; region curvature filter = PCL's default 0.001, the reference never sets it).  Rt: 4x4 sensor -> rig (row-major numpy) or None."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
        arr = (_lib.Plane * max_planes)()
        n = C.c_int()
        rt = None if Rt is None else np.ascontiguousarray(np.asarray(Rt, np.float32).T.reshape(16))
        self._reg._check(self._L.rgbd360_cloud_planes(self._reg._ctx(), _ptr(xyz), rows, cols, sigma_s, sigma_r, max_depth_change_factor,
                                                      normal_smoothing_size, min_inliers, angular_threshold, distance_threshold,
                                                      max_curvature, depth_mode, None if rt is None else _ptr(rt),
                                                      C.cast(arr, C.c_void_p), max_planes, C.byref(n)))
        return _planes_to_dicts(arr, n.value)

    def distance_map(self, xyz, rows, cols, max_depth_change_factor=0.05, depth_mode=1):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
        out = np.empty((rows, cols), np.float32)
        self._reg._check(self._L.rgbd360_distance_map(self._reg._ctx(), _ptr(xyz), rows, cols, max_depth_change_factor, depth_mode,
                                                      _ptr(out)))
        return out

    def plane_fit(self, xyz, normals, rows, cols, min_inliers=40, angular_threshold=0.05, distance_threshold=0.05,
                  max_curvature=0.001, depth_mode=1, max_planes=256):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(rows * cols, 3)
        normals = np.ascontiguousarray(normals, np.float32).reshape(rows * cols, 3)
        labels = np.empty(rows * cols, np.int32)
        arr = (_lib.Plane * max_planes)()
        n = C.c_int()
        self._reg._check(self._L.rgbd360_plane_fit(self._reg._ctx(), _ptr(xyz), _ptr(normals), rows, cols, min_inliers,
                                                   angular_threshold, distance_threshold, max_curvature, depth_mode, _ptr(labels),
                                                   C.cast(arr, C.c_void_p), max_planes, C.byref(n)))
        return labels.reshape(rows, cols), _planes_to_dicts(arr, n.value)

    def frame_planes(self, depth, convention=2, max_depth_change_factor=0.05, normal_smoothing_size=8.0, min_inliers=40,
                     angular_threshold=0.05, distance_threshold=0.05, max_curvature=0.001, depth_mode=1, max_planes=256):
        d = np.ascontiguousarray(depth)
        dt = 0 if d.dtype == np.uint16 else 1
        if dt == 1:
            d = np.ascontiguousarray(d, np.float32)
        rows, cols = d.shape
        xyz = np.empty((rows * cols, 3), np.float32)
        nrm = np.empty((rows * cols, 3), np.float32)
        labels = np.empty(rows * cols, np.int32)
        arr = (_lib.Plane * max_planes)()
        n = C.c_int()
        self._reg._check(self._L.rgbd360_frame_planes(self._reg._ctx(), _ptr(d), d.strides[0], dt, rows, cols, convention,
                                                      max_depth_change_factor, normal_smoothing_size, min_inliers, angular_threshold,
                                                      distance_threshold, max_curvature, depth_mode, _ptr(xyz), _ptr(nrm), _ptr(labels),
                                                      C.cast(arr, C.c_void_p), max_planes, C.byref(n)))
        return dict(xyz=xyz, normals=nrm, labels=labels.reshape(rows, cols), planes=_planes_to_dicts(arr, n.value))


    def stage_timing(self, on: bool = True):
        """rgbd360_frame_planes_stage_timing: HIP events at the stage boundaries of the later frame_planes calls (measurement)."""
        self._reg._check(self._L.rgbd360_frame_planes_stage_timing(self._reg._ctx(), int(on)))

    def stage_times(self):
        """(a13 cloud + depth-edge mask, a14 distance map + normal map, a15 plane stage) of the last frame_planes call, microseconds."""
        us = (C.c_float * 3)()
        self._reg._check(self._L.rgbd360_frame_planes_stage_times(self._reg._ctx(), us))
        return [float(x) for x in us]

    def frame_planes_dev(self, depth_ptr: int, rows: int, cols: int, depth_type: int = 0, convention=2, max_depth_change_factor=0.05,
                         normal_smoothing_size=8.0, min_inliers=40, angular_threshold=0.05, distance_threshold=0.05,
                         max_curvature=0.001, depth_mode=1, max_planes=256, depth_step: int = 0):
        """rgbd360_frame_planes_dev: depth image already in HBM (raw device pointer), maps stay on the device.
        Returns dict(planes=[...], xyz_ptr, normals_ptr, labels_ptr) -- device pointers valid until the next Frame360 call."""
        arr = (_lib.Plane * max_planes)()
        n = C.c_int()
        px, pn, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        step = depth_step or cols * (2 if depth_type == 0 else 4)
        self._reg._check(self._L.rgbd360_frame_planes_dev(self._reg._ctx(), C.c_void_p(depth_ptr), step, depth_type, rows, cols, convention,
                                                          max_depth_change_factor, normal_smoothing_size, min_inliers, angular_threshold,
                                                          distance_threshold, max_curvature, depth_mode, C.cast(arr, C.c_void_p), max_planes,
                                                          C.byref(n), C.byref(px), C.byref(pn), C.byref(pl)))
        return dict(planes=_planes_to_dicts(arr, n.value), xyz_ptr=px.value, normals_ptr=pn.value, labels_ptr=pl.value)


QVGA_K = (262.5, 262.5, 159.5, 119.5)       # Calib360.h:74-77 (fx, fy, cx, cy)


def load_frame_bin(path: str):
    """Frame360::loadFrame: -> (rgb [8,rows,cols,3] uint8, depth [8,rows,cols] uint16 mm).  Host-only, no GPU needed."""
    L = _lib.load()
    r, c = C.c_int(), C.c_int()
    rc = L.rgbd360_load_frame_bin(path.encode(), None, None, C.byref(r), C.byref(c))
    if rc != 0:
        raise Rgbd360Error(f"rgbd360_load_frame_bin({path}) failed ({rc})")
    rgb = np.empty((8, r.value, c.value, 3), np.uint8)
    depth = np.empty((8, r.value, c.value), np.uint16)
    rc = L.rgbd360_load_frame_bin(path.encode(), _ptr(rgb), _ptr(depth), C.byref(r), C.byref(c))
    if rc != 0:
        raise Rgbd360Error(f"rgbd360_load_frame_bin({path}) failed ({rc})")
    return rgb, depth


def stitch_sphere(reg: RegisterPhotoICP, rgb8, depth8, Rt_inv, K=QVGA_K):
    """Frame360::stitchSphericalImage on the device.  Rt_inv: [8,4,4] (row-major numpy)."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    depth8 = np.ascontiguousarray(depth8, np.uint16)
    _, rows, cols, _ = rgb8.shape
    W = rows * 8
    H = int(W * 0.5 * 60.0 / 180)
    M = np.ascontiguousarray(np.asarray(Rt_inv, np.float32).transpose(0, 2, 1).reshape(8 * 16))    # column-major per sensor
    Kc = np.asarray(K, np.float32)
    out_rgb = np.empty((H, W, 3), np.uint8)
    out_d = np.empty((H, W), np.uint16)
    r, c = C.c_int(), C.c_int()
    reg._check(reg._L.rgbd360_stitch_sphere(reg._ctx(), _ptr(rgb8), _ptr(depth8), rows, cols, _ptr(M), _ptr(Kc), _ptr(out_rgb),
                                            _ptr(out_d), C.byref(r), C.byref(c)))
    assert (r.value, c.value) == (H, W)
    return out_rgb, out_d


def Register(frame_trg, frame_src, pose: np.ndarray, method: int = RegisterPhotoICP.PHOTO_DEPTH, reg=None) -> bool:
    """The north-star convenience shape `Register(Frame360&, Frame360&, Matrix4f&) -> bool`: frames are any objects
    with `sphereRGB` / `sphereDepth` (Frame360.h:104-111); `pose` is the initial guess and is overwritten in place."""
    r = reg or RegisterPhotoICP()
    r.setTargetFrame(frame_trg.sphereRGB, frame_trg.sphereDepth)
    r.setSourceFrame(frame_src.sphereRGB, frame_src.sphereDepth)
    rc = r.alignFrames360(pose, method)
    pose[...] = r.getOptimalPose()
    return rc == 0
