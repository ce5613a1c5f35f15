"""One process, several MI355X: ctypes mirror of the native multi-GPU sequence entry (rgbd360_multi_*, csrc/multi_gpu.h).

BASELINE.json configs[3] / SURVEY.md 8e: an odometry sequence (pair j = frames j, j+1; OdometryRGBD360.cpp:141-297) is cut into
contiguous shards of pairs, one per device; one host thread per device drives that device's contexts; the solved poses (with
status / iteration counts / H) are all-gathered once with ncclAllGather (RCCL over xGMI) inside the library.  The
process-per-GPU form of the same sharding (torch.distributed launch) lives in rgbd360_amd/batch.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .register import Rgbd360Error, _ptr, pose_from_cm, pose_to_cm


def shard_range(n_items: int, rank: int, world: int):
    lo, hi = C.c_int(), C.c_int()
    _lib.load().rgbd360_shard_range(int(n_items), int(rank), int(world), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class MultiGpuSequence:
    def __init__(self, n_gpus: int = 1, device_ids=None, n_pyr: int = 4, **params):
        self._L = _lib.load()
        self._p = _lib.Params()
        self._L.rgbd360_default_params(C.byref(self._p))
        self._p.n_pyr = int(n_pyr)
        for k, v in params.items():
            setattr(self._p, k, v)
        ids = None
        if device_ids is not None:
            if len(device_ids) != n_gpus:
                raise Rgbd360Error("device_ids must list n_gpus devices")
            ids = (C.c_int * n_gpus)(*[int(d) for d in device_ids])
        h = C.c_void_p()
        rc = self._L.rgbd360_multi_create(C.byref(self._p), int(n_gpus), ids, C.byref(h))
        if rc != 0:
            raise Rgbd360Error(f"rgbd360_multi_create failed ({rc}): needs {n_gpus} HIP device(s) (and RCCL beyond one); no CPU fallback")
        self._h = h
        self.n_gpus = n_gpus
        self._n_pairs = 0

    @property
    def uses_rccl(self) -> bool:
        return bool(self._L.rgbd360_multi_uses_rccl(self._h))

    def set_index_arithmetic(self, mode: int):
        """rgbd360_multi_set_index_arithmetic: 1 = the warp in the reference's libm arithmetic on every device (see RegisterPhotoICP)."""
        rc = self._L.rgbd360_multi_set_index_arithmetic(self._h, int(mode))
        if rc != 0:
            raise Rgbd360Error(f"rgbd360_multi_set_index_arithmetic failed ({rc}): {self._L.rgbd360_multi_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.rgbd360_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise Rgbd360Error(f"{self._L.rgbd360_multi_last_error(self._h).decode()} ({rc})")
        return rc

    @staticmethod
    def _frame_ptrs(frames):
        rgbs = [np.ascontiguousarray(f[0], np.uint8) for f in frames]
        deps = [np.ascontiguousarray(f[1]) for f in frames]
        shape, dtype = deps[0].shape, deps[0].dtype
        if dtype not in (np.uint16, np.float32):
            raise Rgbd360Error("imgDepth must be uint16 millimetres (CV_16UC1) or float32 metres (CV_32FC1)")
        for r, d in zip(rgbs, deps):
            if r.shape != shape + (3,) or d.shape != shape or d.dtype != dtype:
                raise Rgbd360Error("all frames of a sequence must share one size and depth type")
        rp = (C.c_void_p * len(frames))(*[r.ctypes.data for r in rgbs])
        dp = (C.c_void_p * len(frames))(*[d.ctypes.data for d in deps])
        return rgbs, deps, rp, dp, shape, dtype

    def _unpack(self, n, out, res):
        poses = np.zeros((n, 4, 4), np.float32)
        status = np.zeros(n, np.int32)
        iters = np.zeros((n, self._p.n_pyr), np.int32)
        for j in range(n):
            poses[j] = pose_from_cm(out[16 * j:16 * j + 16])
            status[j] = res[j].status
            iters[j] = [int(res[j].iters[l]) for l in range(self._p.n_pyr)]
        return poses, status, iters

    def align_sequence(self, frames, method: int = 2, occlusion: int = 0, pose_guess=None, n_inflight: int = 32):
        """Host frames [(rgb, depth), ...] -> (poses [n,4,4], status [n], iters [n, n_pyr]) for the n = len(frames)-1 pairs."""
        n = len(frames) - 1
        if n <= 0:
            return self._unpack(0, None, None)
        keep = self._frame_ptrs(frames)
        _, _, rp, dp, shape, dtype = keep
        out = np.zeros(n * 16, np.float32)
        res = (_lib.Result * n)()
        g = None if pose_guess is None else _ptr(pose_to_cm(pose_guess))
        self._check(self._L.rgbd360_multi_align_sequence(self._h, len(frames), rp, shape[1] * 3, dp, shape[1] * dtype.itemsize,
                                                         0 if dtype == np.uint16 else 1, shape[0], shape[1], g, int(method),
                                                         int(occlusion), int(n_inflight), _ptr(out), res))
        return self._unpack(n, out, res)

    def load_sequence(self, frames):
        """Copies every device's frames into its HBM (the resident variant's set-up; not part of any timed region)."""
        keep = self._frame_ptrs(frames)
        _, _, rp, dp, shape, dtype = keep
        self._check(self._L.rgbd360_multi_load_sequence(self._h, len(frames), rp, shape[1] * 3, dp, shape[1] * dtype.itemsize,
                                                        0 if dtype == np.uint16 else 1, shape[0], shape[1]))
        self._n_pairs = len(frames) - 1

    def align_resident(self, method: int = 2, occlusion: int = 0, pose_guess=None, n_inflight: int = 32):
        n = self._n_pairs
        if n <= 0:
            raise Rgbd360Error("load_sequence must be called first")
        out = np.zeros(n * 16, np.float32)
        res = (_lib.Result * n)()
        g = None if pose_guess is None else _ptr(pose_to_cm(pose_guess))
        self._check(self._L.rgbd360_multi_align_resident(self._h, g, int(method), int(occlusion), int(n_inflight), _ptr(out), res))
        return self._unpack(n, out, res)
