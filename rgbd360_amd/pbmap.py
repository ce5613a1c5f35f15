"""RegisterRGBD360's PbMap side (reference include/RegisterRGBD360.h:47-338) over the C ABI: plane-graph matching of two
frames' planar regions and the closed-form pose of the matched planes -- the initial-guess provider in front of the dense
alignment.  Method names follow the reference class; the work is `rgbd360_register_planes` (host C++
inside the HIP library).  No CPU fallback: `_lib.load()` raises when the library is missing."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

DEFAULT_6DoF, PLANAR_3DoF, ODOMETRY_6DoF, PLANAR_ODOMETRY_3DoF = 0, 1, 2, 3      # RegisterRGBD360.h:258-264
DOF = 6                                                                        # RegisterRGBD360.h:42


def default_params(odometry: bool = False) -> _lib.PbmapParams:
    """configLocaliser_spherical.ini (odometry False) / configLocaliser_sphericalOdometry.ini (True)."""
    p = _lib.PbmapParams()
    _lib.load().rgbd360_pbmap_default_params(C.byref(p), 1 if odometry else 0)
    return p


def planes_to_array(planes):
    """list of plane dicts (Frame360Stages.plane_fit / frame_planes) -> ctypes array of rgbd360_plane."""
    arr = (_lib.Plane * max(len(planes), 1))()
    for i, pl in enumerate(planes):
        a = arr[i]
        for k in range(3):
            a.centroid[k] = float(pl["centroid"][k])
            a.normal[k] = float(pl["normal"][k])
            a.ppal_dir[k] = float(pl.get("ppal_dir", (0, 0, 0))[k])
        a.d = float(pl["d"])
        a.curvature = float(pl.get("curvature", 0.0))
        a.count = int(pl.get("count", 0))
        a.root = int(pl.get("root", i))
        a.area = float(pl["area"])
        a.elongation = float(pl["elongation"])
        # hull stage (device planes carry these; hand-made records may not: hull_points = 0 makes the library fall back to centroid / area)
        a.area_moment = float(pl.get("area_moment", 0.0))
        a.hull_points = int(pl.get("hull_points", 0))
        ch = pl.get("center_hull", pl["centroid"])
        for k in range(3):
            a.center_hull[k] = float(ch[k])
        hull = pl.get("hull")            # the hull polygon ([n <= 64][3]; rgbd360_merge_planes tests proximity on it)
        a.hull_n = 0
        if hull is not None and len(hull) >= 3:
            a.hull_n = min(len(hull), 64)
            for v in range(a.hull_n):
                for k in range(3):
                    a.hull[v][k] = float(hull[v][k])
        # colour descriptors (color_count 0 = none: the matcher skips its colour tests for this plane)
        a.color_count = int(pl.get("color_count", 0))
        if a.color_count > 0:
            for k in range(3):
                a.color_nrgb[k] = float(pl["color_nrgb"][k])
                a.color_dev[k] = float(pl.get("color_dev", (0, 0, 0))[k])
            a.intensity = float(pl.get("intensity", 0.0))
            hh = pl.get("hist_h")
            if hh is not None:
                for k in range(74):
                    a.hist_h[k] = float(hh[k])
            a.color_mode_count = int(pl.get("color_mode_count", 0))        # the dominant colour (0: the matcher compares the means)
            if a.color_mode_count > 0:
                for k in range(3):
                    a.color_mode[k] = float(pl["color_mode"][k])
                a.intensity_mode = float(pl.get("intensity_mode", pl.get("intensity", 0.0)))
                a.color_concentration = float(pl.get("color_concentration", 1.0))
    return arr


def register_planes(ref_planes, trg_planes, max_match_planes: int = 0, regist_mode: int = DEFAULT_6DoF, params=None):
    """rgbd360_register_planes.  Returns dict(status, pose [4,4] f32 (p_ref = R p_trg + t), info [6,6] f32, match
    {ref index: trg index}, area_matched)."""
    L = _lib.load()
    ra, ta = planes_to_array(ref_planes), planes_to_array(trg_planes)
    pose = np.zeros(16, np.float32)
    info = np.zeros(36, np.float32)
    match = np.full(max(len(ref_planes), 1), -1, np.int32)
    nm = C.c_int(0)
    area = C.c_float(0)
    st = L.rgbd360_register_planes(C.cast(ra, C.c_void_p), len(ref_planes), C.cast(ta, C.c_void_p), len(trg_planes),
                                   int(max_match_planes), int(regist_mode), C.byref(params) if params is not None else None,
                                   pose.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p),
                                   match.ctypes.data_as(C.c_void_p), C.byref(nm), C.byref(area))
    if st < 0:
        raise ValueError("rgbd360_register_planes: bad arguments")
    m = {i: int(match[i]) for i in range(len(ref_planes)) if match[i] >= 0}
    assert len(m) == nm.value
    return dict(status=st, pose=pose.reshape(4, 4).T.copy(), info=info.reshape(6, 6).T.copy(), match=m,
                area_matched=float(area.value))


def merge_planes(planes, max_curvature=0.0013, cos_normal=0.99, dist_d=0.45, proximity=0.3, normal_offset=0.06, min_area=0.12,
                 max_elongation=6.0):
    """rgbd360_merge_planes (Frame360::mergePlanes, Frame360.h:655-733, after the size filters of Frame360.h:1034,1041; defaults = the reference's constants).  Returns the merged plane dicts."""
    from .register import _planes_to_dicts
    L = _lib.load()
    arr = planes_to_array(planes)
    out = (_lib.Plane * max(len(planes), 1))()
    n = C.c_int(0)
    rc = L.rgbd360_merge_planes(C.cast(arr, C.c_void_p), len(planes), max_curvature, min_area, max_elongation, cos_normal, dist_d, proximity, normal_offset,
                                C.cast(out, C.c_void_p), len(planes), C.byref(n))
    if rc != 0:
        raise ValueError("rgbd360_merge_planes: bad arguments")
    return _planes_to_dicts(out, n.value)


def pool_sensor_planes(planes, max_curvature=0.0013, min_area=0.12, max_elongation=6.0, cos_normal=0.99, dist_normal=0.05, proximity=0.2):
    """rgbd360_pool_sensor_planes (the tail of Frame360::getPlanesSensor, Frame360.h:1034-1068): one sensor's regions -> local_planes_[sensor]:
    small / narrow regions dropped, flat regions of one surface (isSamePlane(0.99, 0.05, 0.2)) pooled in input order."""
    from .register import _planes_to_dicts
    L = _lib.load()
    arr = planes_to_array(planes)
    out = (_lib.Plane * max(len(planes), 1))()
    n = C.c_int(0)
    rc = L.rgbd360_pool_sensor_planes(C.cast(arr, C.c_void_p), len(planes), max_curvature, min_area, max_elongation, cos_normal, dist_normal, proximity,
                                      C.cast(out, C.c_void_p), len(planes), C.byref(n))
    if rc != 0:
        raise ValueError("rgbd360_pool_sensor_planes: bad arguments")
    return _planes_to_dicts(out, n.value)


def group_planes(planes_per_sensor, max_curvature=0.0013, min_area=0.5, cos_normal=0.99, dist_d=0.45, max_dist_hull=0.5, max_dist_parallel_hull=0.09):
    """rgbd360_group_planes (Frame360::groupPlanes, Frame360.h:741-833; defaults = the reference's constants): the plane lists of the rig's
    sensors (in the rig frame, sensor order) -> the frame's plane list, pieces of one surface seen by neighbouring sensors pooled."""
    from .register import _planes_to_dicts
    L = _lib.load()
    flat = [p for lst in planes_per_sensor for p in lst]
    arr = planes_to_array(flat)
    counts = (C.c_int32 * len(planes_per_sensor))(*[len(lst) for lst in planes_per_sensor])
    out = (_lib.Plane * max(len(flat), 1))()
    n = C.c_int(0)
    rc = L.rgbd360_group_planes(C.cast(arr, C.c_void_p), counts, len(planes_per_sensor), max_curvature, min_area, cos_normal, dist_d, max_dist_hull,
                                max_dist_parallel_hull, C.cast(out, C.c_void_p), len(flat), C.byref(n))
    if rc != 0:
        raise ValueError("rgbd360_group_planes: bad arguments")
    return _planes_to_dicts(out, n.value)


class RegisterRGBD360:
    """Mirror of the reference class (RegisterRGBD360.h:47): setReference / setTarget / RegisterPbMap / getPose /
    getInfoMat / getCovMat / calcEntropy / getMatchedPlanes / getAreaMatched.  A "frame" here is the plane list of a
    Frame360 (Frame360Stages.frame_planes(...)["planes"]) -- the only part of Frame360 this class reads (`planes.vPlanes`)."""

    def __init__(self, odometry_config: bool = False, params=None):        # :97-105 loads the .ini thresholds
        self.params = params if params is not None else default_params(odometry_config)
        self._ref = self._trg = None
        self._max_ref = self._max_trg = 0
        self._done = False
        self.rigidTransf = np.eye(4, dtype=np.float32)
        self.informationM = np.zeros((6, 6), np.float32)
        self.bestMatch = {}
        self.areaMatched = 0.0
        self.areaSource = self.areaTarget = 0.0
        self._mode = DEFAULT_6DoF
        self._good = False

    def setReference(self, ref_planes, max_match_planes: int = 0):          # :110-157
        self._ref, self._max_ref, self._done = ref_planes, max_match_planes, False

    def setTarget(self, trg_planes, max_match_planes: int = 0):             # :163-195
        self._trg, self._max_trg, self._done = trg_planes, max_match_planes, False

    def RegisterPbMap(self, frame1=None, frame2=None, max_match_planes: int = 0, registMode: int = DEFAULT_6DoF) -> bool:
        """:276-338.  True = good alignment."""
        if frame1 is not None:
            self.setReference(frame1, max_match_planes)
        if frame2 is not None:
            self.setTarget(frame2, max_match_planes)
        if self._ref is None or self._trg is None:
            raise RuntimeError("RegisterPbMap: reference and target frames must be set")
        self._mode = registMode
        self._done = True
        r = register_planes(self._ref, self._trg, max(self._max_ref, self._max_trg), registMode, self.params)
        self.bestMatch, self.areaMatched = r["match"], r["area_matched"]
        self._good = r["status"] == 0
        if self._good:
            self.rigidTransf, self.informationM = r["pose"], r["info"]
            p = self.params

            def subgraph_area(planes, max_match):                             # :325-333: area of the planes that entered the matching
                kept = [pl for pl in planes if not pl["area"] < p.min_area_plane and not pl["elongation"] > p.max_elongation_plane]
                areas = [float(pl["area"]) if pl["curvature"] < p.max_curvature_plane else 0.0 for pl in kept]
                if max_match > 0 and len(kept) > max_match:                   # :121-150: the max_match largest areas
                    thr = sorted(areas)[len(kept) - max_match - 1]
                    return sum(a for a in areas if a > thr)
                return sum(a for a, pl in zip(areas, kept) if pl["curvature"] < p.max_curvature_plane)
            mm = max(self._max_ref, self._max_trg)
            self.areaSource, self.areaTarget = subgraph_area(self._ref, mm), subgraph_area(self._trg, mm)
        return self._good

    def _ensure(self):
        if not self._done:
            self.RegisterPbMap(registMode=self._mode)

    def getPose(self):                                                        # :198-204
        self._ensure()
        return self.rigidTransf

    def getInfoMat(self):                                                     # :218-224
        self._ensure()
        return self.informationM

    def getCovMat(self):                                                      # :207-215
        self._ensure()
        return np.linalg.inv(self.informationM.astype(np.float64)).astype(np.float32)

    def calcEntropy(self) -> float:                                           # :229-238
        cov = np.linalg.inv(self.getInfoMat().astype(np.float64))
        return 0.5 * (DOF * (1 + math.log(2 * math.pi)) + math.log(np.linalg.det(cov)))

    def getMatchedPlanes(self):                                               # :241-247
        self._ensure()
        return self.bestMatch

    def getAreaMatched(self) -> float:                                        # :250-256
        self._ensure()
        return self.areaMatched
