"""Timing of the keyframe-link sequence at full size (KFsphere_SLAM.cpp:129-163 shape): planes of both frames on the device
(rgbd360_frame_planes_dev) -> RegisterPbMap (host matcher + pose) -> dense alignment seeded with the plane pose.
python tools/pbmap_perf.py [W] [trans_m] [rot_deg]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rgbd360_amd import pbmap, synth
from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
trans = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rot = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
H = W // 2
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=5, trans=trans, rot_deg=rot)
reg = RegisterPhotoICP()
reg.setNumPyr(4)
st = Frame360Stages(reg)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
ptrs = []
for d in (dA, dB):
    p = C.c_void_p()
    dc = np.ascontiguousarray(d)
    assert hip.hipMalloc(C.byref(p), dc.nbytes) == 0 and hip.hipMemcpy(p, dc.ctypes.data_as(C.c_void_p), dc.nbytes, 1) == 0
    ptrs.append(p)


# PlaneCoefficientComparator links neighbouring pixels whose (window-averaged) normals differ by less than the angular
# threshold.  Along an image row that crosses a wall / ceiling edge at a shallow angle the blended normal turns by
# (90 degrees / 8-pixel window) x (slope of the edge in the image) per pixel -- 0.7-1.3 degrees per pixel at 2048x1024 -- so the
# 0.03 rad that separates the walls at 512x256 merges the whole room into one (curved, rejected) region at full size: the
# threshold has to shrink with the pixel pitch.
ANG = float(os.environ.get("PBMAP_ANG", 0.03 * 1024 / W))
MIN_INLIERS = int(os.environ.get("PBMAP_MIN_INLIERS", 40 * (W // 512) ** 2))      # the same solid angle at every resolution


def planes_of(p):
    return st.frame_planes_dev(p.value, H, W, 0, convention=2, angular_threshold=ANG, min_inliers=MIN_INLIERS, max_curvature=0.0013,
                               max_planes=2048)["planes"]


registerer = pbmap.RegisterRGBD360(odometry_config=True)
pa, pb = planes_of(ptrs[0]), planes_of(ptrs[1])
n = 20
t0 = time.perf_counter()
for _ in range(n):
    pa, pb = planes_of(ptrs[0]), planes_of(ptrs[1])
t_planes = (time.perf_counter() - t0) / n
arrs = (pbmap.planes_to_array(pa), pbmap.planes_to_array(pb))
L = pbmap._lib.load()
pose = np.zeros(16, np.float32)
info = np.zeros(36, np.float32)
match = np.full(len(pa), -1, np.int32)
nm, area = C.c_int(0), C.c_float(0)
t0 = time.perf_counter()
for _ in range(200):
    st_pb = L.rgbd360_register_planes(C.cast(arrs[0], C.c_void_p), len(pa), C.cast(arrs[1], C.c_void_p), len(pb), 25, pbmap.ODOMETRY_6DoF,
                                      C.byref(registerer.params), pose.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p),
                                      match.ctypes.data_as(C.c_void_p), C.byref(nm), C.byref(area))
t_match = (time.perf_counter() - t0) / 200
good = registerer.RegisterPbMap(pa, pb, 25, pbmap.ODOMETRY_6DoF)
guess = registerer.getPose()
reg.setTargetFrame(rgbA, dA)
reg.setSourceFrame(rgbB, dB)
out = {}
for name, g in (("identity", np.eye(4)), ("pbmap", guess)):
    reg.alignFrames360(g, 2)
    t0 = time.perf_counter()
    for _ in range(n):
        rc = reg.alignFrames360(g, 2)
    out[name] = ((time.perf_counter() - t0) / n, rc, list(reg.num_iterations), synth.pose_error(reg.getOptimalPose(), T))
print("%dx%d, motion %.2f m / %.1f deg, angular threshold %.4f rad, min_inliers %d: planes of two frames (device, plane lists to the host) %.3f ms; %d / %d planes" % (
    W, H, trans, rot, ANG, MIN_INLIERS, t_planes * 1e3, len(pa), len(pb)))
print("RegisterPbMap (host, ODOMETRY_6DoF, max 25): %.1f us, status %d, %d matched, pose error vs truth %.2e rad %.2e m; entropy %.2f" % (
    t_match * 1e6, st_pb, nm.value, *synth.pose_error(guess, T), registerer.calcEntropy() if good else float("nan")))
for name, (t, rc, iters, err) in out.items():
    print("dense PHOTO_DEPTH alignment from %-8s: %.3f ms, status %d, iterations %s, error vs truth %.2e rad %.2e m" % (name, t * 1e3, rc, iters, *err))
