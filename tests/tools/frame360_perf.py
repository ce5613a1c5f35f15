"""Timing + parity of the Frame360 stages at full size: python tools/frame360_perf.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
from oracle import oracle as O
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = W // 2
(rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
st = Frame360Stages(RegisterPhotoICP())
out = st.frame_planes(dA, convention=2, angular_threshold=0.03)
t0 = time.perf_counter()
for _ in range(5): out = st.frame_planes(dA, convention=2, angular_threshold=0.03)
print("frame_planes %dx%d (cloud + normals + regions, host in/out): %.2f ms; planes %d, big %s" % (W, H, (time.perf_counter() - t0) * 200, len(out["planes"]), [(p["count"], np.round(p["normal"], 3).tolist(), round(p["d"], 3)) for p in out["planes"] if p["count"] > 20000]))
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dptr = C.c_void_p()
dA_c = np.ascontiguousarray(dA)
assert hip.hipMalloc(C.byref(dptr), dA_c.nbytes) == 0 and hip.hipMemcpy(dptr, dA_c.ctypes.data_as(C.c_void_p), dA_c.nbytes, 1) == 0
dev = st.frame_planes_dev(dptr.value, H, W, 0 if dA.dtype == np.uint16 else 1, convention=2, angular_threshold=0.03)
t0 = time.perf_counter()
for _ in range(20): dev = st.frame_planes_dev(dptr.value, H, W, 0 if dA.dtype == np.uint16 else 1, convention=2, angular_threshold=0.03)
print("frame_planes_dev %dx%d (depth resident, maps stay in HBM, plane list to the host): %.3f ms; planes %d (host variant %d)" % (
    W, H, (time.perf_counter() - t0) * 50, len(dev["planes"]), len(out["planes"])))
t0 = time.time(); xyz = O.sphere_cloud(dA, 2); nrm, win = O.f360_normals(xyz, H, W, 0.05, 8.0, 1); labels, planes = O.f360_plane_segment(xyz, nrm, H, W, 40, 0.03, 0.05, 0.001, 1); t_cpu = time.time() - t0
ok = ~np.isnan(nrm[:, 0])
print("oracle %.2f s; cloud equal %s; normals nan-equal %s max diff %.2e; labels equal (oracle normals in) %s; planes %d vs %d" % (
    t_cpu, np.array_equal(np.nan_to_num(xyz), np.nan_to_num(out["xyz"])), np.array_equal(np.isnan(out["normals"][:, 0]), ~ok),
    np.abs(out["normals"][ok] - nrm[ok]).max(), np.array_equal(st.plane_fit(xyz, nrm, H, W, 40, 0.03, 0.05, 0.001, 1)[0], labels), len(out["planes"]), len(planes)))
