"""Wall time of the C call rgbd360_frame_planes_dev alone (no Python conversion of the plane list), next to the kernel time the
rocprofv3 traces give (0.22 ms at 2048x1024): what the host side of the chain costs.  python tools/frame_planes_call_perf.py [W [angular_threshold [refine 0|1 [colour 0|1]]]]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rgbd360_amd import _lib, synth
from rgbd360_amd.register import RegisterPhotoICP

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ANG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
REFINE = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 1: with segmentAndRefine's refinement (rgbd360_set_plane_refinement)
COLOUR = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: with a registered colour image (colour descriptors + dominant colour per plane)
H = W // 2
(rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
reg = RegisterPhotoICP()
L = _lib.load()
if REFINE:
    assert L.rgbd360_set_plane_refinement(reg._ctx(), 1, C.c_float(0.02)) == 0
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
p = C.c_void_p()
dc = np.ascontiguousarray(dA)
assert hip.hipMalloc(C.byref(p), dc.nbytes) == 0 and hip.hipMemcpy(p, dc.ctypes.data_as(C.c_void_p), dc.nbytes, 1) == 0
if COLOUR:
    rgb = np.ascontiguousarray(rgbA)
    assert L.rgbd360_set_plane_color_image(reg._ctx(), rgb.ctypes.data_as(C.c_void_p), C.c_size_t(W * 3), H, W, 1, 0) == 0
arr = (_lib.Plane * 4096)()
n = C.c_int()
px, pn, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()


def call():
    rc = L.rgbd360_frame_planes_dev(reg._ctx(), p, W * 2, 0, H, W, 2, 0.05, 8.0, 40, ANG, 0.05, 0.0013, 1, C.cast(arr, C.c_void_p), 4096,
                                    C.byref(n), C.byref(px), C.byref(pn), C.byref(pl))
    assert rc == 0


for _ in range(5):
    call()
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("rgbd360_frame_planes_dev %dx%d%s (angular threshold %.4f, %d planes): median %.3f ms, best %.3f ms, worst %.3f ms per call" % (
    W, H, " + colour" if COLOUR else "", ANG, n.value, ts[25] * 1e3, ts[0] * 1e3, ts[-1] * 1e3))
