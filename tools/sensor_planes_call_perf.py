"""Wall time of the C call rgbd360_sensor_planes alone (one 320x240 sensor image -> 160x120 cloud -> filter, normals, regions, planes in the
rig frame), plane array allocated once: python tools/sensor_planes_call_perf.py [refine 0|1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import _lib, synth
from rgbd360_amd.register import RegisterPhotoICP
L = _lib.load()
reg = RegisterPhotoICP()
(rgbA, dA), _, _, K = synth.make_pinhole_pair(320, 240, seed=7)
d = np.ascontiguousarray(dA, np.uint16)
arr = (_lib.Plane * 512)()
n = C.c_int()
def call(sig_s):
    rc = L.rgbd360_sensor_planes(reg._ctx(), d.ctypes.data_as(C.c_void_p), d.strides[0], 240, 320, 2, C.c_float(0.3), C.c_float(10.0), C.c_float(sig_s), C.c_float(0.05),
                                 C.c_float(0.02), C.c_float(8.0), 80, C.c_float(0.0398), C.c_float(0.02), C.c_float(0.001), None, C.cast(arr, C.c_void_p), 512, C.byref(n))
    assert rc == 0, rc
REFINE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if REFINE: assert L.rgbd360_set_plane_refinement(reg._ctx(), 1, C.c_float(0.02)) == 0
for sig_s, name in ((10.0, "with the bilateral filter" + (" + refinement" if REFINE else "")), (0.0, "without the filter" + (" + refinement" if REFINE else ""))):
    for _ in range(5): call(sig_s)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); call(sig_s); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("rgbd360_sensor_planes 320x240 step 2 %s: median %.3f ms, best %.3f ms (%d planes)" % (name, ts[25] * 1e3, ts[0] * 1e3, n.value))
