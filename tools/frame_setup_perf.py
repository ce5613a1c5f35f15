"""Wall time of the frame set-up calls with device-resident inputs (rgbd360_set_target_dev / _set_source_dev / promote) and of a whole
odometry step (promote + set_source_dev + align360): python tools/frame_setup_perf.py [W]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = W // 2
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=5)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
def dev(a):
    a = np.ascontiguousarray(a); p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return p.value
pa, da, pb, db = dev(rgbA), dev(dA), dev(rgbB), dev(dB)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
def med(f, n=30):
    t = []
    for _ in range(n):
        reg.sync(); t0 = time.perf_counter(); f(); reg.sync(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3
st = lambda: reg.setTargetFrameDev(pa, W * 3, da, W * 2, 0, H, W)
ss = lambda: reg.setSourceFrameDev(pb, W * 3, db, W * 2, 0, H, W)
st(); ss()
print("%dx%d set_target_dev %.3f ms, set_source_dev %.3f ms (each with a stream synchronise behind it)" % (W, H, med(st), med(ss)))
reg.alignFrames360(np.eye(4), 2)
print("align360 PHOTO_DEPTH %.3f ms" % med(lambda: reg.alignFrames360(np.eye(4), 2)))
def step():
    reg.promoteSourceToTarget() if hasattr(reg, "promoteSourceToTarget") else st()
    ss()
    reg.alignFrames360(np.eye(4), 2)
print("odometry step (promote + set_source_dev + align360): %.3f ms" % med(step))
