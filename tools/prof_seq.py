"""One resident 2048x1024 sequence through the lock-step engine, for rocprofv3 --kernel-trace --stats: python tools/prof_seq.py [n_pairs] [n_inflight]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 2048, 1024
uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(5)]
idx, k, step = [], 0, 1
for _ in range(n + 1):
    idx.append(k)
    if k + step < 0 or k + step >= len(uniq):
        step = -step
    k += step
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
def to_device(a):
    a = np.ascontiguousarray(a); p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return p.value
ru = [to_device(f[0]) for f in uniq]; du = [to_device(f[1]) for f in uniq]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
for _ in range(2):
    p, s, it = reg.alignSequenceDev([ru[i] for i in idx], [du[i] for i in idx], H, W, 0, method=2, n_inflight=ni)
print("ok", bool((s == 0).all()), it.mean(0))
