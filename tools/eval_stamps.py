"""Diagnostic: phase timestamps inside k_eval (needs the stamps build of the library)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librgbd360_hip_stamps.so")
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
reg._L.rgbd360_debug_eval_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for level in (0, 3):
    us = reg.time_eval_kernel(level, T, 0, True, 20)
    out = np.zeros(12, np.float64)
    reg._L.rgbd360_debug_eval_stamps(reg._ctx(), level, out.ctypes.data_as(C.c_void_p))
    for name, o in (("block 0", out[:6]), ("last block", out[6:])):
        print("level %d (%.2f us/launch) %s: pose loaded %.2f | first warp+gather issued %.2f | loop done %.2f | wave-reduced %.2f | end %.2f us" % ((level, us, name) + tuple(o[:5] / 100.0)))
    print("   last block started %.2f us after block 0" % ((out[11] - out[5]) / 100.0))

reg._L.rgbd360_debug_eval_history.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
for level, method in ((0, 0), (3, 0)):
    reg._L.rgbd360_debug_eval_history(reg._ctx(), level, 1, None)
    reg.time_eval_kernel(level, T, method, True, 20)
    h = np.zeros(120, np.float64)
    reg._L.rgbd360_debug_eval_history(reg._ctx(), level, 0, h.ctypes.data_as(C.c_void_p))
    h = h.reshape(60, 2)[:21]
    h = h[np.argsort(h[:, 0])]
    life = (h[:, 1] - h[:, 0]) / 100.0
    gap = (h[1:, 0] - h[:-1, 1]) / 100.0
    print("level %d: block-0 lifetime %.2f us (min %.2f), dead time between block 0 of consecutive launches %.2f us (min %.2f)" % (level, life[3:].mean(), life.min(), gap[3:].mean(), gap.min()))

reg._L.rgbd360_debug_eval_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for method in (0, 2):
    reg.time_eval_kernel(0, T, method, True, 5)
    pb = np.zeros(2 * 256, np.float64)
    nb = reg._L.rgbd360_debug_eval_blocks(reg._ctx(), 0, pb.ctypes.data_as(C.c_void_p))
    pb = pb.reshape(256, 2)[:nb]
    t0 = pb[:, 0].min()
    st, en = (pb[:, 0] - t0) / 100.0, (pb[:, 1] - t0) / 100.0
    print("method %d, %d blocks: start skew max %.2f us; block end min %.2f median %.2f max %.2f us; lifetime min %.2f max %.2f" % (method, nb, st.max(), en.min(), np.median(en), en.max(), (en - st).min(), (en - st).max()))
    order = np.argsort(en)
    print("   slowest blocks:", [(int(b), round(float(en[b]), 2)) for b in order[-6:]], " fastest:", [(int(b), round(float(en[b]), 2)) for b in order[:4]])
