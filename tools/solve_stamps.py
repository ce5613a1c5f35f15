"""Diagnostic: phase timestamps inside k_solve (needs the RGBD360_SOLVE_STAMPS build of the library)."""
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
for level in (0, 3):
    reg.forced_iters(level, np.eye(4), 0, 20)
    out = np.zeros(8, np.uint64)
    reg._L.rgbd360_debug_solve_stamps.argtypes = [C.c_void_p, C.c_void_p]
    reg._L.rgbd360_debug_solve_stamps(reg._ctx(), out.ctypes.data_as(C.c_void_p))
    print("level", level, "k_solve phase ends (us from kernel start): reduce %.2f | bookkeeping %.2f | LU/QR %.2f | exp %.2f | end %.2f" % tuple(out[:5] / 100.0))
