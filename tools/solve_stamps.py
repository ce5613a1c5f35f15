"""Diagnostic: phase timestamps inside the solve (k_solve, or the prologue of the fused launch k_eval_fs).
Needs the RGBD360_SOLVE_STAMPS build of the library:  python tools/solve_stamps.py build   (cross-compiles, no GPU needed),
then on the GPU box:  python tools/solve_stamps.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbd360_amd import _lib, build as B
STAMPS_LIB = os.path.join(os.path.dirname(B.LIB), "librgbd360_hip_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call([B.hipcc()] + B.FLAGS + ["-DRGBD360_SOLVE_STAMPS", "-o", STAMPS_LIB, B.SRC] + B.LINK)
    print(STAMPS_LIB)
    sys.exit(0)
import numpy as np
_lib.LIB_PATH = STAMPS_LIB
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
names = ["staged+row sums", "totals", "bookkeeping (wave 0)", "loads issued", "end", "inverse (wave 1)", "update+exp+cand (wave 1)", "rank (wave 2)"]
for level in (0, 3):
    for method in (0, 2):
        reg.forced_iters(level, np.eye(4), method, 20)
        out = np.zeros(8, np.uint64)
        reg._L.rgbd360_debug_solve_stamps.argtypes = [C.c_void_p, C.c_void_p]
        reg._L.rgbd360_debug_solve_stamps(reg._ctx(), out.ctypes.data_as(C.c_void_p))
        print("level", level, "method", method, "| phase ends, us from kernel start:", "; ".join("%s %.2f" % (n, v / 100.0) for n, v in zip(names, out)))
