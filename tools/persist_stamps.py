"""Per-trip clocks of block 0 of the resident coarse-level launch (k_coarse_persist) in a build with -DRGBD360_PERSIST_STAMPS:
    python tools/ab_libs.py build pstamps=-DRGBD360_PERSIST_STAMPS        (no GPU needed)
    RGBD360_PERSIST_COARSE=1 RGBD360_LIB=rgbd360_amd/lib/librgbd360_hip_pstamps.so python tools/persist_stamps.py      (GPU box)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
for _ in range(5): reg.alignFrames360(np.eye(4), 2)
w = np.zeros(128, np.uint64)
reg._L.rgbd360_debug_persist(reg._ctx(), w.ctypes.data_as(C.c_void_p))
print("gave up:", int(w[0]), "fallen back:", int(w[1]) & 1, "resident levels mask:", int(w[1]) >> 8, "iterations", reg.num_iterations)
t = w[2:2 + 12 * 8].reshape(12, 8).astype(np.float64) / 100.0      # us
prev_end = 0.0
for k in range(12):
    if t[k, 4] == 0: break
    print("trip %2d: start %6.2f | rows polled +%5.2f summed +%5.2f | solve decided +%5.2f | pass done +%5.2f   (trip %5.2f us)" % (
        k, t[k, 0], (t[k, 1] - t[k, 0]) if k else 0.0, (t[k, 2] - t[k, 1]) if k else 0.0, t[k, 4] - t[k, 3], (t[k, 5] - t[k, 4]) if t[k, 5] else 0.0,
        (t[k + 1, 0] - t[k, 0]) if k < 11 and t[k + 1, 0] else 0.0))
