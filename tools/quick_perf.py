"""Quick device timing of the hot kernels (development aid): python tools/quick_perf.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
rc = reg.alignFrames360(np.eye(4), 2)
pose = reg.getOptimalPose()
print("align rc", rc, "iters", reg.num_iterations, "err vs gt", synth.pose_error(pose, T))
for method, bpp in ((0, 28), (2, 40)):
    for hg in (True, False):
        us = min(reg.time_eval_kernel(0, pose, method, hg, 50) for _ in range(3))
        print("method %d hg %d: %.2f us  -> %.0f GB/s (%.1f%% of 8 TB/s)" % (method, hg, us, bpp*W*H/us/1e3, bpp*W*H/us/1e3/80))
    out = reg.forced_iters(0, np.eye(4), method, 200)
    out = reg.forced_iters(0, np.eye(4), method, 200)
    print("method %d forced: %.2f us/iter" % (method, out["elapsed_ms"]*1e3/200))
reg.forced_iters(0, np.eye(4), 2, 3)
print("solve kernel: full %.2f us, reduce-only %.2f us" % (reg.time_solve_kernel(0, 0, 50), reg.time_solve_kernel(0, 1, 50)))
for lvl in (1, 2, 3):
    us = reg.time_eval_kernel(lvl, pose, 2, True, 50)
    print("level %d eval: %.2f us" % (lvl, us))
t0 = time.perf_counter()
for _ in range(10): reg.alignFrames360(np.eye(4), 2)
print("full alignment (PHOTO_DEPTH): %.3f ms" % ((time.perf_counter()-t0)*100))
t0 = time.perf_counter()
for _ in range(5):
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
print("set_target+set_source (host images): %.3f ms" % ((time.perf_counter()-t0)*200))
