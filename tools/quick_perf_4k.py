"""4096x2048 (BASELINE.json configs[4]) timing + parity vs the oracle: python tools/quick_perf_4k.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W, H = 4096, 2048
t0 = time.time(); (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=1234); print("render %.1fs" % (time.time() - t0))
reg = RegisterPhotoICP(); reg.setNumPyr(5)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
rc = reg.alignFrames360(np.eye(4), 2); pose = reg.getOptimalPose()
print("align rc", rc, "iters", reg.num_iterations, "err vs gt", synth.pose_error(pose, T))
for method, bpp in ((0, 28), (2, 40)):
    us = min(reg.time_eval_kernel(0, pose, method, True, 30) for _ in range(3))
    print("method %d: %.2f us -> %.0f GB/s (%.1f%% of 8 TB/s)" % (method, us, bpp * W * H / us / 1e3, bpp * W * H / us / 1e3 / 80))
    out = reg.forced_iters(0, np.eye(4), method, 100)
    print("method %d forced: %.2f us/iter" % (method, out["elapsed_ms"] * 1e3 / 100))
t0 = time.perf_counter()
for _ in range(5): reg.alignFrames360(np.eye(4), 2)
print("full alignment (PHOTO_DEPTH, 5 levels): %.3f ms" % ((time.perf_counter() - t0) * 200))
if len(sys.argv) > 1:
    from oracle import oracle as O
    ora = O.Oracle(n_pyr=5, math_mode=1, reduce_mode=1); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    t0 = time.time(); st, pose_ref = ora.align360(np.eye(4), 2); print("oracle %.2fs iters" % (time.time() - t0), list(ora.result.iters)[:5], "gpu vs oracle", synth.pose_error(pose, pose_ref))
