"""How long one occlusion-aware spherical evaluation takes when a (diverged) pose piles many source pixels on few target pixels:
python tools/occ_longlist_perf.py [W]   (stops at the first evaluation slower than 50 ms)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, W // 2, seed=5)
reg = RegisterPhotoICP(); reg.setNumPyr(1)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
for push in (0.0, 1.0, 3.0, 10.0, 30.0, 100.0, 500.0):
    P = np.eye(4); P[2, 3] = push
    pose = P @ np.array(T, dtype=np.float64)
    slow = False
    for occ in (1, 2):
        t0 = time.perf_counter(); e = reg.eval(0, pose, 2, occ); first = time.perf_counter() - t0
        if first > 0.05:
            print("%dx%d push %5.1f m occlusion %d: %.1f ms (first call; stopping)" % (W, W // 2, push, occ, first * 1e3), flush=True)
            slow = True
            break
        t0 = time.perf_counter()
        for _ in range(5): e = reg.eval(0, pose, 2, occ)
        dt = (time.perf_counter() - t0) / 5
        print("%dx%d push %5.1f m occlusion %d: %8.3f ms per evaluation, numVisible %d" % (W, W // 2, push, occ, dt * 1e3, e["n_visible"]), flush=True)
    if slow: break
