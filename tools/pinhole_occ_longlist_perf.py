"""How long one occlusion-aware pinhole evaluation takes when many source pixels pile on few target pixels (a zoom-out / a collapse: poses
a Levenberg-Marquardt trial can propose): python tools/pinhole_occ_longlist_perf.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
for (W, H) in ((320, 240), (640, 480)):
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=7, trans=0.03, rot_deg=1.0)
    reg = RegisterPhotoICP(); reg.setNumPyr(1); reg.setMaskSeams(False); reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    for name, push in (("rendered motion", 0.0), ("zoom-out 1 m", 1.0), ("zoom-out 3 m", 3.0), ("push 20 m", 20.0), ("collapse 100 m", 100.0), ("collapse 500 m", 500.0)):
        P = np.eye(4); P[2, 3] = push
        pose = P @ np.array(T, dtype=np.float64)
        reg.eval_pinhole(0, pose, 2, 1)
        t0 = time.perf_counter()
        for _ in range(5): e = reg.eval_pinhole(0, pose, 2, 1)
        dt = (time.perf_counter() - t0) / 5
        print("%dx%d %-16s: %8.3f ms per evaluation (occlusion 1, PHOTO_DEPTH), numVisible %d" % (W, H, name, dt * 1e3, e["n_rows"]))
    reg.close()
