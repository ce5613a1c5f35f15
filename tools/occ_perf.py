"""Timing of the occlusion-aware alignments at full size: python tools/occ_perf.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pair = synth.add_occluder(synth.make_pair(W, W // 2, seed=5))
(rgbA, dA), (rgbB, dB), T = pair
reg = RegisterPhotoICP()
reg.setTargetFrame(rgbA, dA)
reg.setSourceFrame(rgbB, dB)
for occ, m in ((0, 2), (1, 2), (2, 0), (2, 2)):
    reg.alignFrames360(np.eye(4), m, occ)
    t0 = time.perf_counter()
    for _ in range(20): rc = reg.alignFrames360(np.eye(4), m, occ)
    dt = (time.perf_counter() - t0) / 20
    print("occlusion %d method %d: %.4f ms/alignment, iters %s, rc %d, err vs gt %s" % (occ, m, dt * 1e3, reg.num_iterations, rc, synth.pose_error(reg.getOptimalPose(), T)))
