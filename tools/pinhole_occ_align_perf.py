"""Timing of the pinhole alignment with and without the occlusion-aware passes: python tools/pinhole_occ_align_perf.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
for (W, H) in ((320, 240), (640, 480)):
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=77)
    reg = RegisterPhotoICP(); reg.setNumPyr(3); reg.setMaskSeams(False); reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    for occ in (0, 1, 2):
        reg.alignFrames(np.eye(4), 2, occ)
        t0 = time.perf_counter()
        for _ in range(20): rc = reg.alignFrames(np.eye(4), 2, occ)
        dt = (time.perf_counter() - t0) / 20
        print("%dx%d PHOTO_DEPTH occlusion %d: %.3f ms/alignment (iters %s, rc %d) vs gt %s" % (W, H, occ, dt * 1e3, reg.num_iterations, rc, synth.pose_error(reg.getOptimalPose(), T)))
