"""Timing of the pinhole single-sensor alignment (320x240 and 640x480): python tools/pinhole_perf.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
from oracle import oracle as O
for (W, H) in ((320, 240), (640, 480)):
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=77)
    reg = RegisterPhotoICP(); reg.setNumPyr(3); reg.setMaskSeams(False); reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    reg.alignFrames(np.eye(4), 2)
    t0 = time.perf_counter()
    for _ in range(20): rc = reg.alignFrames(np.eye(4), 2)
    dt = (time.perf_counter() - t0) / 20
    ora = O.Oracle(n_pyr=3, math_mode=0, reduce_mode=0, mask_seams=0); ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    t0 = time.perf_counter(); st, P = ora.align_pinhole(np.eye(4), 2); t_cpu = time.perf_counter() - t0
    print("%dx%d PHOTO_DEPTH: GPU %.3f ms/alignment (iters %s, rc %d), CPU oracle %.1f ms (%d threads); pose diff %s; vs gt %s" % (
        W, H, dt * 1e3, reg.num_iterations, rc, t_cpu * 1e3, O.num_threads(), synth.pose_error(reg.getOptimalPose(), P), synth.pose_error(reg.getOptimalPose(), T)))
