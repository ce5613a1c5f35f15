"""RegisterDensePhotoICP (csrc/rig_dense.h) at the rig's real sensor size: python tools/rig_dense_perf.py [width height]
Call times of frame set-up and of the 4-level Levenberg-Marquardt alignment of the 8 sensors, with the pose error vs the known motion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.rig import RegisterDensePhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 320
H = int(sys.argv[2]) if len(sys.argv) > 2 else 240
f1, f2, M, Rt, K = synth.make_rig_pair(W, H, seed=3, trans=0.04, rot_deg=1.5)
reg = RegisterDensePhotoICP(Rt, K, n_pyr=4)
for method in (0, 2):
    reg.setTargetFrame(f1); reg.setSourceFrame(f2); reg.align(np.eye(4), method)
    t0 = time.perf_counter()
    for _ in range(10):
        reg.setTargetFrame(f1); reg.setSourceFrame(f2)
    t_set = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        ok = reg.align(np.eye(4), method)
    t_al = (time.perf_counter() - t0) / 10
    rot, trans = synth.pose_error(reg.getPose(), M)
    print("%dx%d x 8 sensors, method %d: set-up of both frames %.2f ms, alignment %.2f ms (ok %s), pose error %.2e rad / %.2e m"
          % (W, H, method, t_set * 1e3, t_al * 1e3, ok, rot, trans), flush=True)
