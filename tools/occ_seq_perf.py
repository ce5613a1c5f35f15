import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W, H = 1024, 512
frames = [synth.render(synth.trajectory_pose(k % 9, 7), W, H, 7) for k in range(65)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
for ni in (1, 2, 3, 4, 6, 8, 16):
    reg.alignSequence(frames[:ni + 2], method=2, occlusion=1, n_inflight=ni)
    t0 = time.perf_counter()
    p, s, i = reg.alignSequence(frames, method=2, occlusion=1, n_inflight=ni)
    dt = time.perf_counter() - t0
    print("occlusion 1 sequence, %d pairs %dx%d, n_inflight %2d: %.1f ms -> %.0f alignments/s, status ok %s" % (len(frames) - 1, W, H, ni, dt * 1e3, (len(frames) - 1) / dt, bool((s == 0).all())), flush=True)
