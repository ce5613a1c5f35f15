"""Stress: many contexts in flight, repeated; checks identical poses every time.  python tools/stress_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
frames = [synth.render(synth.trajectory_pose(k, 7), 512, 256, 7) for k in range(33)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
ref = None
t0 = time.time()
for rep in range(6):
    for k in (1, 3, 8, 16):
        for occ in (0, 2):
            p, s, it = reg.alignSequence(frames, method=2, occlusion=occ, n_inflight=k)
            key = occ
            if ref is None: ref = {}
            if key not in ref: ref[key] = p.copy()
            assert np.array_equal(ref[key], p), (rep, k, occ)
            assert (s == 0).all()
print("stress ok: %d sequence runs of 32 pairs in %.1f s" % (6 * 4 * 2, time.time() - t0))
