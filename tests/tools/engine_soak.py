"""Soak check of the lock-step sequence engine against the per-context route: random trajectories, sizes, methods, depth types and
slot counts; poses, status and iteration counts must agree BIT FOR BIT.  python tests/tools/engine_soak.py [n_trials [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)      # [seed]: another draw of cases
bad = 0
for t in range(n_trials):
    W = int(rng.choice([256, 320, 512, 640, 1024]))
    H = W // 2
    n_pyr = int(rng.integers(2, 5))
    if (W >> (n_pyr - 1)) < 32:
        n_pyr = 2
    nf = int(rng.integers(3, 12))
    seed = int(rng.integers(0, 1000))
    frames = [synth.render(synth.trajectory_pose(k, seed), W, H, seed) for k in range(nf)]
    if rng.random() < 0.3:
        frames = [(f[0], f[1].astype(np.float32) * np.float32(0.001)) for f in frames]
    method = int(rng.integers(0, 3))
    slots = int(rng.choice([1, 2, 3, 5, 8, 16, 32]))
    reg = RegisterPhotoICP()
    reg.setNumPyr(n_pyr)
    reg.debug_set_sequence_route(True)
    p0, s0, i0 = reg.alignSequence(frames, method=method, n_inflight=3)
    reg.debug_set_sequence_route(False)
    p1, s1, i1 = reg.alignSequence(frames, method=method, n_inflight=slots)
    same = np.array_equal(p0, p1) and np.array_equal(s0, s1) and np.array_equal(i0, i1)
    bad += 0 if same else 1
    print("trial %2d: %4dx%-4d n_pyr %d frames %2d method %d depth %s slots %2d -> %s (iters of pair 0: %s)" % (
        t, W, H, n_pyr, nf, method, frames[0][1].dtype, slots, "identical" if same else "DIFFERENT", list(i1[0])), flush=True)
    reg.close()
print("engine soak: %d / %d trials identical" % (n_trials - bad, n_trials))
sys.exit(1 if bad else 0)
