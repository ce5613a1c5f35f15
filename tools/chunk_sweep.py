"""Alignment latency / sequence throughput against the speculative-enqueue chunk sizes (env knobs read at context creation)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
frames = [synth.render(synth.trajectory_pose(k, 7), 2048, 1024, 7) for k in range(9)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(*frames[0]); reg.setSourceFrame(*frames[1])
for m in (2, 0):
    reg.alignFrames360(np.eye(4), m)
    t0 = time.perf_counter()
    for _ in range(20): reg.alignFrames360(np.eye(4), m)
    print("method %%d: %%.1f us/alignment iters %%s" %% (m, (time.perf_counter() - t0) / 20 * 1e6, reg.num_iterations), end="; ")
reg.alignSequence(frames[:5], method=2, n_inflight=2)
t0 = time.perf_counter()
p, s, it = reg.alignSequence(frames, method=2, n_inflight=2)
print("sequence %%.0f alignments/s, mean iters %%s" %% (8 / (time.perf_counter() - t0), it.mean(0).round(2).tolist()))
''' % ROOT
for l0, poll, first in ((3, 3, 8), (3, 3, 4), (3, 3, 3), (4, 3, 4), (3, 2, 3), (4, 4, 4), (3, 3, 6)):
    env = dict(os.environ, RGBD360_L0_CHUNK=str(l0), RGBD360_POLL_CHUNK=str(poll), RGBD360_FIRST_CHUNK=str(first))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    print("L0 %d poll %d first %d | %s" % (l0, poll, first, r.stdout.strip() or r.stderr.strip()[-300:]))
