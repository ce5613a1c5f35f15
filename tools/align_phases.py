"""Host-side phases of one alignment: time spent in _begin (enqueue) and in _finish (wait + top-ups), per chunk setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
frames = [synth.render(synth.trajectory_pose(k, 7), 2048, 1024, 7) for k in range(2)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(*frames[0]); reg.setSourceFrame(*frames[1])
for _ in range(3): reg.alignFrames360(np.eye(4), 2)
tb = tf = 0.0
N = 30
for _ in range(N):
    t0 = time.perf_counter(); reg.alignFrames360_begin(np.eye(4), 2); t1 = time.perf_counter(); reg.alignFrames360_finish(); t2 = time.perf_counter()
    tb += t1 - t0; tf += t2 - t1
print("begin %%.1f us, finish %%.1f us, total %%.1f us, iters %%s" %% (tb / N * 1e6, tf / N * 1e6, (tb + tf) / N * 1e6, reg.num_iterations))
''' % ROOT
for l0, poll, first in ((3, 3, 8), (6, 3, 8), (3, 3, 4), (3, 3, 6), (6, 3, 6), (7, 3, 6)):
    env = dict(os.environ, RGBD360_L0_CHUNK=str(l0), RGBD360_POLL_CHUNK=str(poll), RGBD360_FIRST_CHUNK=str(first))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    print("L0 %d poll %d first %d | %s" % (l0, poll, first, r.stdout.strip() or r.stderr.strip()[-300:]))
