import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
frames = [synth.render(synth.trajectory_pose(k, 7), 2048, 1024, 7) for k in range(3)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(*frames[0]); reg.setSourceFrame(*frames[1])
out = []
for m in (2, 0):
    reg.alignFrames360(np.eye(4), m)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): reg.alignFrames360(np.eye(4), m)
        best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    out.append("method %%d: %%.1f us/alignment iters %%s" %% (m, best, reg.num_iterations))
print("; ".join(out))
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib)], capture_output=True, text=True)
        print(os.path.basename(lib), "|", r.stdout.strip() or r.stderr.strip()[-300:])
