"""A/B timing of two builds of the library on the SAME box (box-to-box spread is +-5 %):
   python tools/ab_eval.py old.so [new.so]   -- each library is timed in its own child process, alternating, 3 rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
reg.alignFrames360(np.eye(4), 2)
pose = reg.getOptimalPose()
out = []
for method in (0, 2):
    us = min(reg.time_eval_kernel(0, pose, method, True, 100) for _ in range(5))
    reg.forced_iters(0, np.eye(4), method, 200)
    it = min(reg.forced_iters(0, np.eye(4), method, 400)["elapsed_ms"] * 1e3 / 400 for _ in range(3))
    out.append("m%%d eval %%.2f us, iter %%.2f us" %% (method, us, it))
print("; ".join(out))
''' % ROOT
libs = [os.path.abspath(p) for p in sys.argv[1:]] or [os.path.join(ROOT, "rgbd360_amd", "lib", "librgbd360_hip.so")]
for rnd in range(3):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print(os.path.basename(lib), "|", r.stdout.strip() or r.stderr.strip()[-300:])
