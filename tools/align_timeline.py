"""Kernel timeline of full 4-level alignments (run under rocprofv3 --kernel-trace; tools/align_timeline.sh prints spans and gaps).
   python tools/align_timeline.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB); reg.sync()
import time
for _ in range(5): reg.alignFrames360(np.eye(4), 2)
t0 = time.perf_counter()
for _ in range(n): reg.alignFrames360(np.eye(4), 2)
print("alignment %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6), list(reg.num_iterations))
