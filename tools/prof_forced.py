"""Forced GN iterations only (for rocprofv3 --kernel-trace): python3 tools/prof_forced.py [method] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
method = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
reg.forced_iters(0, np.eye(4), method, 20)
out = reg.forced_iters(0, np.eye(4), method, iters)
print("us/iter", out["elapsed_ms"] * 1e3 / iters)
