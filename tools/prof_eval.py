"""Launches the fused per-pixel kernel a few times (for rocprofv3 --pmc / --kernel-trace passes).
usage: python3 tools/prof_eval.py [W] [H] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
for method in (0, 2):
    us = reg.time_eval_kernel(0, T, method, True, reps)       # k_eval: the per-pixel pass alone
    print("method", method, "k_eval avg us", us)
    try:
        us = reg.time_eval_kernel(0, T, method, 2, reps)      # k_eval_fs: solve prologue + pass (forced schedule)
        print("method", method, "k_eval_fs avg us", us)
    except Exception as e:                                    # (the fused schedule switched off: a debug build with RGBD360_FUSED_SOLVE=0)
        print("method", method, "k_eval_fs not run:", e)
