"""The per-context route (one host thread + context + stream per span) against the number of contexts: python tools/ctx_threads_perf.py
Runs plain and occlusion-aware sequences through RGBD360_SEQ_ROUTE=contexts with the route's cap lifted (RGBD360_CTX_ROUTE_CAP).
Both are debug knobs since round 6: build the library with `python -m rgbd360_amd.build --debug-knobs` first (csrc/knobs.h)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RGBD360_SEQ_ROUTE"] = "contexts"
os.environ.setdefault("RGBD360_CTX_ROUTE_CAP", "16")
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W, H = 1024, 512
frames = [synth.render(synth.trajectory_pose(k % 9, 7), W, H, 7) for k in range(65)]
reg = RegisterPhotoICP(); reg.setNumPyr(4)
for occ in (0, 1):
    for ni in (1, 2, 3, 4, 6, 8, 16):
        reg.alignSequence(frames[:ni + 2], method=2, occlusion=occ, n_inflight=ni)
        t0 = time.perf_counter()
        p, s, i = reg.alignSequence(frames, method=2, occlusion=occ, n_inflight=ni)
        dt = time.perf_counter() - t0
        print("occlusion %d, %d pairs %dx%d host frames, %2d contexts: %.1f ms -> %.0f alignments/s" % (occ, len(frames) - 1, W, H, ni, dt * 1e3, (len(frames) - 1) / dt), flush=True)
