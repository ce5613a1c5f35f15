"""What the reference's warp arithmetic (rgbd360_set_index_arithmetic(ctx, 1)) costs: the per-pixel pass, the fused launch and a whole
alignment at 2048 x 1024 in both arithmetics, same context (GPU box).  python tools/libm_mode_perf.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
(rgbA, dA), (rgbB, dB), T = synth.make_pair(W, W // 2, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
for mode in (0, 1, 0, 1):
    reg.set_index_arithmetic(mode)
    out = []
    for method in (0, 2):
        p = reg.time_eval_kernel(0, T, method, True, 20)
        f = reg.time_eval_kernel(0, T, method, 2, 20)
        reg.alignFrames360(np.eye(4), method)
        t0 = time.perf_counter()
        for _ in range(20):
            reg.alignFrames360(np.eye(4), method)
        al = (time.perf_counter() - t0) / 20 * 1e6
        out.append("method %d: pass %.2f us, fused launch %.2f us, alignment %.1f us (iters %s)" % (method, p, f, al, list(reg.num_iterations)))
    print("index arithmetic %d (%s) | %s" % (mode, "reference libm" if mode else "device definition", " | ".join(out)), flush=True)
