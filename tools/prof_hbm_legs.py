"""The HBM-fed measurements of the bench line, one regime per process, so that a `rocprofv3 --kernel-trace --stats` (or --pmc)
run of this script has ONE row per kernel and regime (tools/collect_profiles.sh).  Prints the HIP-event figure beside it.
usage: python3 tools/prof_hbm_legs.py rotating|4k|batch [reps]
  rotating  k_eval<m,true> / k_eval_fs<m>, m = 0, 2, at 2048x1024, launches rotating over 5 (photo+depth: 4) copies of the pair:
            bench.py's roofline_hbm_rotating (rgbd360_time_eval_kernel_rotating)
  4k        the same kernels at 4096x2048 on one pair: bench.py's roofline_4096x2048
  batch     k_eval_b<m,true> over 16 slots at 2048x1024: bench.py's iteration_lockstep.pass / pass_photo_depth
No alignment is run: the pose is the pair's ground truth, so the kernels named above are launched in the one regime only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP

mode = sys.argv[1] if len(sys.argv) > 1 else "rotating"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
LLC = 256 << 20


def mk(pair, n_pyr):
    (rgbA, dA), (rgbB, dB), _ = pair
    r = RegisterPhotoICP()
    r.setNumPyr(n_pyr)
    r.setTargetFrame(rgbA, dA)
    r.setSourceFrame(rgbB, dB)
    return r


if mode == "rotating":
    W, H = 2048, 1024
    pair = synth.make_pair(W, H, seed=1234)
    n_rot = max(2, int(np.ceil(1.6 * LLC / (28 * W * H))))          # bench.py's count (sized on the photo working set)
    regs = [mk(pair, 4) for _ in range(n_rot)]
    for m in (0, 2):
        for hg, name in ((True, "k_eval<%d,true>" % m), (2, "k_eval_fs<%d>" % m)):
            us = [RegisterPhotoICP.time_eval_kernel_rotating(regs, 0, pair[2], m, hg, reps * n_rot) for _ in range(3)]
            by = (28 if m == 0 else 40) * W * H
            print("rotating %-16s copies %d  HIP events avg us %s  -> %.3f of 8 TB/s" % (name, n_rot, ["%.2f" % u for u in us], by / (sorted(us)[1] * 1e-6) / 8e12))
elif mode == "4k":
    W, H = 4096, 2048
    pair = synth.make_pair(W, H, seed=1234)
    reg = mk(pair, 5)
    for m in (0, 2):
        for hg, name in ((True, "k_eval<%d,true>" % m), (2, "k_eval_fs<%d>" % m)):
            us = [reg.time_eval_kernel(0, pair[2], m, hg, 3 * reps) for _ in range(3)]
            by = (20 if m == 0 else 32) * W * H          # levels of 4 Mpx and more run the recompute form of the source stream (8 B per source pixel)
            print("4096x2048 %-16s HIP events avg us %s  -> %.3f of 8 TB/s" % (name, ["%.2f" % u for u in us], by / (sorted(us)[1] * 1e-6) / 8e12))
elif mode == "batch":
    W, H, P = 2048, 1024, 16
    pair = synth.make_pair(W, H, seed=1234)
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    for m in (0, 2):
        fb = [reg.forced_iters_batch(P, pair[0], pair[1], 0, np.eye(4), m, 4) for _ in range(3)]
        us = [f["pass_avg_us"] for f in fb]
        by = P * (20 if m == 0 else 32) * W * H          # the engine's large levels carry 8-byte {depth, I} source records
        print("batch k_eval_b<%d,true> %d slots  HIP events avg launch us %s  -> %.3f of 8 TB/s" % (m, P, ["%.1f" % u for u in us], by / (sorted(us)[1] * 1e-6) / 8e12))
else:
    raise SystemExit("mode: rotating | 4k | batch")
