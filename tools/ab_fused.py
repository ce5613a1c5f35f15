"""A/B on ONE box of the single-pair schedules: fused solve (k_eval_fs: the solve of a pass rides in the next pass's launch) against
{k_eval, k_solve} pairs (RGBD360_FUSED_SOLVE=0).  Child process per variant, alternating; prints kernel / iteration / alignment times
and a hash of the poses (the two schedules must agree bit for bit).
   python tools/ab_fused.py [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, hashlib, numpy as np
sys.path.insert(0, %r)
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
(rgbA, dA), (rgbB, dB), T = synth.make_pair(2048, 1024, seed=1234)
reg = RegisterPhotoICP(); reg.setNumPyr(4)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
h = hashlib.sha1()
out = []
for method in (0, 2):
    reg.alignFrames360(np.eye(4), method)
    pose = reg.getOptimalPose()
    h.update(pose.tobytes()); h.update(np.asarray(reg.num_iterations, np.int32).tobytes())
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): reg.alignFrames360(np.eye(4), method)
        best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    us = min(reg.time_eval_kernel(0, pose, method, True, 100) for _ in range(5))
    try:
        fus = min(reg.time_eval_kernel(0, pose, method, 2, 100) for _ in range(5))
    except Exception:
        fus = float("nan")
    reg.forced_iters(0, np.eye(4), method, 200)
    r = [reg.forced_iters(0, np.eye(4), method, 400) for _ in range(3)]
    it = min(x["elapsed_ms"] * 1e3 / 400 for x in r)
    h.update(r[0]["pose"].tobytes())
    out.append("m%%d k_eval %%.2f us, k_eval_fs %%.2f us, iter %%.2f us, align %%.1f us %%s" %% (method, us, fus, it, best, reg.num_iterations))
print("; ".join(out), "| poses", h.hexdigest()[:12])
''' % ROOT
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rnd in range(rounds):
    for fused in ("1", "0"):
        env = dict(os.environ, RGBD360_FUSED_SOLVE=fused)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
        print("fused=" + fused, "|", r.stdout.strip() or r.stderr.strip()[-600:], flush=True)
