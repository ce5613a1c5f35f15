"""Same-box A/B of the sequence engine's source-record form: alignments/s of a resident 2048x1024 sequence (distinct device copies per
frame, PHOTO_DEPTH, 32 pairs in flight) with RGBD360_SEQ_RECOMPUTE_MIN_PX at its default (compact {depth, I} records on the large
levels) and switched off, each arm in its own process, alternating; poses hashed (the two forms are bit-identical).
    python tools/seq_ab.py [n_pairs=128] [rounds=2]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, hashlib, numpy as np
sys.path.insert(0, %r)
import torch
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n = int(sys.argv[1]); W, H = 2048, 1024
uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(9)]
idx, k, step = [], 0, 1
for _ in range(n + 1):
    idx.append(k)
    if k + step < 0 or k + step >= len(uniq): step = -step
    k += step
dev = torch.device("cuda", 0)
rgb_t = [torch.from_numpy(uniq[i][0]).to(dev) for i in idx]
dep_t = [torch.from_numpy(uniq[i][1].view(np.int16)).to(dev) for i in idx]
torch.cuda.synchronize()
reg = RegisterPhotoICP(device=0); reg.setNumPyr(4)
rp, dp = [t.data_ptr() for t in rgb_t], [t.data_ptr() for t in dep_t]
reg.alignSequenceDev(rp[:65], dp[:65], H, W, 0, method=2, n_inflight=32)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); res = reg.alignSequenceDev(rp, dp, H, W, 0, method=2, n_inflight=32); best = min(best, time.perf_counter() - t0)
print("%%.0f alignments/s  (%%.3f ms per pair)  status ok %%s  poses %%s" %% (n / best, best / n * 1e3, bool((res[1] == 0).all()), hashlib.sha1(res[0].tobytes()).hexdigest()[:12]))
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "128"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for _ in range(rounds):
    for tag, val in (("compact records (default)", None), ("16-byte records", str(1 << 30))):
        env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
        if val: env["RGBD360_SEQ_RECOMPUTE_MIN_PX"] = val
        r = subprocess.run([sys.executable, "-c", CHILD, n], capture_output=True, text=True, env=env)
        print("%-28s|" % tag, r.stdout.strip() or r.stderr.strip()[-500:], flush=True)
