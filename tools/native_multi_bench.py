"""The single-process multi-GPU sequence entry (rgbd360_multi_*, csrc/multi_gpu.h) on n_gpus devices: one host thread per device,
contiguous shards of pairs, one ncclAllGather of the solved poses.  Started by bench.py as a child process (its own HIP runtime, no
torch); prints ONE JSON line.
    python tools/native_multi_bench.py <n_gpus> <n_pairs> <W> <H> [frames.npz]
The frames (a few unique ones, walked back and forth as bench.py's sequence block does) come from the .npz bench.py wrote, or are
rendered here."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.multi import MultiGpuSequence, shard_range

n_gpus, n_pairs, W, H = (int(x) for x in sys.argv[1:5])
if len(sys.argv) > 5:
    z = np.load(sys.argv[5])
    uniq = [(z["rgb"][k], z["depth"][k]) for k in range(len(z["rgb"]))]
else:
    uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(9)]
idx, k, step = [], 0, 1
for _ in range(n_pairs + 1):
    idx.append(k)
    if k + step < 0 or k + step >= len(uniq):
        step = -step
    k += step
frames = [uniq[i] for i in idx]
t0 = time.perf_counter()
m = MultiGpuSequence(n_gpus=n_gpus, n_pyr=4)
t_create = time.perf_counter() - t0
out = {"n_gpus": n_gpus, "pairs_total": n_pairs, "uses_rccl": m.uses_rccl, "create_s": t_create,
       "entry": "rgbd360_multi_load_sequence + rgbd360_multi_align_resident / rgbd360_multi_align_sequence (one process, one host thread per device)"}
m.load_sequence(frames)
m.align_resident(method=2, n_inflight=32)                  # warm: engines, buffers, RCCL channels
for name, fn in (("resident", lambda: m.align_resident(method=2, n_inflight=32)), ("host_frames", lambda: m.align_sequence(frames, method=2, n_inflight=32))):
    if name == "host_frames":
        fn()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        poses, status, iters = fn()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    same = True
    seen = {}
    for j in range(n_pairs):        # only inside a shard: the recurrence of a pair on another device is equal too, but compare like with like
        key = (idx[j], idx[j + 1])
        if key in seen:
            same &= bool(np.array_equal(poses[seen[key]], poses[j]))
        else:
            seen[key] = j
    out[name] = {"alignments_per_s": n_pairs / med, "elapsed_ms_median": med * 1e3, "elapsed_ms_all": [t * 1e3 for t in times],
                 "all_status_ok": bool((status == 0).all()), "repeated_pairs_bit_identical": same,
                 "mean_iters_per_level": np.round(iters.mean(0), 3).tolist()}
m.close()
print(json.dumps(out), flush=True)
