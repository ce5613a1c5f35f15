"""BASELINE.json configs[3]: a sequence of n_pairs consecutive 2048x1024 pairs sharded over the GPUs of one node, one process
per GPU, poses all-gathered once over RCCL.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        tools/batch_bench_multi.py [n_pairs=256] [W=2048]
(single process: python tools/batch_bench_multi.py 32).  Frames are rendered on the fly per rank (synthetic odometry loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from rgbd360_amd import synth
from rgbd360_amd.batch import shard_range, align_sequence_native, gather_poses, compose_trajectory

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())       # several ranks may share a GPU (dev boxes)
shared = world > max(1, torch.cuda.device_count())        # dev boxes: several ranks on one GPU -> RCCL refuses, use gloo
if world > 1:
    torch.cuda.set_device(local)
    dist.init_process_group("gloo" if shared else "nccl")
from rgbd360_amd.register import RegisterPhotoICP          # after torch has picked its device
reg = RegisterPhotoICP(); reg.setNumPyr(4); reg._p.device = local if world > 1 else 0
lo, hi = shard_range(n_pairs, rank, world)
frames = {k: synth.render(synth.trajectory_pose(k, 7), W, W // 2, 7) for k in range(lo, hi + 1)}      # not timed
align_sequence_native(reg, lambda k: frames[k], lo, min(hi, lo + 2), 2)                                  # warm-up
if world > 1:
    dist.barrier()
torch.cuda.synchronize() if world > 1 else None
t0 = time.perf_counter()
poses, status, iters = align_sequence_native(reg, lambda k: frames[k], lo, hi, 2, n_inflight=3)
full = gather_poses(poses, n_pairs, dist if world > 1 else None, device=torch.device("cuda", local) if (world > 1 and not shared) else None)
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], device=torch.device("cuda", local) if not shared else None); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
if rank == 0:
    traj = compose_trajectory(full)
    gt = np.linalg.inv(synth.trajectory_pose(0, 7)) @ synth.trajectory_pose(n_pairs, 7)
    print("%d pairs %dx%d on %d GPU(s): %.1f ms -> %.0f alignments/s (host frames, H2D included); end-pose error vs ground truth %s"
          % (n_pairs, W, W // 2, world, dt * 1e3, n_pairs / dt, synth.pose_error(traj[-1], gt)))
if world > 1:
    dist.destroy_process_group()
