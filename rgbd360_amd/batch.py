"""Batches of independent frame-pair alignments sharded over the GPUs of one node.

Unit of work = one frame pair (each alignment is a closed problem).  An odometry sequence of n_pairs pairs
(pair i = frames i, i+1; OdometryRGBD360.cpp:141-297) is cut into contiguous chunks, one per rank, so that inside
a chunk frame i+1's pyramids are reused as the next pair's target (`promoteSourceToTarget`); one frame is rendered
twice at each chunk boundary.  The data path has no collective; the solved 4x4 poses (+ status / iteration counts)
are exchanged once at the end with an all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests): 16 floats per pair,
latency-bound, not bandwidth-bound.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced partition: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def align_sequence(reg, get_frame, lo: int, hi: int, method: int, guess=None):
    """Aligns pairs lo..hi-1 of a sequence on one context.  get_frame(k) -> (rgb uint8 HxWx3, depth).
    Returns (poses [n,4,4] float32, status [n] int32, iters [n, n_pyr] int32)."""
    n = hi - lo
    poses = np.zeros((n, 4, 4), np.float32)
    status = np.zeros(n, np.int32)
    iters = np.zeros((n, reg.nPyrLevels), np.int32)
    if n == 0:
        return poses, status, iters
    rgb, d = get_frame(lo)
    reg.setTargetFrame(rgb, d)
    for j in range(n):
        rgb, d = get_frame(lo + j + 1)
        reg.setSourceFrame(rgb, d)
        status[j] = reg.alignFrames360(np.eye(4) if guess is None else guess, method)
        poses[j] = reg.getOptimalPose()
        iters[j] = reg.num_iterations
        if j + 1 < n:
            reg.promoteSourceToTarget()       # frame lo+j+1 becomes the next pair's target without re-upload
    return poses, status, iters


def align_sequence_concurrent(regs, get_frame, lo: int, hi: int, method: int, guess=None):
    """Same result as align_sequence, with len(regs) contexts (each on its own HIP stream) working on interleaved
    sub-chunks at once: in every step each context's alignment is enqueued (`alignFrames360_begin`) before any is waited
    for, so the coarse-level launches of different pairs overlap on the GPU."""
    n = hi - lo
    k = max(1, min(len(regs), n))
    n_pyr = regs[0].nPyrLevels
    poses = np.zeros((n, 4, 4), np.float32)
    status = np.zeros(n, np.int32)
    iters = np.zeros((n, n_pyr), np.int32)
    spans = [shard_range(n, c, k) for c in range(k)]          # context c owns pairs lo+a .. lo+b-1 (contiguous: frame reuse)
    for c, (a, b) in enumerate(spans):
        if b > a:
            regs[c].setTargetFrame(*get_frame(lo + a))
    steps = max((b - a) for a, b in spans) if n else 0
    g = np.eye(4) if guess is None else guess
    for s in range(steps):
        live = [c for c, (a, b) in enumerate(spans) if a + s < b]
        for c in live:
            regs[c].setSourceFrame(*get_frame(lo + spans[c][0] + s + 1))
            regs[c].alignFrames360_begin(g, method)
        for c in live:
            j = spans[c][0] + s
            status[j] = regs[c].alignFrames360_finish()
            poses[j] = regs[c].getOptimalPose()
            iters[j] = regs[c].num_iterations
            if j + 1 < spans[c][1]:
                regs[c].promoteSourceToTarget()
    return poses, status, iters


def align_sequence_native(reg, get_frame, lo: int, hi: int, method: int, guess=None, n_inflight: int = 32, occlusion: int = 0):
    """Same result as align_sequence through ONE call of the C ABI (rgbd360_align360_batch): the frames lo..hi of the chunk
    are handed over together, the library walks them with n_inflight pairs in flight (lock-step slots) and no Python between the pairs."""
    n = hi - lo
    if n <= 0:
        return np.zeros((0, 4, 4), np.float32), np.zeros(0, np.int32), np.zeros((0, reg.nPyrLevels), np.int32)
    frames = [get_frame(k) for k in range(lo, hi + 1)]
    return reg.alignSequence(frames, method=method, occlusion=occlusion, pose_guess=guess, n_inflight=n_inflight)


def gather_poses(local_poses: np.ndarray, n_total: int, dist=None, device=None, status=None, iters=None):
    """All-gather of the per-rank result blocks into full arrays on every rank: poses [n_total,4,4]; with `status` ([n] int) and
    `iters` ([n, n_pyr] int) given, the tuple (poses, status, iters) -- a pair's non-zero status must survive the exchange.
    One collective: every pair travels as a row of 16 + 1 + 8 float32 (status and iteration counts are small integers)."""
    want_meta = status is not None or iters is not None
    n_loc = local_poses.shape[0]
    st = np.zeros(n_loc, np.int32) if status is None else np.asarray(status, np.int32).reshape(n_loc)
    it = np.zeros((n_loc, 0), np.int32) if iters is None else np.asarray(iters, np.int32).reshape(n_loc, -1)
    assert it.shape[1] <= 8
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert n_loc == n_total
        return (local_poses.copy(), st.copy(), it.copy()) if want_meta else local_poses.copy()
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    max_chunk = (n_total + world - 1) // world
    ROW = 25
    lo, hi = shard_range(n_total, rank, world)
    assert n_loc == hi - lo
    rows = np.zeros((max_chunk, ROW), np.float32)
    rows[:n_loc, :16] = np.ascontiguousarray(local_poses, np.float32).reshape(n_loc, 16)
    rows[:n_loc, 16] = st
    rows[:n_loc, 17:17 + it.shape[1]] = it
    buf = torch.from_numpy(rows.reshape(-1)).to(device) if device is not None else torch.from_numpy(rows.reshape(-1))
    out = torch.empty(world * max_chunk * ROW, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, max_chunk, ROW)
    full = np.zeros((n_total, 4, 4), np.float32)
    full_st = np.zeros(n_total, np.int32)
    full_it = np.zeros((n_total, it.shape[1]), np.int32)
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        full[a:b] = out[r, : b - a, :16].reshape(b - a, 4, 4)
        full_st[a:b] = np.rint(out[r, : b - a, 16]).astype(np.int32)
        full_it[a:b] = np.rint(out[r, : b - a, 17:17 + it.shape[1]]).astype(np.int32)
    return (full, full_st, full_it) if want_meta else full


def compose_trajectory(rel_poses: np.ndarray, first=None) -> np.ndarray:
    """currentPose *= rel (OdometryRGBD360.cpp:257): serial prefix product on the host, float64."""
    T = np.eye(4) if first is None else np.asarray(first, np.float64)
    out = [T.copy()]
    for P in rel_poses:
        T = T @ np.asarray(P, np.float64)
        out.append(T.copy())
    return np.stack(out)
