"""BASELINE.json configs[3] (a sequence of pairs sharded over the GPUs of a node, poses all-gathered over RCCL) on what a one-GPU box
can show: the native single-process entry (rgbd360_multi_*, csrc/multi_gpu.h) with one device -- with and without the RCCL
exchange -- against the pair-by-pair schedule, and the process-per-GPU path with two ranks running REAL alignments on a shared
device (gloo for the exchange: RCCL refuses two ranks on one GPU).  The 8-GPU run itself is the driver's.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from rgbd360_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(n, seed=7, W=256, H=128):
    return [synth.render(synth.trajectory_pose(k, seed), W, H, seed) for k in range(n)]


def _pairwise(frames, n_pyr=3, method=2):
    from rgbd360_amd.batch import align_sequence
    from rgbd360_amd.register import RegisterPhotoICP
    reg = RegisterPhotoICP()
    reg.setNumPyr(n_pyr)
    return align_sequence(reg, lambda k: frames[k], 0, len(frames) - 1, method)


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True])
def test_native_multi_entry_one_gpu_equals_pairwise(hip_lib, force_rccl, monkeypatch):
    """rgbd360_multi_create(n_gpus = 1) + _align_sequence / _load_sequence + _align_resident == pair-by-pair alignment, bit for
    bit; with RGBD360_FORCE_RCCL=1 the rows really travel through ncclAllGather (one rank) and come back identical."""
    from rgbd360_amd.multi import MultiGpuSequence
    frames = _frames(7)
    poses, status, iters = _pairwise(frames)
    if force_rccl:
        monkeypatch.setenv("RGBD360_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("RGBD360_FORCE_RCCL", raising=False)
    m = MultiGpuSequence(n_gpus=1, n_pyr=3)
    assert m.uses_rccl == force_rccl
    for k in (1, 3):
        p, s, i = m.align_sequence(frames, method=2, n_inflight=k)
        assert np.array_equal(p, poses) and np.array_equal(s, status) and np.array_equal(i, iters), k
    m.load_sequence(frames)
    p, s, i = m.align_resident(method=2, n_inflight=2)
    assert np.array_equal(p, poses) and np.array_equal(s, status) and np.array_equal(i, iters)
    m.close()


@pytest.mark.gpu
def test_one_shot_batch_multi_and_argument_errors(hip_lib):
    import ctypes as C
    from rgbd360_amd import _lib
    from rgbd360_amd.multi import MultiGpuSequence, shard_range
    from rgbd360_amd.register import Rgbd360Error
    frames = _frames(4, seed=11)
    poses, status, iters = _pairwise(frames)
    L = _lib.load()
    p = _lib.Params()
    L.rgbd360_default_params(C.byref(p))
    p.n_pyr = 3
    n = len(frames) - 1
    rp = (C.c_void_p * len(frames))(*[f[0].ctypes.data for f in frames])
    dp = (C.c_void_p * len(frames))(*[f[1].ctypes.data for f in frames])
    out = np.zeros(n * 16, np.float32)
    res = (_lib.Result * n)()
    rc = L.rgbd360_align360_batch_multi(C.byref(p), len(frames), rp, 256 * 3, dp, 256 * 2, 0, 128, 256, None, 2, 0, 2, 1, None,
                                        out.ctypes.data_as(C.c_void_p), res)
    assert rc == 0
    for j in range(n):
        assert np.array_equal(out[16 * j:16 * j + 16].reshape(4, 4).T, poses[j]) and res[j].status == status[j]
    # more devices than the box has, a repeated device, a bad count: refused at creation, nothing half-built
    ndev = L.rgbd360_device_count()
    with pytest.raises(Rgbd360Error):
        MultiGpuSequence(n_gpus=ndev + 1)
    with pytest.raises(Rgbd360Error):
        MultiGpuSequence(n_gpus=2, device_ids=[0, 0])
    with pytest.raises(Rgbd360Error):
        MultiGpuSequence(n_gpus=0)
    m = MultiGpuSequence(n_gpus=1, n_pyr=3)
    with pytest.raises(Rgbd360Error):
        m.align_resident()
    assert m.align_sequence(frames[:1])[0].shape == (0, 4, 4)
    # the library's sharding is the Python one
    from rgbd360_amd.batch import shard_range as py_shard
    for n_items in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            assert [shard_range(n_items, r, world) for r in range(world)] == [py_shard(n_items, r, world) for r in range(world)]


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from rgbd360_amd import synth
from rgbd360_amd.batch import align_sequence_native, gather_poses, shard_range
from rgbd360_amd.register import RegisterPhotoICP
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_pairs = %(n_pairs)d
lo, hi = shard_range(n_pairs, rank, world)
frames = {k: synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(lo, hi + 1)}
if rank == 1:                                   # one pair of rank 1's shard has a blank source: its status must survive the exchange
    frames[hi] = (frames[hi][0], np.zeros_like(frames[hi][1]))
reg = RegisterPhotoICP(device=0)
reg.setNumPyr(3)
poses, status, iters = align_sequence_native(reg, lambda k: frames[k], lo, hi, 2, n_inflight=2)
full, st, it = gather_poses(poses, n_pairs, dist, status=status, iters=iters)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), poses=full, status=st, iters=it)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_run_real_alignments_and_gather(hip_lib, tmp_path):
    """Process-per-GPU path with world_size 2, both ranks on this box's one device (the exchange goes through gloo): every rank
    aligns its contiguous shard with the product library and ends up holding the whole trajectory -- poses, status and iteration
    counts equal to the single-process pair-by-pair result."""
    n_pairs = 7
    frames = _frames(n_pairs + 1)
    frames[n_pairs] = (frames[n_pairs][0], np.zeros_like(frames[n_pairs][1]))       # as rank 1 does
    poses, status, iters = _pairwise(frames)
    assert status[n_pairs - 1] != 0 and (status[:-1] == 0).all()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % dict(root=ROOT, n_pairs=n_pairs, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), outs
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(z["poses"], poses) and np.array_equal(z["status"], status) and np.array_equal(z["iters"], iters), r


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_with_two_ranks(hip_lib):
    """The driver's N > 1 command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with N = 2 on this box's one device (BENCH_SHARE_DEVICE=1: both
    ranks on device 0, gloo for the exchange because RCCL refuses two ranks on one GPU), started as a child process (the launcher
    runs before anything in it touches the GPU).  One JSON line from rank 0, n_gpus 2, the gathered pose tensor has one row per
    rank, and the sharded configs[3] sequence (resident and host frames) converged on both ranks with its size-independent checks."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "5",
           "--no-cpu-baseline", "--no-4k", "--no-rotating", "--seq-pairs", "24"]
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["metric"].startswith("Gauss-Newton iters/sec") and r["value"] > 0 and r["rccl_ranks_seen"] == 2
    assert abs(r["value"] - 2 * 5 / (r["ms_per_step"] * 5e-3)) <= 1e-6 * r["value"]          # whole-job aggregate: N * K / t
    assert "roofline" in r and 0 < r["roofline"]["frac"] < 1
    seq = r["sequence"]
    assert seq["pairs_total"] == 24 and seq["pairs_per_rank"] == 12 and seq["exchange"].startswith("gloo")
    for variant in ("resident", "host_frames"):
        v = seq[variant]
        assert v["all_status_ok"] and v["repeated_pairs_bit_identical"] and v["alignments_per_s"] > 0, (variant, v)
        assert v["max_forward_backward_residual"] < 1e-3
    pcie = seq["host_frames"]["pcie"]          # the host-frame leg against the box's measured host-to-device copy rate
    assert pcie["bound"] == "pcie" and pcie["bytes_per_alignment"] == 2048 * 1024 * 5 and 0 < pcie["frac"] < 1.5 and pcie["peak"] > 1.0, pcie


_F360_BLOCK_SCRIPT = r"""
import importlib.util, json, os, sys
sys.path.insert(0, %(root)r)
import torch                                   # first: bench.py's order (torch brings its own HIP runtime; the library then binds to it)
torch.cuda.init()
from rgbd360_amd import synth
from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(%(root)r, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
(_, dA), _, _ = synth.make_pair(1024, 512, seed=5)
print(json.dumps(bench.frame360_roofline(torch, Frame360Stages, RegisterPhotoICP, 0, dA, reps=5)))
"""


@pytest.mark.gpu
def test_bench_frame360_roofline_block(hip_lib):
    """bench.py's `roofline_frame360` block (rows a13-a15 from HIP events at the stage boundaries of rgbd360_frame_planes_dev) on a small
    frame, in a child process that imports torch first like bench.py does: three stages with SURVEY 8d's bytes, positive times that add up
    to the chain's, fractions consistent with them."""
    import json
    out = subprocess.run([sys.executable, "-c", _F360_BLOCK_SCRIPT % dict(root=ROOT)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    blk = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert blk["width"] == 1024 and blk["height"] == 512 and blk["planes"] >= 3
    st = blk["stages"]
    assert [st[k]["bytes_per_pixel"] for k in ("a13_sphere_cloud", "a14_normal_map", "a15_plane_stage")] == [14, 24, 16]
    assert all(v["us"] > 1.0 and 0 < v["frac"] < 1 for v in st.values())
    assert abs(sum(v["us"] for v in st.values()) - blk["chain"]["us"]) < 1e-3 * blk["chain"]["us"]
    assert abs(blk["chain"]["frac"] - 54 * 1024 * 512 / (blk["chain"]["us"] * 1e-6) / 8e12) < 1e-9


@pytest.mark.gpu
def test_config3_at_its_stated_size_256_pairs_resident(hip_lib):
    """BASELINE.json configs[3] at the size it names -- 256 consecutive 2048 x 1024 pairs -- through the entry bench.py's `native_multi`
    block times: rgbd360_multi_load_sequence (257 frames, each its own 10.5 MB copy in HBM, like a recorded sequence) +
    rgbd360_multi_align_resident (the lock-step engines, 32 pairs in flight, one device here).  The walk goes back and forth over 9
    rendered frames, so the 256 pairs are 16 distinct (target, source) pairs: EVERY pose, status and iteration count of the run is
    held against the pair-by-pair rgbd360_align360 of its pair, bit for bit; the walk composes to where it started."""
    from rgbd360_amd.batch import compose_trajectory
    from rgbd360_amd.multi import MultiGpuSequence
    from rgbd360_amd.register import RegisterPhotoICP
    from tests.test_gpu_parity import _pingpong
    W, H, n_unique, n_pairs = 2048, 1024, 9, 256
    uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(n_unique)]
    order = _pingpong(n_pairs, n_unique)
    assert len(order) == 257 and len(set(zip(order[:-1], order[1:]))) == 16
    one = RegisterPhotoICP()
    one.setNumPyr(4)
    pairwise = {}
    for a, b in sorted(set(zip(order[:-1], order[1:]))):
        one.setTargetFrame(*uniq[a])
        one.setSourceFrame(*uniq[b])
        rc = one.alignFrames360(np.eye(4), 2)
        assert rc == 0
        pairwise[(a, b)] = (one.getOptimalPose().copy(), list(one.num_iterations))
    m = MultiGpuSequence(n_gpus=1, n_pyr=4)
    m.load_sequence([uniq[k] for k in order])
    poses, status, iters = m.align_resident(method=2, n_inflight=32)
    m.close()
    assert poses.shape == (256, 4, 4) and not status.any()
    for j in range(n_pairs):
        p, it = pairwise[(order[j], order[j + 1])]
        assert list(iters[j])[:4] == it and np.array_equal(poses[j], p), (j, order[j], order[j + 1])
    # size-independent property: 256 = 16 x 16 steps of the back-and-forth walk end on frame 0 again
    assert order[-1] == 0
    traj = compose_trajectory(poses)
    rot, trans = synth.pose_error(traj[-1], np.eye(4))
    print(f"256-pair walk, drift at the end: {rot:.2e} rad {trans:.2e} m")
    assert rot < 2e-2 and trans < 0.1, (rot, trans)          # (128 there-and-back pairs of <= 2e-4 rad / 1e-3 m each, were they all biased one way)
