"""world_size-2 gloo test of the N > 1 path: sharding of the pair sequence and the all-gather of solved poses."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_pose(i):
    from rgbd360_amd import synth
    return synth.make_pose(synth.rodrigues([1, 0.2 * i, 0.1], 0.01 * (i + 1)), np.array([0.01 * i, 0.02, -0.03])).astype(np.float32)


class FakeReg:
    """Stands in for RegisterPhotoICP on machines without a GPU: records the call protocol of align_sequence."""
    nPyrLevels = 3

    def __init__(self):
        self.calls = []
        self.cur_src = None
        self.cur_trg = None

    def setTargetFrame(self, rgb, d):
        self.cur_trg = int(d[0, 0]); self.calls.append(("T", self.cur_trg))

    def setSourceFrame(self, rgb, d):
        self.cur_src = int(d[0, 0]); self.calls.append(("S", self.cur_src))

    def promoteSourceToTarget(self):
        self.cur_trg = self.cur_src; self.cur_src = None; self.calls.append(("P", self.cur_trg))

    def alignFrames360(self, guess, method):
        assert self.cur_src == self.cur_trg + 1
        self.num_iterations = [1, 2, 3 + self.cur_trg]
        self._pose = fake_pose(self.cur_trg)
        return 1 if self.cur_trg == 5 else 0          # pair 5 reports ILL-POSED: the status must survive the exchange

    def getOptimalPose(self):
        return self._pose


def get_frame(k):
    return np.zeros((2, 2, 3), np.uint8), np.full((2, 2), k, np.uint16)


def test_shard_range_is_a_contiguous_partition():
    from rgbd360_amd.batch import shard_range
    for n in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_native_gather_layout_for_uneven_shards():
    """The index arithmetic of the native multi-GPU entry's one exchange step (csrc/multi_gpu.h: every device contributes
    ceil(n_pairs / G) rows to ncclAllGather, the host scatters the gathered table back into sequence order) for 2, 3 and 8 devices,
    even and uneven shards, fewer pairs than devices: every pair has exactly one row, inside its owner's block, the owner is the rank
    whose rgbd360_shard_range holds the pair, rows keep the pair order inside a block, and the library's partition equals the
    process-per-GPU path's (rgbd360_amd.batch.shard_range).  Pure host code behind the C ABI: no device involved."""
    import ctypes as C
    from rgbd360_amd import _lib, batch, multi
    L = _lib.load()
    for world in (1, 2, 3, 8):
        for n in (1, 2, 3, 5, 7, 8, 9, 31, 255, 256, 257):
            rank, row, chunk = C.c_int(), C.c_int(), C.c_int()
            spans = [multi.shard_range(n, r, world) for r in range(world)]
            assert spans == [batch.shard_range(n, r, world) for r in range(world)]
            seen = set()
            for j in range(n):
                L.rgbd360_gather_slot(n, world, j, C.byref(rank), C.byref(row), C.byref(chunk))
                assert chunk.value == -(-n // world)
                lo, hi = spans[rank.value]
                assert lo <= j < hi, (world, n, j, rank.value, spans)
                assert row.value == rank.value * chunk.value + (j - lo)
                assert 0 <= row.value < world * chunk.value and row.value not in seen
                seen.add(row.value)
            assert len(seen) == n
            # the rows of a rank's block behind its last pair are padding: never addressed
            for r, (lo, hi) in enumerate(spans):
                assert all(r * chunk.value + k not in seen for k in range(hi - lo, chunk.value))
    # out-of-range queries answer -1 instead of indexing
    rank, row, chunk = C.c_int(), C.c_int(), C.c_int()
    L.rgbd360_gather_slot(7, 3, 7, C.byref(rank), C.byref(row), C.byref(chunk))
    assert (rank.value, row.value) == (-1, -1)


def test_align_sequence_reuses_frames_inside_a_chunk():
    from rgbd360_amd.batch import align_sequence
    reg = FakeReg()
    poses, status, iters = align_sequence(reg, get_frame, 3, 7, 2)
    assert poses.shape == (4, 4, 4) and list(status) == [0, 0, 1, 0] and iters.shape == (4, 3)
    assert [c for c in reg.calls if c[0] == "T"] == [("T", 3)]          # one target upload per chunk
    assert [c[1] for c in reg.calls if c[0] == "S"] == [4, 5, 6, 7]
    assert [c[1] for c in reg.calls if c[0] == "P"] == [4, 5, 6]
    for j in range(4):
        assert np.array_equal(poses[j], fake_pose(3 + j))


def _worker(rank, world, port, n_pairs, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from rgbd360_amd.batch import align_sequence, gather_poses, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_pairs, rank, world)
    poses, status, iters = align_sequence(FakeReg(), get_frame, lo, hi, 2)
    full, full_st, full_it = gather_poses(poses, n_pairs, dist, status=status, iters=iters)
    assert np.array_equal(gather_poses(poses, n_pairs, dist), full)           # poses-only form: same collective, same result
    np.save(os.path.join(out_dir, "full_%d.npy" % rank), full)
    np.save(os.path.join(out_dir, "status_%d.npy" % rank), full_st)
    np.save(os.path.join(out_dir, "iters_%d.npy" % rank), full_it)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [7, 8])
def test_two_rank_gloo_gather_equals_single_process(tmp_path, n_pairs):
    import torch.multiprocessing as mp
    from rgbd360_amd.batch import compose_trajectory
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_pairs, str(tmp_path)), nprocs=2, join=True)
    ref = np.stack([fake_pose(i) for i in range(n_pairs)])
    for r in range(2):
        full = np.load(tmp_path / ("full_%d.npy" % r))
        assert np.array_equal(full, ref)                               # every rank holds every pose, in pair order
        st = np.load(tmp_path / ("status_%d.npy" % r))
        assert list(st) == [1 if i == 5 else 0 for i in range(n_pairs)]
        it = np.load(tmp_path / ("iters_%d.npy" % r))
        assert np.array_equal(it, np.array([[1, 2, 3 + i] for i in range(n_pairs)]))
    traj = compose_trajectory(ref)
    assert traj.shape == (n_pairs + 1, 4, 4)
    assert np.allclose(traj[3], ref[0].astype(np.float64) @ ref[1] @ ref[2])
