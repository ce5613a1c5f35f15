"""Config-4 style throughput on one GPU: n consecutive pairs of the synthetic odometry sequence, frames uploaded from host
memory, frame reuse inside the chunk.  python tools/batch_perf.py [n_pairs] [W] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.batch import align_sequence, align_sequence_concurrent
from rgbd360_amd.register import RegisterPhotoICP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
H = int(sys.argv[3]) if len(sys.argv) > 3 else W // 2
t0 = time.time()
frames = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(n + 1)]
print("rendered %d frames in %.1f s" % (n + 1, time.time() - t0))
reg = RegisterPhotoICP(); reg.setNumPyr(4)
align_sequence(reg, lambda k: frames[k], 0, 2, 2)          # warm
for method, name in ((2, "PHOTO_DEPTH"), (0, "PHOTO_CONSISTENCY")):
    t0 = time.perf_counter()
    poses, status, iters = align_sequence(reg, lambda k: frames[k], 0, n, method)
    dt = time.perf_counter() - t0
    if method == 2:
        ref_poses = poses.copy()
    errs = [synth.pose_error(poses[j], np.linalg.inv(synth.trajectory_pose(j, 7)) @ synth.trajectory_pose(j + 1, 7)) for j in range(n)]
    print("%s: %d pairs in %.2f ms -> %.0f alignments/s (%.3f ms/pair incl. H2D upload + pyramids of one new frame); status ok %d/%d; "
          "mean iters/level %s; max pose err vs ground truth %.2e rad %.2e m" % (name, n, dt * 1e3, n / dt, dt * 1e3 / n, int((status == 0).sum()), n,
          np.round(iters.mean(0), 2).tolist(), max(e[0] for e in errs), max(e[1] for e in errs)))

for k in (2, 4, 8):
    regs = []
    for _ in range(k):
        r = RegisterPhotoICP(); r.setNumPyr(4); regs.append(r)
    align_sequence_concurrent(regs, lambda i: frames[i], 0, min(n, 2 * k), 2)      # warm
    t0 = time.perf_counter()
    p2, st2, it2 = align_sequence_concurrent(regs, lambda i: frames[i], 0, n, 2)
    dt = time.perf_counter() - t0
    print("PHOTO_DEPTH, %d contexts in flight: %d pairs in %.2f ms -> %.0f alignments/s; identical poses: %s" % (k, n, dt * 1e3, n / dt, bool(np.array_equal(p2, ref_poses))))

# the same through ONE C call (rgbd360_align360_batch): no Python between the pairs
reg2 = RegisterPhotoICP(); reg2.setNumPyr(4)
for k in (1, 2, 4, 8):
    reg2.alignSequence(frames[: 2 * k + 1], method=2, n_inflight=k)      # warm (creates the sibling contexts)
    t0 = time.perf_counter()
    p3, st3, it3 = reg2.alignSequence(frames, method=2, n_inflight=k)
    dt = time.perf_counter() - t0
    print("PHOTO_DEPTH, rgbd360_align360_batch n_inflight=%d: %d pairs in %.2f ms -> %.0f alignments/s; identical poses: %s" % (k, n, dt * 1e3, n / dt, bool(np.array_equal(p3, ref_poses))))

# frames resident in HBM (rgbd360_align360_batch_dev): the compute-side rate of the sequence path
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def to_device(a):
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return p.value


rgb_d = [to_device(f[0]) for f in frames]
dep_d = [to_device(f[1]) for f in frames]
for k in (1, 2, 4):
    reg2.alignSequenceDev(rgb_d, dep_d, H, W, 0, method=2, n_inflight=k)
    t0 = time.perf_counter()
    p4, st4, it4 = reg2.alignSequenceDev(rgb_d, dep_d, H, W, 0, method=2, n_inflight=k)
    dt = time.perf_counter() - t0
    print("PHOTO_DEPTH, rgbd360_align360_batch_dev n_inflight=%d: %d pairs in %.2f ms -> %.0f alignments/s; identical poses: %s" % (k, n, dt * 1e3, n / dt, bool(np.array_equal(p4, ref_poses))))

# host frames in PINNED memory (hipHostMalloc): the uploads become real asynchronous DMA that overlaps the other contexts' kernels
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]


def pinned_copy(a):
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), a.nbytes, 0) == 0
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(a.nbytes,)).view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


frames_p = [(pinned_copy(f[0]), pinned_copy(f[1])) for f in frames]
for k in (1, 2, 4, 8):
    reg2.alignSequence(frames_p[: 2 * k + 1], method=2, n_inflight=k)
    t0 = time.perf_counter()
    p5, st5, it5 = reg2.alignSequence(frames_p, method=2, n_inflight=k)
    dt = time.perf_counter() - t0
    print("PHOTO_DEPTH, rgbd360_align360_batch PINNED host frames n_inflight=%d: %d pairs in %.2f ms -> %.0f alignments/s; identical poses: %s" % (k, n, dt * 1e3, n / dt, bool(np.array_equal(p5, ref_poses))))
