"""Host-frame sequence path only (rgbd360_align360_batch), pageable and pinned frames.  python tools/batch_perf_host.py [n_pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H = 2048, 1024
base = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(9)]
frames = [base[k % 9] if (k // 9) % 2 == 0 else base[8 - k % 9] for k in range(n + 1)]      # a back-and-forth walk: cheap to render
print("frame bytes: rgb %d depth %d (%s)" % (frames[0][0].nbytes, frames[0][1].nbytes, frames[0][1].dtype))
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]


def pinned_copy(a):
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), a.nbytes, 0) == 0
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(a.nbytes,)).view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


reg = RegisterPhotoICP(); reg.setNumPyr(4)
ref = None
frames_c = [(f[0].copy(), f[1].copy()) for f in frames]          # distinct pageable buffers per frame
frames_p = [(pinned_copy(f[0]), pinned_copy(f[1])) for f in frames]
for name, fr in (("pageable", frames_c), ("pinned", frames_p)):
    for k in (1, 2, 3, 4, 6, 8):
        reg.alignSequence(fr[: 2 * k + 1], method=2, n_inflight=k)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            p, st, it = reg.alignSequence(fr, method=2, n_inflight=k)
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = p.copy()
        print("%s n_inflight=%d: %.2f ms -> %.0f alignments/s (%.1f GB/s H2D); identical: %s" % (name, k, best * 1e3, n / best,
              n / best * (frames[0][0].nbytes + frames[0][1].nbytes) / 1e9, bool(np.array_equal(p, ref))))

# resident frames (rgbd360_align360_batch_dev) for the same walk
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def to_device(a):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return p.value


rgb_d = [to_device(f[0]) for f in frames_c]
dep_d = [to_device(f[1]) for f in frames_c]
for k in (1, 2, 3, 4, 6, 8):
    reg.alignSequenceDev(rgb_d[: 2 * k + 1], dep_d[: 2 * k + 1], H, W, 0, method=2, n_inflight=k)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        p, st, it = reg.alignSequenceDev(rgb_d, dep_d, H, W, 0, method=2, n_inflight=k)
        best = min(best, time.perf_counter() - t0)
    print("resident n_inflight=%d: %.2f ms -> %.0f alignments/s; identical: %s" % (k, best * 1e3, n / best, bool(np.array_equal(p, ref))))
