"""(The RGBD360_SEQ_ROUTE / RGBD360_SEQ_CHUNKS arms need a library built with `python -m rgbd360_amd.build --debug-knobs`, csrc/knobs.h.)
Sequence throughput on one GPU, frames resident in HBM: the lock-step engine (sequence_engine.h) over slot counts / engine counts /
speculation depths, next to the per-context route.  python tools/seq_perf.py [n_pairs=256] [W=2048] [quick]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
quick = len(sys.argv) > 3
H = W // 2
uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(9)]
idx, k, step = [], 0, 1
for _ in range(n + 1):
    idx.append(k)
    if k + step < 0 or k + step >= len(uniq):
        step = -step
    k += step
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def to_device(a):
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return p.value


rgb_u = [to_device(f[0]) for f in uniq]
dep_u = [to_device(f[1]) for f in uniq]
rgb_d = [rgb_u[i] for i in idx]
dep_d = [dep_u[i] for i in idx]
host = [uniq[i] for i in idx]
ref = None


def run(tag, n_inflight, env, host_frames=False):
    global ref
    for k_, v in env.items():
        os.environ[k_] = v
    reg = RegisterPhotoICP(); reg.setNumPyr(4)
    f = (lambda: reg.alignSequence(host, method=2, n_inflight=n_inflight)) if host_frames else \
        (lambda: reg.alignSequenceDev(rgb_d, dep_d, H, W, 0, method=2, n_inflight=n_inflight))
    f()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p, s, it = f(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    if ref is None:
        ref = p
    print("%-46s n_inflight %2d: %7.2f ms -> %6.0f alignments/s (%.3f ms/pair); status ok %s; identical to first config: %s; iters %s"
          % (tag, n_inflight, dt * 1e3, n / dt, dt * 1e3 / n, bool((s == 0).all()), bool(np.array_equal(p, ref)), np.round(it.mean(0), 2).tolist()), flush=True)
    reg.close()
    for k_ in env:
        del os.environ[k_]


run("per-context route (round 1)", 3, {"RGBD360_SEQ_ROUTE": "contexts"})
for ni in ((8, 16) if quick else (1, 2, 4, 8, 16, 32)):
    run("lock-step, 1 engine", ni, {"RGBD360_SEQ_ENGINES": "1"})
for ni in ((16,) if quick else (8, 16, 32, 64)):
    run("lock-step, 2 engines", ni, {"RGBD360_SEQ_ENGINES": "2"})
if not quick:
    run("lock-step, 3 engines", 24, {"RGBD360_SEQ_ENGINES": "3"})
    run("lock-step, 4 engines", 32, {"RGBD360_SEQ_ENGINES": "4"})
    for ch in ("8,3,3", "8,5,4", "8,6,5", "10,4,3"):
        run("lock-step, 2 engines, chunks " + ch, 16, {"RGBD360_SEQ_ENGINES": "2", "RGBD360_SEQ_CHUNKS": ch})
run("lock-step, 2 engines, HOST frames", 16, {"RGBD360_SEQ_ENGINES": "2"}, host_frames=True)
run("per-context route, HOST frames", 3, {"RGBD360_SEQ_ROUTE": "contexts"}, host_frames=True)
