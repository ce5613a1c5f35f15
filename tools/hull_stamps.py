"""Per-wave clock accounting of k_f360_hull_extremes (diagnostic build with -DRGBD360_HULL_DBG):
     python tools/hull_stamps.py build            cross-compiles rgbd360_amd/lib/librgbd360_hip_hulldbg.so
     python tools/hull_stamps.py [width [angular_threshold [min_inliers]]]      on the GPU box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "rgbd360_amd", "lib", "librgbd360_hip_hulldbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from rgbd360_amd import build as B
    B.compile_library(LIB, ["-DRGBD360_HULL_DBG"], selects_vop3=False)
    print(LIB); sys.exit(0)
os.environ["RGBD360_LIB"] = LIB
import ctypes as C
import numpy as np
from rgbd360_amd import synth, _lib
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ANG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
MIN_INLIERS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dA = synth.make_pair(W, W // 2, seed=5)[0][1]
st = Frame360Stages(RegisterPhotoICP())
for _ in range(3):
    out = st.frame_planes(dA, convention=2, angular_threshold=ANG, min_inliers=MIN_INLIERS, max_curvature=0.0013, max_planes=4096)
buf = np.zeros((4096, 8), np.uint64)
L = C.CDLL(LIB)
assert L.rgbd360_debug_hull_stats(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.astype(np.float64)
names = ["setup", "stretches", "tail", "uv", "walk", "flush", "entries", "runs"]
print("planes", len(out["planes"]), "| cycle counter ticks per wave (s_memtime; 100 ticks = 1 us at 100 MHz, ~24 ticks per us... see the ratio below)")
tot = b[:, 0] + b[:, 1] + b[:, 2]
order = np.argsort(-tot)
print("mean per wave: " + "  ".join("%s %.0f" % (n, b[:, i].mean()) for i, n in enumerate(names)))
print("slowest waves (total | " + " ".join(names) + "):")
for w in order[:12]:
    print("  wave %4d  %8.0f | " % (w, tot[w]) + " ".join("%8.0f" % x for x in b[w]))
for q in (50, 90, 99, 100):
    print("percentile %3d of wave totals: %.0f" % (q, np.percentile(tot, q)))
bw = int(order[0]) // 16
print("all waves of the slowest block (%d), by time in the stretches:" % bw)
for w in sorted(range(bw * 16, bw * 16 + 16), key=lambda w: -b[w, 1]):
    print("  wave %4d  %8.0f | " % (w, tot[w]) + " ".join("%8.0f" % x for x in b[w]))
blk = tot.reshape(256, 16).max(1)
print("block = max of its waves: mean %.0f  max %.0f" % (blk.mean(), blk.max()))
