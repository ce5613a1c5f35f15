"""frame_planes with the refinement of segmentAndRefine switched on (what the C++ adapter does by default), for rocprofv3:
   python tools/prof_refine.py [W] [noise_sigma_m]      ANG=<angular threshold> (default 0.015: the walls stay separate planes)
   tools/prof_refine.sh runs the refinement tests, the kernel trace of a clean and a noisy frame and tools/refine_perf.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NOISE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dA = synth.make_pair(W, W // 2, seed=5)[0][1]
if NOISE > 0:
    rng = np.random.default_rng(11)
    d = dA.astype(np.float32) * np.float32(0.001) if dA.dtype == np.uint16 else dA.copy()
    mask = rng.random(d.shape) < 0.25
    d[mask] += rng.normal(0, NOISE, d.shape).astype(np.float32)[mask]
    dA = d
st = Frame360Stages(RegisterPhotoICP())
st.set_refinement(True, 0.02)
for _ in range(3): o = st.frame_planes(dA, convention=2, angular_threshold=float(os.environ.get("ANG", "0.015")), min_inliers=40, max_curvature=0.0013, max_planes=4096)
print("planes", len(o["planes"]))
