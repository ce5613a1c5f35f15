"""segmentAndRefine's refinement at full size: sweeps, relabelled pixels and call time of rgbd360_frame_planes with / without it.
python tools/refine_perf.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = W // 2
(rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
rng = np.random.default_rng(1)
d = dA.astype(np.float32)
for _ in range(40):        # noisy patches (sensor noise inside 2 cm of the walls), holes and a box: planes have something to grow into
    r0, c0 = int(rng.integers(0, H - H // 16)), int(rng.integers(0, W - W // 16))
    hh, ww = int(rng.integers(8, H // 16)), int(rng.integers(8, W // 16))
    d[r0:r0 + hh, c0:c0 + ww] += rng.normal(0, 10.0, size=(hh, ww)).astype(np.float32)
d[H // 3:H // 3 + 20, W // 4:W // 4 + 40] = 0
d[H // 2:H // 2 + 60, W // 2:W // 2 + 120] *= 0.8
d = np.clip(d, 0, 65535).astype(np.uint16)
st = Frame360Stages(RegisterPhotoICP())
ang = 0.03 * 1024 / W
for on in (False, True, False, True):
    st.set_refinement(on, 0.02)
    st.frame_planes(d, convention=2, angular_threshold=ang, min_inliers=40 * W // 512, max_planes=1024)
    t0 = time.perf_counter()
    for _ in range(5):
        out = st.frame_planes(d, convention=2, angular_threshold=ang, min_inliers=40 * W // 512, max_planes=1024)
    dt = (time.perf_counter() - t0) / 5
    print("refine %-5s: %.2f ms per frame_planes call (host depth in, maps out), %d planes, inliers %d, stats %s"
          % (on, dt * 1e3, len(out["planes"]), sum(p["count"] for p in out["planes"]), st.refinement_stats() if on else "-"), flush=True)
