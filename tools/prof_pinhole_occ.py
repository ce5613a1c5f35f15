"""The pinhole occlusion-aware passes for rocprofv3 (kernel trace + stats) and as wall time per alignment:
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/prof_pinhole_occ.py [W H]
Runs alignFrames(occlusion 0 / 1) on a synthetic sensor pair (PHOTO_DEPTH) and prints the call times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
W = int(sys.argv[1]) if len(sys.argv) > 1 else 320
H = int(sys.argv[2]) if len(sys.argv) > 2 else 240
(rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=77)
reg = RegisterPhotoICP(); reg.setNumPyr(3); reg.setMaskSeams(False); reg.setCameraMatrix(K)
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
for occ in (0, 1):
    for _ in range(3): reg.alignFrames(np.eye(4), 2, occ)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): rc = reg.alignFrames(np.eye(4), 2, occ)
    ms = (time.perf_counter() - t0) / n * 1e3
    rot, trans = synth.pose_error(reg.getOptimalPose(), T)
    print("%dx%d occlusion %d: %.3f ms per alignFrames call, status %d, iterations %s, pose error %.2e rad %.2e m" % (W, H, occ, ms, rc, reg.num_iterations, rot, trans))
