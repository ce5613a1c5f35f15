"""Soak check of the fused-solve schedule (one k_eval_fs launch per Gauss-Newton iteration, the library's default) against the two-launch
schedule (k_eval + k_solve, rgbd360_debug_set_schedule): random scenes, sizes, pyramid depths, methods, start poses and residual weights; pose,
status, iteration counts and the result record must agree BIT FOR BIT.  A third of the trials run an occlusion-aware mode on an occluder scene: there the fused schedule is
{k_occ_build_fs, k_eval_occ} against {k_occ_build, k_eval_occ, k_solve}.  python tests/tools/fused_soak.py [n_trials [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3003)      # [seed]: another draw of cases
bad = 0
for t in range(n_trials):
    W = int(rng.choice([128, 256, 320, 512, 640, 1024, 2048]))
    H = W // 2
    n_pyr = int(rng.integers(1, 6))
    while n_pyr > 1 and (W >> (n_pyr - 1)) < 32:
        n_pyr -= 1
    seed = int(rng.integers(0, 1000))
    trans, rot = float(rng.choice([0.0, 0.02, 0.06, 0.15])), float(rng.choice([0.0, 1.0, 2.0, 5.0]))
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=seed, trans=trans, rot_deg=rot)
    if rng.random() < 0.3:
        dA, dB = dA.astype(np.float32) * np.float32(0.001), dB.astype(np.float32) * np.float32(0.001)
    if rng.random() < 0.15:
        rgbB = np.zeros_like(rgbB)              # a blank source frame: no salient pixel
    method = int(rng.integers(0, 3))
    occlusion = int(rng.integers(1, 3)) if rng.random() < 0.34 else 0
    if occlusion:
        (rgbA, dA), (rgbB, dB), T = synth.add_occluder(((rgbA, dA), (rgbB, dB), T))
    guess = np.eye(4)
    if rng.random() < 0.3:
        guess = np.linalg.inv(T) if rng.random() < 0.5 else T        # start at the answer / at twice the motion
    res = []
    for fused in ("1", "0"):
        reg = RegisterPhotoICP()
        reg.setNumPyr(n_pyr)
        reg.debug_set_schedule(fused_solve=(fused == "1") or bool(occlusion), fused_occ=(fused == "1") or not occlusion)
        if fused == "1":
            var = (float(rng.choice([3.0, 6.0, 12.0])) / 255.0, float(rng.choice([0.005, 0.01, 0.03]))) if rng.random() < 0.3 else None
        if var is not None:
            reg.setGrayVariance(var[0]); reg.setDepthVariance(var[1])
        reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
        rc = reg.alignFrames360(guess, method, occlusion)
        res.append((rc, reg.getOptimalPose().copy(), list(reg.num_iterations), reg.getHessian().copy(), reg.getGradient().copy()))
        reg.close()
    a, b = res
    same = a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    bad += 0 if same else 1
    print("trial %2d: %4dx%-4d n_pyr %d method %d occlusion %d motion %.2f m / %.0f deg depth %s -> status %d iters %s %s" % (
        t, W, H, n_pyr, method, occlusion, trans, rot, dA.dtype, a[0], a[2], "identical" if same else "DIFFERENT"), flush=True)
print("fused-solve soak: %d / %d trials identical" % (n_trials - bad, n_trials))
sys.exit(1 if bad else 0)
