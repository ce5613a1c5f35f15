"""Soak check of the resident coarse-level launches (k_coarse_persist: a pyramid level of <= 32 blocks as ONE launch that loops {solve, pass}
itself, the default of rgbd360_align360) against the launch-per-iteration schedule (RGBD360_PERSIST_COARSE=0): random scenes, sizes, pyramid
depths, methods, start poses and residual weights; pose, status, iteration counts and the result record must agree BIT FOR BIT.
    python tests/tools/persist_soak.py [n_trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(4004)
bad = 0
for t in range(n_trials):
    W = int(rng.choice([128, 256, 320, 512, 640, 1024, 2048]))
    H = W // 2
    n_pyr = int(rng.integers(1, 7))
    while n_pyr > 1 and (W >> (n_pyr - 1)) < 32:
        n_pyr -= 1
    seed = int(rng.integers(0, 1000))
    trans, rot = float(rng.choice([0.0, 0.02, 0.06, 0.15])), float(rng.choice([0.0, 1.0, 2.0, 5.0]))
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=seed, trans=trans, rot_deg=rot)
    if rng.random() < 0.3:
        dA, dB = dA.astype(np.float32) * np.float32(0.001), dB.astype(np.float32) * np.float32(0.001)
    if rng.random() < 0.12:
        rgbB = np.zeros_like(rgbB)              # a blank source frame: no salient pixel
    method = int(rng.integers(0, 3))
    guess = np.eye(4)
    if rng.random() < 0.3:
        guess = np.linalg.inv(T) if rng.random() < 0.5 else T        # start at the answer / at twice the motion
    var = (float(rng.choice([3.0, 6.0, 12.0])) / 255.0, float(rng.choice([0.005, 0.01, 0.03]))) if rng.random() < 0.3 else None
    res, us = [], []
    for persist in ("1", "0"):
        os.environ["RGBD360_PERSIST_COARSE"] = persist
        reg = RegisterPhotoICP()
        reg.setNumPyr(n_pyr)
        if var is not None:
            reg.setGrayVariance(var[0]); reg.setDepthVariance(var[1])
        reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
        rc = reg.alignFrames360(guess, method)
        res.append((rc, reg.getOptimalPose().copy(), list(reg.num_iterations), reg.getHessian().copy(), reg.getGradient().copy()))
        for _ in range(3): reg.alignFrames360(guess, method)
        t0 = time.perf_counter()
        for _ in range(10): reg.alignFrames360(guess, method)
        us.append((time.perf_counter() - t0) / 10 * 1e6)
        same_again = reg.alignFrames360(guess, method) == rc and np.array_equal(reg.getOptimalPose(), res[-1][1])
        reg.close()
        if not same_again:
            print("   NOT REPRODUCIBLE with persist", persist); bad += 1
    os.environ.pop("RGBD360_PERSIST_COARSE")
    a, b = res
    same = a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    bad += 0 if same else 1
    print("trial %2d: %4dx%-4d n_pyr %d method %d motion %.2f m / %.0f deg -> status %d iters %s  %6.1f us resident / %6.1f us per-iteration launches  %s" % (
        t, W, H, n_pyr, method, trans, rot, a[0], a[2], us[0], us[1], "identical" if same else "DIFFERENT %s %s" % (b[0], b[2])), flush=True)
print("%d trials, %d mismatches" % (n_trials, bad))
sys.exit(1 if bad else 0)
