"""Soak check of the pinhole occlusion-aware passes (k_pin_occ_keys -> stable sort -> k_pin_occ_walk) against the CPU oracle
(errorPhotoICP_Occ1/2, calcHessGrad_Occ1/2 in the device-arithmetic mode): random scenes, sizes, levels, methods, occlusion modes and
poses -- the rendered motion, zoom-outs that pile many source pixels on a target pixel, half-turns that put the points BEHIND the camera
(negative inverse depths: the z-buffer's `buf > 0` / `buf == 0` tests), collapses onto a few pixels.  Every count must agree exactly
(the z-buffer's accept chain is integer work), the sums to the plain pass's tolerances.  python tests/tools/pinhole_occ_soak.py [n_trials [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
from oracle import oracle as O
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 515)      # [seed]: another draw of cases
bad = 0
for t in range(n_trials):
    W, H = [(160, 120), (320, 240), (640, 480), (200, 152)][int(rng.integers(0, 4))]
    n_pyr = int(rng.integers(1, 4))
    seed = int(rng.integers(0, 1000))
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=seed, trans=float(rng.choice([0.0, 0.03, 0.1])), rot_deg=float(rng.choice([0.0, 1.0, 4.0])),
                                                            depth_f32=bool(rng.random() < 0.3))
    reg = RegisterPhotoICP(); reg.setNumPyr(n_pyr); reg.setMaskSeams(False); reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    ora = O.Oracle(n_pyr=n_pyr, math_mode=1, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    kind = int(rng.integers(0, 5))
    pose = np.array(T, dtype=np.float64)
    if kind == 1:                                   # zoom out
        P = np.eye(4); P[2, 3] = float(rng.uniform(0.3, 3.0)); pose = P @ pose
    elif kind == 2:                                 # half turn about the image's vertical axis + a push: points behind the camera
        P = synth.make_pose(synth.rodrigues(np.array([0.0, 1.0, 0.0]), np.pi + float(rng.normal() * 0.05)), np.array([0.0, 0.0, float(rng.uniform(-1.0, 1.0))])); pose = P @ pose
    elif kind == 3:                                 # collapse
        P = np.eye(4); P[2, 3] = float(rng.uniform(50.0, 500.0)); pose = P @ pose
    elif kind == 4:                                 # a random nearby pose
        pose = synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.05), rng.normal(size=3) * 0.1) @ pose
    level = int(rng.integers(0, n_pyr))
    method, occ = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    e = reg.eval_pinhole(level, pose, method, occ)
    _, sp, sd, n_p, n_d = ora.error_pinhole_occ(level, pose, method, occ)
    H6, g6, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pose, method, occ)
    idx = ora.warp_indices_pinhole(level, pose)
    v = idx[:, 0] >= 0
    longest = int(np.unique(idx[v, 0] * 65536 + idx[v, 1], return_counts=True)[1].max()) if v.any() else 0
    # (a point behind the camera has a negative depth, hence a negative Huber scale sigma * Z: its weight is the square root of a negative
    # number on both sides -- NaN sums agree when both are NaN)
    def close(a, b, tol):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return bool(np.array_equal(np.isnan(a), np.isnan(b)) and (np.abs(np.nan_to_num(a) - np.nan_to_num(b)) <= tol).all())
    ok = list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nvis
    ok = ok and close(e["err2_split"][0], sp, 2e-6 * max(1.0, abs(np.nan_to_num(sp)))) and close(e["err2_split"][1], sd, 2e-6 * max(1.0, abs(np.nan_to_num(sd))))
    sh = max(np.nanmax(np.abs(Hd)) if np.isfinite(Hd).any() else 0.0, 1e-30)
    gmax = np.nanmax(np.abs(gd)) if np.isfinite(gd).any() else 0.0
    ok = ok and close(e["H64"], Hd, 2e-5 * sh) and close(e["g64"], gd, 2e-5 * max(gmax, 1e-3 * np.sqrt(sh)))
    bad += 0 if ok else 1
    if not ok:
        print("   sums", e["err2_split"], sp, sd, "H", np.abs(np.nan_to_num(e["H64"]) - np.nan_to_num(Hd)).max(), sh, "nan H", int(np.isnan(e["H64"]).sum()), int(np.isnan(Hd).sum()))
    print("trial %2d: %3dx%-3d level %d method %d occ %d pose kind %d: visible %6d longest list %5d  counts %s / %s  numVisible %d / %d  %s" % (
        t, W, H, level, method, occ, kind, int(v.sum()), longest, list(e["n_split"]), [n_p, n_d], e["n_rows"], nvis, "ok" if ok else "MISMATCH"))
    reg.close()
print("%d trials, %d mismatches" % (n_trials, bad))
sys.exit(1 if bad else 0)
