"""One draw of tests/tools/pinhole_align_soak.py in detail: at every iterate of the device-arithmetic oracle's trace, the device's
normal equations against the oracle's (relative difference, condition number of H), and the device's pose against the oracle's after
each level.  python tests/tools/pinhole_case.py SEED TRIAL"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
from oracle import oracle as O
O.set_num_threads(min(16, os.cpu_count() or 1))
rng = np.random.default_rng(int(sys.argv[1]))
want = int(sys.argv[2])
for t in range(want + 1):
    W, H = [(160, 120), (200, 152), (320, 240), (480, 360), (640, 480)][int(rng.integers(0, 5))]
    n_pyr = int(rng.integers(1, 4))
    method = int(rng.integers(1, 3))
    occlusion = int(rng.choice([0, 0, 1, 2]))
    if occlusion == 1:
        method = 2
    trans = float(rng.choice([0.0, 0.01, 0.03, 0.08]))
    rot = float(rng.choice([0.0, 0.5, 1.0, 3.0]))
    f32 = bool(rng.random() < 0.4)
    seed = int(rng.integers(0, 1000))
    g_draw = rng.random() < 0.4
    guess = np.eye(4)
    if g_draw:
        guess = synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.005), rng.normal(size=3) * 0.005)
print("draw: %dx%d n_pyr %d method %d occlusion %d motion %.2f / %.1f %s seed %d" % (W, H, n_pyr, method, occlusion, trans, rot, "f32" if f32 else "u16", seed))
(rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(W, H, seed=seed, trans=trans, rot_deg=rot, depth_f32=f32)
reg = RegisterPhotoICP(); reg.setNumPyr(n_pyr); reg.setMaskSeams(False)
reg.setCameraMatrix(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]]))
reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
ora = O.Oracle(n_pyr=n_pyr, math_mode=1, reduce_mode=1, mask_seams=0)
ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
rc = reg.alignFrames(guess, method, occlusion)
st, pose_ref = ora.align_pinhole(guess, method, occlusion)
print("status %d / %d, iters %s / %s, pose diff %.2e rad %.2e m; vs ground truth: device %s oracle %s" % (
    rc, st, list(reg.num_iterations), list(ora.result.iters)[:n_pyr], *synth.pose_error(reg.getOptimalPose(), pose_ref),
    "%.2e rad %.2e m" % synth.pose_error(reg.getOptimalPose(), T), "%.2e rad %.2e m" % synth.pose_error(pose_ref, T)))
for x in ora.trace():
    pose = x["pose"]
    print("oracle step: level %d it %d update %s" % (x["level"], x["it"], " ".join("%.9g" % v for v in x["update"])))
    e = reg.eval_pinhole(x["level"], pose, method, occlusion)
    if occlusion:
        Hs, gs, Hd, gd, nv = ora.hessgrad_pinhole_occ(x["level"], pose, method, occlusion)
    else:
        Hs, gs, Hd, gd, nv = ora.hessgrad_pinhole(x["level"], pose, method)
    dH = np.abs(e["H64"] - Hd).max() / np.abs(Hd).max()
    dg = np.abs(e["g64"] - gd).max() / max(np.abs(gd).max(), 1e-30)
    cond = np.linalg.cond(Hd)
    step_o = np.linalg.solve(Hd, -gd)
    step_d = np.linalg.solve(e["H64"], -e["g64"])
    print("level %d it %2d %s error %.8f -> %.8f: rows %d / %d, |dH|/|H| %.1e |dg|/|g| %.1e, cond(H) %.1e, Gauss-Newton step of the two systems differs by %.1e (relative %.1e)" % (
        x["level"], x["it"], "accepted" if x["accepted"] else "refused ", x["error"], x["new_error"], e["n_rows"], nv, dH, dg, cond,
        np.abs(step_o - step_d).max(), np.abs(step_o - step_d).max() / max(np.abs(step_o).max(), 1e-30)))
