"""Soak check of the 8-sensor dense registration (rgbd360_rig_*: RegisterRGBD360::RegisterDensePhotoICP with the reference's three
defects fixed, RegisterRGBD360.h:344-520) against the CPU oracle: random sensor sizes, pyramid depths, methods, motions, guesses, depth
types, 4 / 8 sensors.  Against the oracle in the device's arithmetic: same status, the same accept / reject / LM-retry sequence, pose
within 5e-5 rad / 2e-4 m (the pinhole tests' tolerance); against the reference-faithful arithmetic (libm rounding, double projection,
float32 accumulators) the pose is reported with the count inside 1e-4 rad / 1e-3 m.  python tests/tools/rig_dense_soak.py [n_trials [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.rig import RegisterDensePhotoICP
from oracle import oracle as O
O.set_num_threads(min(16, os.cpu_count() or 1))
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 41)      # [seed]: another draw of cases
bad = n00 = nseq = coin = 0
for t in range(n_trials):
    W, H = [(80, 60), (120, 88), (160, 120), (240, 180), (320, 240)][int(rng.integers(0, 5))]
    n_pyr = int(rng.integers(1, 4))
    method = int(rng.integers(0, 3))
    trans = float(rng.choice([0.0, 0.02, 0.04, 0.1]))
    rot = float(rng.choice([0.0, 0.5, 1.5, 4.0]))
    f32 = bool(rng.random() < 0.3)
    n_sens = int(rng.choice([4, 8, 8]))
    f1, f2, M, Rt, K = synth.make_rig_pair(W, H, seed=int(rng.integers(0, 1000)), trans=trans, rot_deg=rot, n_sensors=n_sens, depth_f32=f32)
    guess = np.eye(4)
    if rng.random() < 0.4:
        guess = synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.005), rng.normal(size=3) * 0.005)
    reg = RegisterDensePhotoICP(Rt, K, n_pyr=n_pyr)
    reg.setTargetFrame(f1)
    reg.setSourceFrame(f2)
    ok = reg.align(guess, method)
    rigs = {}
    for mm in ((1, 1), (0, 0)):
        rig = O.RigOracle(Rt, K, n_pyr=n_pyr, math_mode=mm[0], reduce_mode=mm[1])
        for s in range(len(Rt)):
            rig.set_frame(s, True, *f1[s])
            rig.set_frame(s, False, *f2[s])
        rigs[mm] = (rig,) + tuple(rig.align(guess, method))
    rig1, st1, pose1 = rigs[(1, 1)]
    rig0, st0, pose0 = rigs[(0, 0)]
    r1, t1 = synth.pose_error(reg.getPose(), pose1)
    same = bool(ok) == (st1 == 0) and list(reg.num_iterations) == list(rig1.iters) and r1 <= 5e-5 and t1 <= 2e-4
    r0, t0 = synth.pose_error(reg.getPose(), pose0)
    in00 = (st0 == 0) == bool(ok) and list(rig0.iters) == list(reg.num_iterations) and r0 <= 1e-4 and t0 <= 1e-3
    n00 += 1 if in00 else 0
    nseq += 1 if list(rig0.iters) != list(reg.num_iterations) else 0
    note = ""
    if bool(ok) == (st1 == 0) and list(reg.num_iterations) != list(rig1.iters) and r1 <= 5e-5 and t1 <= 2e-4:
        # another sequence, the same pose: a level's loop ends when a step improves the summed error by no more than tol_residual = 0.1
        # (absolute, RegisterRGBD360.h:383) or not at all -- how close did a step of the parting level come?
        k = next(i for i in range(n_pyr) if list(reg.num_iterations)[i] != list(rig1.iters)[i])
        steps = [x for x in rig1.trace() if x[1] >= 0 and x[0] == k]          # (iters[] is indexed by level, 0 = the finest)
        marg = min([min(abs((x[3] - x[4]) - 0.1), abs(x[3] - x[4])) / max(x[3], 1e-12) for x in steps] + [float("inf")])
        note = " (another sequence, same pose: a step of the parting level came within %.1e of the error of a stop threshold; steps (error -> new): %s)" % (
            marg, ", ".join("%.6g->%.6g%s" % (x[3], x[4], "" if x[2] else " refused") for x in steps[:8]))
        same = marg < 1e-5
        coin += 1 if same else 0
    bad += 0 if same else 1
    print("trial %2d: %d x %3dx%-3d n_pyr %d method %d motion %.2f m / %.1f deg %s guess %s -> ok %d / status %d iters %s / %s, device-arithmetic oracle %.1e rad %.1e m; "
          "reference arithmetic: iters %s, %.1e rad %.1e m -> %s" % (t, n_sens, W, H, n_pyr, method, trans, rot, "float32" if f32 else "uint16",
                                                                     "yes" if not np.array_equal(guess, np.eye(4)) else "no", bool(ok), st1, list(reg.num_iterations),
                                                                     list(rig1.iters), r1, t1, list(rig0.iters), r0, t0, ("ok" if same else "FAIL") + note), flush=True)
print("rig dense soak: %d / %d trials ok against the device-arithmetic oracle (%d of them with a coin-toss step: another sequence, same pose); inside 1e-4 rad / 1e-3 m with the same sequence against the reference arithmetic: %d; "
      "another sequence there: %d" % (n_trials - bad, n_trials, coin, n00, nseq))
sys.exit(1 if bad else 0)
