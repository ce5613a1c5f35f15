"""RegisterRGBD360::RegisterDensePhotoICP (RegisterRGBD360.h:344-520): the 8-sensor dense registration with
the reference's three defects fixed (oracle/photo_icp_ref.cpp lists them).  CPU: the oracle restatement recovers a known rig motion
(which the function as written in the reference cannot: it never accepts a step).  GPU: the fused all-sensor pass and the
Levenberg-Marquardt driver of csrc/rig_dense.h against that oracle."""
import numpy as np
import pytest

from rgbd360_amd import synth

ROT_TOL, TRANS_TOL = 1e-4, 1e-3              # north-star pose tolerance
HG_RTOL, ERR2_RTOL = 2e-5, 2e-6
POSE_TOL_DEV = (5e-5, 2e-4)                  # device-arithmetic oracle: float32 weight rounding on a pinhole problem (as test_pinhole_*)


@pytest.fixture(scope="module")
def rig_pair():
    return synth.make_rig_pair(160, 120, seed=3, trans=0.04, rot_deg=1.5)


def _oracle(oracle_mod, rig_pair, mm, n_pyr=3):
    f1, f2, M, Rt, K = rig_pair
    rig = oracle_mod.RigOracle(Rt, K, n_pyr=n_pyr, math_mode=mm[0], reduce_mode=mm[1])
    for s in range(len(Rt)):
        rig.set_frame(s, True, *f1[s])
        rig.set_frame(s, False, *f2[s])
    return rig


@pytest.mark.parametrize("method", [0, 1, 2])
def test_oracle_rig_registration_recovers_the_motion(oracle_mod, rig_pair, method):
    """With fixes A-C the function converges to the rig's true motion from the identity (160x120 sensors, 3 levels); both arithmetic
    modes walk the same accept / reject sequence."""
    M = rig_pair[2]
    traces = []
    for mm in ((0, 0), (1, 1)):
        rig = _oracle(oracle_mod, rig_pair, mm)
        st, pose = rig.align(np.eye(4), method)
        assert st == 0 and sum(rig.iters) >= 3
        rot, trans = synth.pose_error(pose, M)
        assert rot < 5e-4 and trans < 2e-3, (mm, rot, trans)
        H = rig.hessian
        assert np.allclose(H, H.T, rtol=1e-4, atol=1e-3 * np.abs(H).max()) and np.all(np.linalg.eigvalsh(H.astype(np.float64)) > 0)
        traces.append([(t[0], t[1], t[2]) for t in rig.trace()])
    assert traces[0] == traces[1]


def test_oracle_rig_identical_frames_stay_at_identity(oracle_mod, rig_pair):
    f1, _, _, Rt, K = rig_pair
    rig = oracle_mod.RigOracle(Rt, K, n_pyr=3, math_mode=1, reduce_mode=1)
    for s in range(len(Rt)):
        rig.set_frame(s, True, *f1[s])
        rig.set_frame(s, False, *f1[s])
    e, sums = rig.error(0, np.eye(4), 2)
    # every valid pixel warps onto itself; the depth residual is the float32 rounding of Rt^-1 (Rt p) only
    assert sums[0] == 0.0 and e < 1e-6 and sums[2] > 1000 and sums[3] > 1000
    st, pose = rig.align(np.eye(4), 2)
    assert st == 0 and np.array_equal(pose, np.eye(4, dtype=np.float32)) and rig.iters == [0, 0, 0]


def test_oracle_rig_sensor_sum_is_order_independent_in_double(oracle_mod, rig_pair):
    """The 8 per-sensor normal equations live in ONE coordinate system (the rig's): H64 of the rig = sum of H64 of 8 one-sensor rigs."""
    f1, f2, M, Rt, K = rig_pair
    full = _oracle(oracle_mod, rig_pair, (1, 1))
    H, g, Hd, gd, n = full.hessgrad(1, M, 2)
    Hs, ns = np.zeros((6, 6)), 0
    for s in range(len(Rt)):
        one = oracle_mod.RigOracle([Rt[s]], K, n_pyr=3, math_mode=1, reduce_mode=1)
        one.set_frame(0, True, *f1[s])
        one.set_frame(0, False, *f2[s])
        _, _, Hd1, _, n1 = one.hessgrad(1, M, 2)
        Hs += Hd1
        ns += n1
    assert ns == n and np.allclose(Hs, Hd, rtol=1e-12, atol=1e-9 * np.abs(Hd).max())


def _gpu_rig(rig_pair, n_pyr=3):
    from rgbd360_amd.rig import RegisterDensePhotoICP
    f1, f2, M, Rt, K = rig_pair
    reg = RegisterDensePhotoICP(Rt, K, n_pyr=n_pyr)
    reg.setTargetFrame(f1)
    reg.setSourceFrame(f2)
    return reg


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_rig_eval_parity(hip_lib, oracle_mod, rig_pair, method):
    """The fused all-sensor pass: pixel counts and Jacobian-row counts exact, error sums and the summed normal equations to float32
    rounding, at the identity, the true motion and a perturbed pose, on every level."""
    M = rig_pair[2]
    reg = _gpu_rig(rig_pair)
    ora = _oracle(oracle_mod, rig_pair, (1, 1))
    rng = np.random.default_rng(4)
    poses = [np.eye(4), M, synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.03), rng.normal(size=3) * 0.03)]
    for level in range(3):
        for T in poses:
            e = reg.eval(level, T, method)
            err, sums = ora.error(level, T, method)
            H, g, Hd, gd, n = ora.hessgrad(level, T, method)
            assert list(e["n_split"]) == [int(sums[2]), int(sums[3])] and e["n_rows"] == n, (level, e["n_split"], sums, e["n_rows"], n)
            assert abs(e["err2"] - err) <= ERR2_RTOL * max(err, 1.0)
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.abs(Hd).max())


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_rig_align_matches_oracle(hip_lib, oracle_mod, rig_pair, method):
    M = rig_pair[2]
    reg = _gpu_rig(rig_pair)
    assert reg.align(np.eye(4), method)
    ora = _oracle(oracle_mod, rig_pair, (1, 1))
    st, pose_ref = ora.align(np.eye(4), method)
    assert st == 0 and reg.num_iterations == ora.iters          # same accept / reject / LM-retry sequence
    rot, trans = synth.pose_error(reg.getPose(), pose_ref)
    assert rot <= POSE_TOL_DEV[0] and trans <= POSE_TOL_DEV[1], (rot, trans)
    assert np.allclose(reg.getInfoMat(), ora.hessian, rtol=1e-3, atol=1e-4 * np.abs(ora.hessian).max())
    # the reference-faithful arithmetic (libm rounding, double projection, float accumulators): the north-star tolerance
    ora0 = _oracle(oracle_mod, rig_pair, (0, 0))
    st0, pose0 = ora0.align(np.eye(4), method)
    rot, trans = synth.pose_error(reg.getPose(), pose0)
    assert st0 == 0 and rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)
    rot, trans = synth.pose_error(reg.getPose(), M)
    assert rot < 5e-4 and trans < 2e-3, (rot, trans)
    # bitwise reproducible (fixed-order reductions, no atomics)
    p1 = reg.getPose()
    assert reg.align(np.eye(4), method) and np.array_equal(p1, reg.getPose())


@pytest.mark.gpu
def test_rig_salient_pixel_list_mode(hip_lib, oracle_mod, rig_pair):
    """useSaliency(true) on the per-sensor objects (RPI.h:4930-5003, 5121-5262): both passes over vSalientPixels only -- counts exact,
    fewer pixels than the full pass, same accept / reject sequence and pose as the oracle; switching it off restores the full pass."""
    M = rig_pair[2]
    reg = _gpu_rig(rig_pair)
    ora = _oracle(oracle_mod, rig_pair, (1, 1))
    full = reg.eval(1, M, 2)
    reg.useSaliency(True)
    ora.use_saliency(True, 0.01)
    for level in range(3):
        for T in (np.eye(4), M):
            for method in (0, 1, 2):
                e = reg.eval(level, T, method)
                err, sums = ora.error(level, T, method)
                H, g, Hd, gd, n = ora.hessgrad(level, T, method)
                assert list(e["n_split"]) == [int(sums[2]), int(sums[3])] and e["n_rows"] == n
                assert abs(e["err2"] - err) <= ERR2_RTOL * max(err, 1.0)
                assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
    sal = reg.eval(1, M, 2)
    assert 0 < sal["n_split"][0] < full["n_split"][0] and sal["n_rows"] < full["n_rows"]
    assert reg.align(np.eye(4), 2)
    st, pose_ref = ora.align(np.eye(4), 2)
    assert st == 0 and reg.num_iterations == ora.iters
    rot, trans = synth.pose_error(reg.getPose(), pose_ref)
    assert rot <= POSE_TOL_DEV[0] and trans <= POSE_TOL_DEV[1], (rot, trans)
    reg.useSaliency(False)
    assert list(reg.eval(1, M, 2)["n_split"]) == list(full["n_split"])


@pytest.mark.gpu
def test_rig_full_size_sensors_and_float_depth(hip_lib, oracle_mod):
    """The rig's real geometry: eight 320x240 sensors, 4 levels, a 5 cm / 2 degree motion; float32 depth images give the same pose
    as the millimetre ones they were converted from."""
    pair = synth.make_rig_pair(320, 240, seed=9, trans=0.05, rot_deg=2.0)
    f1, f2, M, Rt, K = pair
    reg = _gpu_rig(pair, n_pyr=4)
    assert reg.align(np.eye(4), 2)
    rot, trans = synth.pose_error(reg.getPose(), M)
    assert rot < 3e-4 and trans < 1.5e-3, (rot, trans)
    ora = _oracle(oracle_mod, pair, (1, 1), n_pyr=4)
    st, pose_ref = ora.align(np.eye(4), 2)
    assert st == 0 and reg.num_iterations == ora.iters
    rot, trans = synth.pose_error(reg.getPose(), pose_ref)
    assert rot <= POSE_TOL_DEV[0] and trans <= POSE_TOL_DEV[1], (rot, trans)
    from rgbd360_amd.rig import RegisterDensePhotoICP
    regf = RegisterDensePhotoICP(Rt, K, n_pyr=4)
    regf.setTargetFrame([(a, d.astype(np.float32) * np.float32(0.001)) for a, d in f1])
    regf.setSourceFrame([(a, d.astype(np.float32) * np.float32(0.001)) for a, d in f2])
    assert regf.align(np.eye(4), 2) and np.array_equal(regf.getPose(), reg.getPose())


@pytest.mark.gpu
def test_rig_argument_errors_and_ill_posed(hip_lib, rig_pair):
    from rgbd360_amd.register import Rgbd360Error
    from rgbd360_amd.rig import RegisterDensePhotoICP
    f1, f2, M, Rt, K = rig_pair
    with pytest.raises(Rgbd360Error):
        RegisterDensePhotoICP(Rt + [Rt[0]], K)                     # more than 8 sensors
    reg = RegisterDensePhotoICP(Rt, K, n_pyr=3)
    with pytest.raises(Rgbd360Error):
        reg.align(np.eye(4), 0)                                    # no frames yet
    with pytest.raises(Rgbd360Error):
        reg.setTargetFrame(f1[:5])
    # no valid pixel anywhere: the error is 0, the loop of RegisterRGBD360.h:414 never runs, the guess comes back as "registered"
    blank = [(np.zeros_like(a), np.zeros_like(d)) for a, d in f1]
    reg.setTargetFrame(blank)
    reg.setSourceFrame(blank)
    assert reg.align(np.eye(4), 2) and reg.num_iterations == [0, 0, 0]
    assert np.array_equal(reg.getPose(), np.eye(4, dtype=np.float32))
    # a residual without any image gradient (two different flat grey levels, photometric only): H = 0 -> "The problem is ILL-POSED",
    # the pose reached so far (the guess) is returned with status 1 (RegisterRGBD360.h:443-449)
    flat1 = [(np.full_like(a, 60), d) for a, d in f1]
    flat2 = [(np.full_like(a, 200), d) for a, d in f1]
    reg.setTargetFrame(flat1)
    reg.setSourceFrame(flat2)
    guess = synth.make_pose(synth.rodrigues(np.array([0.0, 0.0, 1.0]), 0.01), np.array([0.01, 0.0, 0.0])).astype(np.float32)
    assert not reg.align(guess, 0) and reg.status == 1
    assert np.array_equal(reg.getPose(), guess)
