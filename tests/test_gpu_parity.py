"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer / index work bit-exact; float32 planes bit-exact (same operation order);
normal equations within float32 rounding of the float64-accumulated oracle; final SE(3) pose within
1e-4 rad / 1e-3 m of the oracle.
"""
import math

import os
import numpy as np
import pytest

from rgbd360_amd import synth

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-4, 1e-3     # north_star pose tolerance (rad, m)
HG_RTOL = 2e-5                      # H,g: float32 rows, FMA vs mul+add rounding, float64 accumulation on both sides
POSE_TOL_DEV = 5e-6                 # GPU vs oracle in device-arithmetic mode: same pixels, float32 Jacobian rounding only
ERR2_RTOL = 2e-6                    # sum of squared residuals: float32 weights (hardware sqrt/rcp) vs the oracle's
PINHOLE_ROT_TOL_DEV, PINHOLE_TRANS_TOL_DEV = 5e-5, 2e-4     # the 320x240 narrow-FOV problem is worse conditioned (rotation /
                                    # translation trade off): 1e-7 relative rounding differences of the float32 weights move the
                                    # LM solution by ~2e-5 rad / 6e-5 m (observed when one mul+sub became an fma); north star: 1e-4 / 1e-3


def _mk(hip_lib, n_pyr=3, **kw):
    from rgbd360_amd.register import RegisterPhotoICP
    r = RegisterPhotoICP()
    r.setNumPyr(n_pyr)
    for k, v in kw.items():
        getattr(r, k)(v)
    return r


def _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=3, math_mode=1):
    (rgbA, dA), (rgbB, dB), T = pair
    reg = _mk(hip_lib, n_pyr)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=n_pyr, math_mode=math_mode, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    return reg, ora, T


def _assert_libm_oracle_agrees(reg, ora, method, n_pyr, occlusion=0, status=0):
    """The reference-faithful oracle (math_mode 0: libm asinf / atan2f / roundf as RPI.h:2674-2680 writes them; reduce_mode 0:
    the reference's float32 accumulators) against the pose the device has just produced: the north-star tolerance and the
    same accept / reject sequence (iterations per level)."""
    pose_gpu, iters_gpu = reg.getOptimalPose(), list(reg.num_iterations)
    ora.set_modes(0, 0)
    st, pose_libm = ora.align360(np.eye(4), method, occlusion)
    assert st == status
    assert iters_gpu == list(ora.result.iters)[:n_pyr], (iters_gpu, list(ora.result.iters)[:n_pyr])
    rot, trans = synth.pose_error(pose_gpu, pose_libm)
    assert rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)
    ora.set_modes(1, 1)      # back to the device-arithmetic mode the exact-count checks use


def _poses(T_gt):
    rng = np.random.default_rng(5)
    out = [np.eye(4), np.asarray(T_gt)]
    for _ in range(2):
        out.append(synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.05), rng.normal(size=3) * 0.05))
    return out


def test_device_sqrt_and_reciprocal_are_correctly_rounded(hip_lib):
    """The warp front end relies on sqrt_rn / rcp_rn being IEEE correctly rounded (that is what makes the CPU oracle's
    sqrtf and 1.f/x bit-identical), and on v_cvt_rpi_i32_f32 being floor(x + 0.5) with an exact sum.
    Exhaustive over every float in [2^-60, 2^60] (1.0e9 values; the rounding check covers |x| < 1e9, both signs)."""
    reg = _mk(hip_lib, 3)
    first, last = 0x21800000, 0x5D800000
    bad_sqrt, bad_rcp, bad_round = reg.selftest_math(first, last - first)
    assert bad_sqrt == 0 and bad_rcp == 0 and bad_round == 0, (bad_sqrt, bad_rcp, bad_round)


def test_planes_bit_exact(hip_lib, oracle_mod, small_pair):
    reg, ora, _ = _pair_ctx(hip_lib, oracle_mod, small_pair)
    for level in range(3):
        ora.prepare_level(level)      # applies the seam mask like alignFrames360 does on entering a level
        for name in ("gray_src", "gray_trg", "depth_src", "depth_trg", "gx", "gy", "dgx", "dgy"):
            a, b = reg.plane(name, level), ora.plane(name, level)
            assert a.shape == b.shape
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, level, np.abs(a - b).max())
        la, lb = reg.lut(level), ora.lut(level)
        valid = lb[:, 0] != -10000
        assert np.array_equal(la[:, 0] != -10000, valid)
        assert np.array_equal(la[valid].view(np.uint32), lb[valid].view(np.uint32)), level


def test_warp_indices_bit_exact(hip_lib, oracle_mod, small_pair):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    for level in range(3):
        for pose in _poses(T):
            a, b = reg.warp_indices(level, pose), ora.warp_indices(level, pose)
            assert np.array_equal(a, b), (level, int((a != b).any(axis=1).sum()))


INDEX_FLIP_BOUND = 2e-4             # warped index, device arithmetic vs the reference's asinf / atan2f / roundf: fraction of valid pixels


@pytest.mark.parametrize("W,H", [(2048, 1024), (4096, 2048)])
def test_warp_indices_against_the_reference_arithmetic_full_size(hip_lib, oracle_mod, W, H):
    """Index parity with the REFERENCE's arithmetic (oracle math_mode 0: Eigen-order rotation, asinf / atan2f + double PI / roundf,
    RPI.h:2663-2684), not with the oracle's mirror of the device's (math_mode 1, test_warp_indices_bit_exact), at the sizes of
    BASELINE.json configs[1]/[2] and configs[4], at four poses.  The two definitions agree except where a float32 rounding
    difference of the angle (1e-7 rad) straddles a pixel boundary: the differing fraction grows with pixels per radian
    (measured 8e-5 at 2048 x 1024, 1.6e-4 at 4096 x 2048, all of it present between the oracle's own two modes) and is bounded
    here, so that a cheaper device atan2 cannot raise it unnoticed.  Every difference is one pixel in exactly one coordinate;
    visibility differs only where the reference's index falls one step outside the image border."""
    pair = synth.make_pair(W, H, seed=1234)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=1, math_mode=0)
    ora.set_modes(0, 0)
    worst = 0.0
    for pose in _poses(T):
        a, b = reg.warp_indices(0, pose), ora.warp_indices(0, pose)
        vis_a, vis_b = a[:, 0] >= 0, b[:, 0] >= 0
        valid = vis_a | vis_b
        diff = (a != b).any(axis=1)
        frac = diff.sum() / max(1, valid.sum())
        worst = max(worst, frac)
        assert frac <= INDEX_FLIP_BOUND, (W, H, int(diff.sum()), int(valid.sum()), frac)
        both = diff & vis_a & vis_b
        assert (np.abs(a[both] - b[both]).sum(axis=1) == 1).all()
        for only, idx in ((vis_a & ~vis_b, a), (vis_b & ~vis_a, b)):      # visible on one side only: a border pixel on that side
            rc = idx[only]
            assert ((rc[:, 0] == 0) | (rc[:, 0] == H - 1) | (rc[:, 1] == 0) | (rc[:, 1] == W - 1)).all(), rc[:8]
        assert (vis_a ^ vis_b).sum() <= 1e-5 * valid.sum()
    # at the identity every point lands on its own pixel centre in both definitions
    a, b = reg.warp_indices(0, np.eye(4)), ora.warp_indices(0, np.eye(4))
    assert np.array_equal(a, b)
    print(f"index flips vs the reference arithmetic at {W}x{H}: worst fraction {worst:.3e}")


@pytest.mark.parametrize("method", [0, 1, 2])
def test_eval_parity(hip_lib, oracle_mod, small_pair, method):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    for level in range(3):
        for pose in _poses(T):
            e = reg.eval(level, pose, method)
            rms, err2, nvalid = ora.error(level, pose, method)
            H, g, Hd, gd, nvis = ora.hessgrad(level, pose, method)
            assert e["n_valid"] == nvalid            # integer work: exact
            assert e["n_visible"] == nvis
            assert abs(e["err2"] - err2) <= ERR2_RTOL * max(1.0, abs(err2)), (e["err2"], err2)
            scale_h = np.abs(Hd).max()
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * scale_h
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.sqrt(scale_h))
            assert np.allclose(e["H"], e["H64"].astype(np.float32))


def test_gn_step_matches_oracle(hip_lib, oracle_mod, small_pair):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    H, g, Hd, gd, _ = ora.hessgrad(1, np.eye(4), 2)
    st_o, pose_o, upd_o = oracle_mod.gn_step(H, g, 1.0, np.eye(4))
    st_d, pose_d, upd_d = reg.gn_step(H, g, 1.0, np.eye(4))
    assert st_o == st_d == 0
    assert np.allclose(upd_d, upd_o, rtol=1e-5, atol=1e-8)
    assert np.allclose(pose_d, pose_o, rtol=0, atol=1e-7)
    # rank-deficient system: both report ILL-POSED (RPI.h:4682-4690)
    Hbad = H.copy()
    Hbad[:, 5] = Hbad[:, 4]
    Hbad[5, :] = Hbad[4, :]
    assert oracle_mod.gn_step(Hbad, g, 1.0, np.eye(4))[0] == 1
    assert reg.gn_step(Hbad, g, 1.0, np.eye(4))[0] == 1


@pytest.mark.parametrize("method", [0, 1, 2])
def test_align_small_matches_oracle(hip_lib, oracle_mod, small_pair, method):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    rc = reg.alignFrames360(np.eye(4), method)
    st, pose_ref = ora.align360(np.eye(4), method)
    assert rc == st == 0
    assert reg.num_iterations == list(ora.result.iters)[:3]     # same accept / reject sequence
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    assert abs(reg.avResidual - ora.result.err_final) <= ERR2_RTOL * max(1.0, ora.result.err_final)
    assert np.allclose(reg.getHessian(), np.asarray(list(ora.result.hessian)).reshape(6, 6).T, rtol=1e-4,
                       atol=1e-4 * np.abs(reg.getHessian()).max())
    assert abs(reg.SSO - ora.result.sso) < 1e-6
    # and against the reference-faithful libm oracle: the north-star tolerance
    ora.set_modes(0, 0)
    st, pose_libm = ora.align360(np.eye(4), method)
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_libm)
    assert rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)


def test_forced_schedule_matches_oracle(hip_lib, oracle_mod, small_pair):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    n = 4
    out = reg.forced_iters(0, np.eye(4), 2, n + 1)      # n+1 fused passes apply n steps
    e_ref, pose_ref = ora.forced_iters(0, np.eye(4), 2, n)
    rot, trans = synth.pose_error(out["pose"], pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    assert abs(out["rms"] - e_ref) <= 1e-5


def _partial_row(Hdiag, g, e2=(3.0, 2.0), n=(1000, 800, 1500)):
    row = np.zeros(32)
    for a, h in enumerate(Hdiag):
        row[(a * (13 - a)) // 2] = h          # upper-triangle slot of H(a, a)
    row[21:27] = g
    row[27:29] = e2
    row[29:32] = n
    return row


def test_solve_paths_fused_and_two_launch_agree(hip_lib, oracle_mod, small_pair):
    """The serial part of a Gauss-Newton iteration (RPI.h:4611-4722) on hand-made normal equations, through k_solve and through the
    prologue of the fused launch (which commits the step on the inverse's word and lets one wave of block 0 deliver the rank
    verdict afterwards): an ordinary step, ILL-POSED by the rank test alone ((H + lambda diag H).rank() < 6 with every LU pivot
    non-zero, RPI.h:4682-4690), ILL-POSED by a zero pivot, and no valid pixels.  Same status / flags / candidate / update, bit for
    bit, and the ordinary step equals gn::step on the host (through the oracle)."""
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    g = np.array([0.3, -0.2, 0.1, 0.05, -0.04, 0.02])
    cases = {"step": (_partial_row([4, 5, 6, 7, 8, 9], g), 0, 0),
             "rank": (_partial_row([1, 1, 1, 1, 1, 1e-9], g), 1, 1),
             "pivot": (_partial_row([1, 1, 1, 0, 1, 1], g), 1, 1),
             "empty": (_partial_row([4, 5, 6, 7, 8, 9], g, n=(0, 0, 0)), 2, 1)}
    for name, (row, status, done) in cases.items():
        for level in (0, 2):
            two = reg.debug_solve_partials(level, row, 2, fused=False)
            one = reg.debug_solve_partials(level, row, 2, fused=True)
            assert two["status"] == one["status"] == status, (name, level, two, one)
            assert two["done"] == one["done"] == done, (name, level, two, one)
            for key in ("level_active", "it", "n_evals"):
                assert two[key] == one[key], (name, level, key, two, one)
            assert np.array_equal(two["cand"], one["cand"]) and np.array_equal(two["update"], one["update"]), (name, level)
            assert (one["pend_nb"] > 0) == (not done) and two["pend_nb"] == 0, (name, one, two)       # an accepted step's pass stays pending
            if name == "step":
                st, pose_o, upd_o = oracle_mod.gn_step(np.diag([4, 5, 6, 7, 8, 9.0]).astype(np.float32), g.astype(np.float32), 1.0, np.eye(4))
                assert st == 0 and np.allclose(one["update"], upd_o, rtol=1e-5, atol=1e-8) and np.allclose(one["cand"], pose_o, atol=1e-7)
            else:
                assert np.array_equal(one["cand"], np.eye(4, dtype=np.float32)) and np.array_equal(one["update"], np.ones(6, np.float32))


def test_fused_solve_checks_the_hosts_bound_on_pending_rows(hip_lib):
    """The fused launch requests the pending pass's partial rows before it knows how many there are; the host passes a bound so that the
    coarse levels request 32 rows instead of 256.  The device checks that bound against the state: a pending row beyond the first batch
    behind a launch that was told `one row` is fetched in a second trip -- same step as with the row in place 0, not a silent zero."""
    from rgbd360_amd.register import RegisterPhotoICP
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(1024, 512, seed=3)          # level 0: 64 block rows
    reg = RegisterPhotoICP()
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    row = _partial_row([4, 5, 6, 7, 8, 9], np.array([0.3, -0.2, 0.1, 0.05, -0.04, 0.02]))
    first = reg.debug_solve_partials(0, row, 2, fused=1)
    late = reg.debug_solve_partials(0, row, 2, fused=2)
    assert first["status"] == late["status"] == 0 and first["done"] == late["done"] == 0
    assert np.array_equal(first["update"], late["update"]) and np.array_equal(first["cand"], late["cand"])
    assert np.abs(first["update"]).max() > 1e-3


def test_float_depth_and_strided_inputs(hip_lib, oracle_mod):
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(256, 128, seed=77, depth_f32=True)
    # row-padded (strided) host images, like a cv::Mat ROI
    padA = np.zeros((128, 300, 3), np.uint8)
    padA[:, :256] = rgbA
    padD = np.zeros((128, 290), np.float32)
    padD[:, :256] = dA
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(padA[:, :256], padD[:, :256])
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=1, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    assert np.array_equal(reg.plane("depth_trg", 0), ora.plane("depth_trg", 0))
    assert np.array_equal(reg.plane("gray_trg", 1), ora.plane("gray_trg", 1))
    rc = reg.alignFrames360(np.eye(4), 2)
    st, pose_ref = ora.align360(np.eye(4), 2)
    assert rc == st == 0
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV


def test_no_valid_pixels_and_errors(hip_lib, oracle_mod, small_pair):
    (rgbA, dA), (rgbB, dB), T = small_pair
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, np.zeros_like(dB))          # every source depth invalid
    assert reg.alignFrames360(np.eye(4), 2) == 2         # RGBD360_NO_VALID_PIXELS
    assert np.allclose(reg.getOptimalPose(), np.eye(4))
    from rgbd360_amd.register import Rgbd360Error
    with pytest.raises(Rgbd360Error):
        reg.alignFrames360(np.eye(4), 2, occlusion=3)   # only 0, 1, 2 exist (RPI.h:4517)
    assert reg.alignFrames360(np.eye(4), 2, occlusion=1) == 2
    fresh = _mk(hip_lib, 3)
    fresh.setTargetFrame(rgbA, dA)
    with pytest.raises(Rgbd360Error):                    # source frame missing
        fresh.alignFrames360(np.eye(4), 0)


def test_identity_pair_stays_at_identity(hip_lib, oracle_mod, small_pair):
    (rgbA, dA), _, _ = small_pair
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbA, dA)
    rc = reg.alignFrames360(np.eye(4), 2)
    assert rc == 0
    rot, trans = synth.pose_error(reg.getOptimalPose(), np.eye(4))
    assert rot < 1e-5 and trans < 1e-5


def test_promote_source_to_target(hip_lib, oracle_mod, small_pair):
    (rgbA, dA), (rgbB, dB), T = small_pair
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    reg.promoteSourceToTarget()                          # B becomes the target
    reg.setSourceFrame(rgbA, dA)
    ref = _mk(hip_lib, 3)
    ref.setTargetFrame(rgbB, dB)
    ref.setSourceFrame(rgbA, dA)
    for name in ("gray_trg", "depth_trg", "gx", "dgy"):
        assert np.array_equal(reg.plane(name, 1), ref.plane(name, 1))
    assert reg.alignFrames360(np.eye(4), 2) == ref.alignFrames360(np.eye(4), 2) == 0
    assert np.array_equal(reg.getOptimalPose(), ref.getOptimalPose())


@pytest.mark.parametrize("method", [0, 1, 2])
def test_full_size_2048x1024(hip_lib, oracle_mod, method):
    """BASELINE.json configs 2/3 at full size (and DEPTH_CONSISTENCY, the third of the pass's template space): GPU vs oracle pose, and
    both vs ground truth."""
    pair = synth.make_pair(2048, 1024, seed=1234)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=4)
    rc = reg.alignFrames360(np.eye(4), method)
    st, pose_ref = ora.align360(np.eye(4), method)
    assert rc == st == 0
    assert reg.num_iterations == list(ora.result.iters)[:4]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    rot_gt, trans_gt = synth.pose_error(reg.getOptimalPose(), T)
    # (depth alone, on 1 mm-quantised ranges, recovers the translation to 2.4 mm -- GPU and oracle alike; with the photometric term 0.6 mm)
    assert rot_gt < 5e-4 and trans_gt < (4e-3 if method == 1 else 2e-3), (rot_gt, trans_gt)
    _assert_libm_oracle_agrees(reg, ora, method, 4)
    # size-independent property: re-running is bitwise reproducible (fixed-order reductions, no atomics)
    pose1 = reg.getOptimalPose()
    reg.alignFrames360(np.eye(4), method)
    assert np.array_equal(pose1, reg.getOptimalPose())
    # counts at full size are exact against the oracle
    e = reg.eval(0, pose_ref, method)
    _, err2, nvalid = ora.error(0, pose_ref, method)
    assert e["n_valid"] == nvalid
    assert abs(e["err2"] - err2) <= ERR2_RTOL * err2


def test_full_size_4096x2048_with_planes(hip_lib, oracle_mod):
    """BASELINE.json configs[4] at full size: 4096x2048 photo+depth alignment (5 levels) against the oracle -- identical
    iteration counts, exact pixel counts, pose within the device tolerance -- plus the Frame360 stages on the same frame:
    cloud and region labels bit-exact, normal map equal to the oracle's."""
    pair = synth.make_pair(4096, 2048, seed=1234)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=5)
    rc = reg.alignFrames360(np.eye(4), 2)
    st, pose_ref = ora.align360(np.eye(4), 2)
    assert rc == st == 0
    assert reg.num_iterations == list(ora.result.iters)[:5]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    rot_gt, trans_gt = synth.pose_error(reg.getOptimalPose(), T)
    assert rot_gt < 2e-4 and trans_gt < 1e-3, (rot_gt, trans_gt)
    _assert_libm_oracle_agrees(reg, ora, 2, 5)
    e = reg.eval(0, pose_ref, 2)
    _, err2, nvalid = ora.error(0, pose_ref, 2)
    H, g, Hd, gd, nvis = ora.hessgrad(0, pose_ref, 2)
    assert e["n_valid"] == nvalid and e["n_visible"] == nvis
    assert abs(e["err2"] - err2) <= ERR2_RTOL * err2
    assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
    # Frame360 stages on the target frame
    from rgbd360_amd.register import Frame360Stages
    (rgbA, dA) = pair[0]
    st360 = Frame360Stages(reg)
    out = st360.frame_planes(dA, convention=2, angular_threshold=0.03)
    xyz = oracle_mod.sphere_cloud(dA, 2)
    assert np.array_equal(np.nan_to_num(xyz), np.nan_to_num(out["xyz"]))
    nrm, _ = oracle_mod.f360_normals(xyz, 2048, 4096, 0.05, 8.0, 1)
    ok = ~np.isnan(nrm[:, 0])
    assert np.array_equal(np.isnan(out["normals"][:, 0]), ~ok)
    assert np.abs(out["normals"][ok] - nrm[ok]).max() <= 1.2e-7 and (out["normals"][ok] == nrm[ok]).mean() > 0.9999
    labels, planes = oracle_mod.f360_plane_segment(xyz, out["normals"], 2048, 4096, 40, 0.03, 0.05, 0.001, 1)
    assert np.array_equal(np.asarray(out["labels"]).reshape(-1), np.asarray(labels).reshape(-1))
    assert len(out["planes"]) == len(planes) >= 6
    # hull stage at full size: the 64-direction polygon of the larger regions against the exact hull of their inliers (Qhull)
    checked = 0
    for p_d in sorted(out["planes"], key=lambda q: -q["count"])[:6]:
        exact, center, _nv = oracle_mod.f360_hull_stats(xyz, labels, p_d)
        assert 0.997 * exact <= p_d["area"] <= exact * (1 + 1e-5), (p_d["area"], exact)
        assert np.abs(p_d["center_hull"] - center).max() < 5e-3 and p_d["hull_points"] >= 3
        _check_hull_polygon(p_d)
        checked += 1
    assert checked == 6


def test_hull_areas_read_the_room_at_2048x1024(hip_lib, oracle_mod):
    """Frame360.h:1025-1041 at the size the metric is quoted on: the planes of the synthetic 3 x 6 x 8 m room (segmentation thresholds
    scaled with the pixel pitch, as tools/pbmap_perf.py does: PCL's comparator links window-averaged normals) carry HULL areas --
    floor / ceiling 48 m2, side walls 24 / 18 m2, minus a margin of a few 3-mrad pixels -- where the density-weighted moment rectangle
    of rounds 1-2 read the floor as ~16 m2; min_area_plane (0.12 m2) and the area ranking of RegisterRGBD360.h:126-136 now see
    metric areas.  Each area is also held against the exact hull of the region's inliers (Qhull, oracle/)."""
    from rgbd360_amd.register import Frame360Stages
    W, H = 2048, 1024
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
    out = Frame360Stages(_mk(hip_lib, 3)).frame_planes(dA, convention=2, angular_threshold=0.015, min_inliers=640)
    xyz = oracle_mod.sphere_cloud(dA, 2)
    true_area = {0: 48.0, 1: 24.0, 2: 18.0}      # by the axis of the wall's normal: floor / ceiling, y walls (8 x 3), z walls (6 x 3)
    walls = {}
    for p in out["planes"]:
        ax = int(np.argmax(np.abs(p["normal"])))
        if abs(p["normal"][ax]) < 0.9999 or p["count"] < 50000:
            continue
        exact, center, _nv = oracle_mod.f360_hull_stats(xyz, out["labels"], p)
        # (an inscribed polygon: never larger; what it can lose is a long, slightly bowed edge -- the margin of a wall's region widens
        #  with the range -- whose normals all fall between two directions: 0.7 % on the 8 m floor with 64 directions, hence 256)
        assert 0.997 * exact <= p["area"] <= exact * (1 + 1e-5), (p["area"], exact)
        assert np.abs(p["center_hull"] - center).max() < 5e-3
        if p["area"] < 5.0:        # (the cap of a few thousand tiny pixels around a pole of the sphere is a region of its own)
            continue
        assert 0.90 * true_area[ax] < p["area"] <= true_area[ax] * 1.001, (ax, p["area"], p["area_moment"])      # (a wall loses ~0.1 m of margin per side)
        assert p["area_moment"] < (0.5 if ax == 0 else 0.9) * p["area"]      # what the moment rectangle would have reported
        walls.setdefault(ax, []).append(p["area"])
    # (the wall behind the camera straddles the panorama's first / last column: two regions, neither an axis-aligned 18 m2 one)
    assert len(walls.get(0, [])) == 2 and len(walls.get(1, [])) == 2 and len(walls.get(2, [])) >= 1, walls
    assert all(abs(a - 48.0) <= 2.0 for a in walls[0]), walls[0]          # the floor and the ceiling read 48 +- 2 m2


def _check_hull_polygon(p, area_rtol=2e-3):
    """The polygon a plane record carries (rgbd360_plane::hull, the role of mrpt::pbmap::Plane::polygonContourPtr): at most 64 vertices on the
    fitted plane, convex, counter-clockwise seen from the side the normal points to, its area the record's (to the thinning of a hull of
    more than 64 vertices) and its mass centre the record's center_hull."""
    hv, n, c = p["hull"].astype(np.float64), p["normal"].astype(np.float64), p["centroid"].astype(np.float64)
    assert 3 <= len(hv) <= 64 and len(hv) <= p["hull_points"]
    assert np.abs((hv - c) @ n).max() < 1e-4 * max(1.0, np.abs(hv).max())
    m = len(hv)
    turns = [n @ np.cross(hv[(i + 1) % m] - hv[i], hv[(i + 2) % m] - hv[(i + 1) % m]) for i in range(m)]
    assert min(turns) > -1e-9, min(turns)
    area = 0.5 * abs(sum(n @ np.cross(hv[i] - hv[0], hv[(i + 1) % m] - hv[0]) for i in range(m)))
    assert area <= p["area"] * (1 + 1e-5) and area >= p["area"] * (1 - (area_rtol if p["hull_points"] > 64 else 1e-5)), (area, p["area"], p["hull_points"])


@pytest.mark.gpu
def test_sensor_cloud_from_undistorted_float_depth(hip_lib, tmp_path):
    """The sensor cloud from the float32-metres image Frame360::undistort leaves (rgbd360_sensor_cloud_ex, depth_type 1; getPointCloudUndist,
    CloudRGBD_Ext.h:78-131): with the millimetre image times 0.001 as that image the cloud is the uint16 path's, bit for bit; corrected by
    an intrinsic model with one multiplier everywhere every finite point is scaled by it (x, y and z are all proportional to z)."""
    import struct
    from rgbd360_amd.register import DepthModel, Frame360Stages
    (rgb, dep), _, _, K = synth.make_pinhole_pair(320, 240, seed=9)
    st = Frame360Stages(_mk(hip_lib, 2))
    want = st.sensor_cloud(dep, 2)
    as_m = (0.001 * dep.astype(np.float64)).astype(np.float32)
    got = st.sensor_cloud(as_m, 2)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(want)) and np.isfinite(want).mean() > 0.5
    path = tmp_path / "model"
    with open(path, "wb") as f:            # 640 x 480 model of 8 x 6-pixel bins, five 2 m slices, multiplier 1.03 everywhere
        f.write(b"DiscreteDepthDistortionModel v01\n" + struct.pack("<iiii", 640, 480, 8, 6) + struct.pack("<d", 2.0) + struct.pack("<ii", 80, 80))
        one = struct.pack("<d", 10.0) + struct.pack("<i", 5) + struct.pack("<d", 2.0)
        for vec in (np.full(5, 100.0), np.ones(5), np.ones(5), np.full(5, 1.03)):
            one += struct.pack("<iii", 4, 5, 1) + vec.astype(np.float32).tobytes()
        f.write(one * 6400)
    corrected = DepthModel(path, 2).undistort(as_m)
    assert np.allclose(corrected[as_m > 0], as_m[as_m > 0] * np.float32(1.03), rtol=2e-7)
    scaled = st.sensor_cloud(corrected, 2)
    both = np.isfinite(scaled).all(-1) & np.isfinite(want).all(-1)
    assert both.mean() > 0.5 and np.allclose(scaled[both], want[both] * 1.03, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hull_polygon_of_a_disc_is_thinned_to_64_vertices(hip_lib):
    """A disc on a fronto-parallel plane has an extreme pixel in every one of the hull stage's directions: far more than the 64 vertices a
    plane record carries, so the record's polygon is the hull's extreme vertex in 64 evenly spaced directions (found by ONE walk round the
    hull: the extreme vertex only moves forward as the direction turns).  Convex, counter-clockwise, inside the hull, and its area within
    the inscribed 64-gon's share of the disc (cos-free bound: 1 - (pi / 64)^2 * 2 / 3 > 0.998)."""
    from rgbd360_amd.register import Frame360Stages
    H = W = 256
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    inside = (jj - 127.5) ** 2 + (ii - 127.5) ** 2 <= 100.0 ** 2
    z = np.where(inside, 2.0, np.nan).astype(np.float32)
    x = ((jj - W / 2) * 0.004 * 2.0).astype(np.float32)
    y = ((ii - H / 2) * 0.004 * 2.0).astype(np.float32)
    xyz = np.stack([np.where(inside, x, np.nan), np.where(inside, y, np.nan), z], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.array([0.0, 0.0, -1.0], np.float32), (H * W, 1))
    nrm[~inside.reshape(-1)] = np.nan
    st = Frame360Stages(_mk(hip_lib, 2))
    labels, planes = st.plane_fit(xyz, nrm, H, W, 40, 0.05, 0.05, 0.01, 0)
    assert len(planes) == 1
    p = planes[0]
    assert p["hull_points"] > 64 and len(p["hull"]) > 48, (p["hull_points"], len(p["hull"]))
    _check_hull_polygon(p)
    r = 100.0 * 0.008
    assert 0.99 * np.pi * r * r < p["area"] < 1.001 * np.pi * r * r
    ang = np.unwrap(np.arctan2(p["hull"][:, 1] - p["center_hull"][1], p["hull"][:, 0] - p["center_hull"][0]))
    steps = np.abs(np.diff(np.concatenate([ang, [ang[0] + np.sign(ang[-1] - ang[0]) * 2 * np.pi]])))
    assert steps.max() < 2.5 * 2 * np.pi / 64, steps.max()          # no gap of several directions: every direction found its vertex


@pytest.mark.gpu
def test_colour_stage_on_a_striped_image(hip_lib, oracle_mod):
    """The colour stage where a block of 8192 pixels holds 64 small regions, every pixel of which is a sample of the dominant-colour
    search: more regions than the block's 16-entry table and more samples than its 4096-entry list hold, so sums, hue bins and samples
    also take their direct routes to the global table / the regions' counters.  Descriptors and dominant colours exact against the
    numpy restatement, as on a frame of a few large regions."""
    from rgbd360_amd.register import Frame360Stages
    H, W = 128, 256
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    z = np.where((jj // 4) % 2 == 0, 2.0, 2.6).astype(np.float32)
    x = ((jj - W / 2) * 0.004 * z).astype(np.float32)
    y = ((ii - H / 2) * 0.004 * z).astype(np.float32)
    xyz = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.array([0.0, 0.0, -1.0], np.float32), (H * W, 1))
    rng = np.random.default_rng(12)
    base = rng.integers(30, 226, size=(W // 4, 3))
    rgb = np.clip(base[jj // 4] + rng.integers(-25, 26, size=(H, W, 3)), 0, 255).astype(np.uint8)
    rgb[::7, ::5] = 0                                  # black pixels: no normalised colour, the dark hue bin
    st = Frame360Stages(_mk(hip_lib, 2))
    st.set_color_image(rgb)
    labels, planes = st.plane_fit(xyz, nrm, H, W, 40, 0.05, 0.05, 0.01, 0)
    assert len(planes) == W // 4
    _, want = oracle_mod.f360_plane_colour(labels.reshape(H, W), rgb, planes)
    modes = oracle_mod.f360_plane_colour_mode(labels.reshape(H, W), rgb, planes)
    for p, w, m in zip(planes, want, modes):
        assert p["color_count"] == w["color_count"] > 0
        assert np.abs(p["color_nrgb"] - w["color_nrgb"]).max() <= 1e-7 and np.abs(p["color_dev"] - w["color_dev"]).max() <= 2e-6
        assert np.abs(p["hist_h"] - w["hist_h"]).max() <= 1e-7
        assert p["color_mode_count"] == m["color_mode_count"] == p["color_count"]          # 512-pixel regions: every coloured pixel is a sample
        assert np.array_equal(p["color_mode"], m["color_mode"]) and p["intensity_mode"] == m["intensity_mode"] and p["color_concentration"] == m["color_concentration"]


@pytest.mark.gpu
def test_hull_stage_on_a_striped_image(hip_lib, oracle_mod):
    """The hull stage where EVERY other pixel is a boundary pixel: a fronto-parallel staircase, 4-pixel-wide stripes alternating between
    two depths (the plane comparator's distance test separates them), so a block's share of boundary pixels is several times the
    640 entries its LDS list holds and the kernel walks it in rounds.  Every stripe's hull area and mass centre against the exact hull."""
    from rgbd360_amd.register import Frame360Stages
    H, W = 128, 256
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    z = np.where((jj // 4) % 2 == 0, 2.0, 2.6).astype(np.float32)
    x = ((jj - W / 2) * 0.004 * z).astype(np.float32)
    y = ((ii - H / 2) * 0.004 * z).astype(np.float32)
    xyz = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.array([0.0, 0.0, -1.0], np.float32), (H * W, 1))
    st = Frame360Stages(_mk(hip_lib, 2))
    labels, planes = st.plane_fit(xyz, nrm, H, W, 40, 0.05, 0.05, 0.01, 0)
    labels_ref, planes_ref = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 40, 0.05, 0.05, 0.01, 0)
    assert np.array_equal(labels, labels_ref) and len(planes) == len(planes_ref) == W // 4
    boundary = (labels[:, 1:] != labels[:, :-1]).sum() * 2 + 2 * H + 2 * (W - 2)
    assert boundary > 4 * 640 * ((H * W + 8191) // 8192)          # several rounds of the list in every block
    for p in planes:
        exact, center, _nv = oracle_mod.f360_hull_stats(xyz, labels, p)
        assert p["hull_points"] >= 4 and abs(p["area"] - exact) <= 1e-4 * exact, (p["area"], exact)
        assert np.abs(p["center_hull"] - center).max() < 1e-4
        _check_hull_polygon(p)


def _pingpong(n_pairs, n_unique):
    idx, k, step = [], 0, 1
    for _ in range(n_pairs + 1):
        idx.append(k)
        if k + step < 0 or k + step >= n_unique:
            step = -step
        k += step
    return idx


def _hip_runtime():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return p.value

    def free(p):
        hip.hipFree(C.c_void_p(p))
    return to_device, free


def test_full_size_sequence_config3(hip_lib, oracle_mod):
    """BASELINE.json configs[3] at its stated size: an odometry walk of 33 frames of 2048x1024 (32 consecutive pairs, ping-pong
    over 5 rendered frames like bench.py) through the lock-step sequence engine -- frames resident in HBM with n_inflight = 32
    (two engines x 16 slots, the 512-thread batch pass, the fused frame set-up: the configuration the sequence throughput is quoted
    on) and 8 (four pairs per slot: frame reuse inside a span), and host frames with the default slot count.  Poses, status and
    iteration counts are BIT-IDENTICAL to pair-by-pair alignFrames360 (OdometryRGBD360.cpp:141-297 calls RPI.h:4519-4784 once per
    pair); three of the pairs are also held against the reference-faithful oracle (libm asinf / atan2f / roundf, float32
    accumulators): same iterations per level, pose within the north-star tolerance.  Then the occlusion-aware sequence (per-context
    route) on one full-size pair, the same two checks."""
    W, H, n_unique, n_pairs = 2048, 1024, 5, 32
    uniq = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(n_unique)]
    order = _pingpong(n_pairs, n_unique)
    assert len(order) == 33
    # pair by pair, every distinct (target, source) once
    pairwise = {}
    one = _mk(hip_lib, 4)
    for a, b in sorted(set(zip(order[:-1], order[1:]))):
        one.setTargetFrame(*uniq[a])
        one.setSourceFrame(*uniq[b])
        rc = one.alignFrames360(np.eye(4), 2)
        pairwise[(a, b)] = (one.getOptimalPose().copy(), rc, list(one.num_iterations))
        assert rc == 0
    to_device, free = _hip_runtime()
    rgb_d = [to_device(f[0]) for f in uniq]
    dep_d = [to_device(f[1]) for f in uniq]
    reg = _mk(hip_lib, 4)
    try:
        runs = [("resident/32", reg.alignSequenceDev([rgb_d[k] for k in order], [dep_d[k] for k in order], H, W, 0, method=2, n_inflight=32)),
                ("resident/8", reg.alignSequenceDev([rgb_d[k] for k in order], [dep_d[k] for k in order], H, W, 0, method=2, n_inflight=8))]
    finally:
        for q in rgb_d + dep_d:
            free(q)
    runs.append(("host frames", reg.alignSequence([uniq[k] for k in order], method=2)))
    for name, (poses, status, iters) in runs:
        assert poses.shape == (n_pairs, 4, 4)
        for j in range(n_pairs):
            p, rc, it = pairwise[(order[j], order[j + 1])]
            assert status[j] == rc and list(iters[j])[:4] == it, (name, j, status[j], list(iters[j]), it)
            assert np.array_equal(poses[j], p), (name, j, np.abs(poses[j] - p).max())
    # the reference-faithful oracle on three of the pairs (forward, backward, another forward)
    poses = runs[0][1][0]
    for j in (0, 5, 2):
        a, b = order[j], order[j + 1]
        ora = oracle_mod.Oracle(n_pyr=4, math_mode=0, reduce_mode=0)
        ora.set_target(*uniq[a])
        ora.set_source(*uniq[b])
        st, pose_libm = ora.align360(np.eye(4), 2)
        assert st == 0 and pairwise[(a, b)][2] == list(ora.result.iters)[:4], (j, pairwise[(a, b)][2], list(ora.result.iters)[:4])
        rot, trans = synth.pose_error(poses[j], pose_libm)
        assert rot <= ROT_TOL and trans <= TRANS_TOL, (j, rot, trans)
        T_gt = np.linalg.inv(synth.trajectory_pose(a, 7)) @ synth.trajectory_pose(b, 7)
        rot, trans = synth.pose_error(poses[j], T_gt)
        assert rot < 5e-4 and trans < 2e-3, (j, rot, trans)
    # size-independent property of the walk: a pair and its reverse are inverse motions
    fwd, bwd = pairwise[(0, 1)][0].astype(np.float64), pairwise[(1, 0)][0].astype(np.float64)
    rot, trans = synth.pose_error(fwd @ bwd, np.eye(4))
    assert rot < 2e-4 and trans < 1e-3, (rot, trans)
    # occlusion-aware sequence at full size, one pair (+ its neighbour so that a frame is reused)
    po, so, io = reg.alignSequence([uniq[0], uniq[1], uniq[2]], method=2, occlusion=2)
    for j in range(2):
        one.setTargetFrame(*uniq[j])
        one.setSourceFrame(*uniq[j + 1])
        rc = one.alignFrames360(np.eye(4), 2, 2)
        assert rc == so[j] == 0 and list(io[j])[:4] == list(one.num_iterations)
        assert np.array_equal(po[j], one.getOptimalPose()), j
    ora = oracle_mod.Oracle(n_pyr=4, math_mode=0, reduce_mode=0)
    ora.set_target(*uniq[0])
    ora.set_source(*uniq[1])
    st, pose_libm = ora.align360(np.eye(4), 2, 2)
    assert st == 0 and list(io[0])[:4] == list(ora.result.iters)[:4], (list(io[0]), list(ora.result.iters)[:4])
    rot, trans = synth.pose_error(po[0], pose_libm)
    assert rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)


def test_sequence_batch_matches_pairwise_alignment(hip_lib, oracle_mod):
    """BASELINE.json config 4 in miniature: a chunk of an odometry sequence aligned with frame reuse
    (promoteSourceToTarget) gives the same poses as aligning every pair from scratch, and as the oracle."""
    from rgbd360_amd.batch import align_sequence, compose_trajectory
    frames = {k: synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(4)}
    reg = _mk(hip_lib, 3)
    poses, status, iters = align_sequence(reg, lambda k: frames[k], 0, 3, 2)
    assert (status == 0).all()
    for j in range(3):
        fresh = _mk(hip_lib, 3)
        fresh.setTargetFrame(*frames[j])
        fresh.setSourceFrame(*frames[j + 1])
        assert fresh.alignFrames360(np.eye(4), 2) == 0
        assert np.array_equal(fresh.getOptimalPose(), poses[j])
        ora = oracle_mod.Oracle(n_pyr=3, math_mode=1, reduce_mode=1)
        ora.set_target(*frames[j])
        ora.set_source(*frames[j + 1])
        st, pose_ref = ora.align360(np.eye(4), 2)
        rot, trans = synth.pose_error(poses[j], pose_ref)
        assert st == 0 and rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV
        T_gt = np.linalg.inv(synth.trajectory_pose(j, 7)) @ synth.trajectory_pose(j + 1, 7)
        rot, trans = synth.pose_error(poses[j], T_gt)
        assert rot < 5e-3 and trans < 1e-2, (j, rot, trans)       # 256x128 resolution limit
    assert compose_trajectory(poses).shape == (4, 4, 4)


@pytest.mark.parametrize("convention", [0, 1, 2])
def test_sphere_cloud_bit_exact(hip_lib, oracle_mod, convention):
    """Frame360::buildSphereCloud_fromImage / Frame360_stereo::buildSphereCloud / LUT convention."""
    (rgbA, dA), _, _ = synth.make_pair(256, 128, seed=11, depth_f32=(convention == 1))
    d = dA.copy()
    d[10:20, 30:50] = 0                       # a hole: NaN points
    if convention == 1:
        d[40:44, 5:9] = 20.0                  # beyond the 15 m cut of the stereo variant
    reg = _mk(hip_lib, 3)
    got = reg.sphere_cloud(d, convention)
    ref = oracle_mod.sphere_cloud(d, convention)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.isnan(ref).any() and not np.isnan(ref).all()
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32))


# ---- Frame360 stages: rows a14 (normal map) and a15 (planar regions + inlier moments) ---------------------------------
def _room_cloud(oracle_mod, W=512, H=256, seed=5):
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=seed)
    d = dA.copy()
    d[60:90, 200:230] = 0                      # a hole (NaN points, depth discontinuity all around)
    d[150:170, 300:340] = (d[150:170, 300:340] * 0.7).astype(np.uint16)     # a box in front of the wall: depth steps
    return oracle_mod.sphere_cloud(d, 2), d, H, W


@pytest.mark.parametrize("depth_mode", [0, 1])
def test_normal_map_matches_oracle(hip_lib, oracle_mod, depth_mode):
    from rgbd360_amd.register import Frame360Stages
    xyz, d, H, W = _room_cloud(oracle_mod)
    st = Frame360Stages(_mk(hip_lib, 3))
    dist_ref = oracle_mod.f360_distance_map(xyz, H, W, 0.05, depth_mode)
    dist = st.distance_map(xyz, H, W, 0.05, depth_mode)
    near = dist_ref < 10.0                      # the device map is exact below its truncation radius
    assert np.array_equal(dist_ref == 0, dist == 0)
    assert np.abs(dist[near] - dist_ref[near]).max() < 1e-5
    assert (dist[~near] >= 10.0 - 1e-5).all()
    nrm = st.normals(xyz, H, W, 0.05, 8.0, depth_mode)
    ref, win = oracle_mod.f360_normals(xyz, H, W, 0.05, 8.0, depth_mode)
    assert np.array_equal(np.isnan(nrm[:, 0]), np.isnan(ref[:, 0]))
    ok = ~np.isnan(ref[:, 0])
    assert ok.mean() > 0.3
    # per-tile integral images in float64 vs the oracle's float64 window sums: identical up to a rare 1-ulp rounding flip
    assert np.abs(nrm[ok] - ref[ok]).max() <= 1.2e-7
    assert (nrm[ok] == ref[ok]).mean() > 0.9999
    assert np.allclose(np.linalg.norm(nrm[ok], axis=1), 1.0, atol=1e-5)
    assert ((nrm[ok] * xyz[ok]).sum(1) <= 1e-4).all()  # flipped towards the viewpoint (origin)


@pytest.mark.parametrize("H,W", [(37, 250), (70, 65), (33, 129), (9, 3), (64, 64), (41, 700)])
@pytest.mark.parametrize("depth_mode", [0, 1])
def test_distance_map_ragged_sizes(hip_lib, oracle_mod, H, W, depth_mode):
    """The bit-mask distance map on widths that are not a multiple of the 64-pixel mask word / the 256 x 8 and 64 x 32 tiles:
    blocky depth steps, NaN holes, changes right at the image border."""
    from rgbd360_amd.register import Frame360Stages
    rng = np.random.default_rng(H * 1000 + W + depth_mode)
    z = np.full((H, W), 2.0, np.float32)
    for _ in range(6):
        r, c = rng.integers(0, H), rng.integers(0, W)
        z[r:r + rng.integers(1, 12), c:c + rng.integers(1, 40)] += np.float32(rng.uniform(0.5, 2.0))
    z[:, W - 1] += 1.5 * (rng.random(H) < 0.3)            # steps in the last column / row
    z[H - 1, :] += 1.5 * (rng.random(W) < 0.3)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    xyz = np.stack([(xx - W / 2) * z / 200, (yy - H / 2) * z / 200, z], -1).reshape(-1, 3).astype(np.float32)
    holes = rng.random(H * W) < 0.002
    xyz[holes] = np.nan
    st = Frame360Stages(_mk(hip_lib, 2))
    dist_ref = oracle_mod.f360_distance_map(xyz, H, W, 0.05, depth_mode)
    dist = st.distance_map(xyz, H, W, 0.05, depth_mode)
    assert np.array_equal(dist_ref == 0, dist == 0) and (dist_ref == 0).any()
    # PCL's two passes never update the first / last column, the first row (forward) and the last row (backward): those
    # pixels keep rows + cols unless they are depth changes themselves (they never get a normal); inside, both are the chamfer metric
    inner = np.zeros((H, W), bool)
    inner[1:H - 1, 1:W - 1] = True
    dist, dist_ref = np.asarray(dist).reshape(H, W), np.asarray(dist_ref).reshape(H, W)
    near = (dist_ref < 10.0) & inner
    assert near.any()
    assert np.abs(dist[near] - dist_ref[near]).max() < 1e-5      # the oracle accumulates 1 / 1.4 steps, the device multiplies
    assert (dist[inner & ~near] >= 10.0 - 1e-5).all()


@pytest.mark.parametrize("depth_mode,ang", [(1, 0.05), (1, 0.03), (0, 0.0398)])
def test_plane_regions_match_oracle(hip_lib, oracle_mod, depth_mode, ang):
    """Same normals in, same partition out (labels are the regions' smallest pixel index on both sides), moments and
    plane parameters equal to float32 rounding."""
    from rgbd360_amd.register import Frame360Stages
    xyz, d, H, W = _room_cloud(oracle_mod)
    nrm, _ = oracle_mod.f360_normals(xyz, H, W, 0.05, 8.0, depth_mode)
    st = Frame360Stages(_mk(hip_lib, 3))
    labels, planes = st.plane_fit(xyz, nrm, H, W, 40, ang, 0.05, 0.001, depth_mode)
    labels_ref, planes_ref = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 40, ang, 0.05, 0.001, depth_mode)
    assert np.array_equal(labels, labels_ref)                       # integer / index work: exact
    assert [p["root"] for p in planes] == [p["root"] for p in planes_ref]
    assert [p["count"] for p in planes] == [p["count"] for p in planes_ref]
    for a, b in zip(planes, planes_ref):
        assert np.allclose(a["centroid"], b["centroid"], atol=1e-5)
        assert abs(a["curvature"] - b["curvature"]) < 1e-6
        if b["curvature"] > 1e-9:        # (collinear one-column regions have a degenerate normal direction)
            assert abs(abs(np.dot(a["normal"], b["normal"])) - 1) < 1e-5 and abs(a["d"] - b["d"]) < 1e-4


def test_small_regions_curvature_against_float64(hip_lib):
    """The inlier sums are integers of terms rounded to 2^-28 m^2: on patches of 12-24 pixels 2.5 m away with 1 mm of noise (smallest
    eigenvalue ~1e-6 m^2 under sums of ~6 m^2 per pixel) the curvature is within 0.2 % of numpy's float64 value -- with the 2^-24 of
    rounds 2-5 it was off by up to several per cent, and tests/tools/planes_soak.py met planes on the other side of max_curvature."""
    from rgbd360_amd.register import Frame360Stages
    H, W = 48, 96
    rng = np.random.default_rng(4)
    xyz = np.full((H, W, 3), np.nan, np.float32)
    nrm = np.full((H, W, 3), np.nan, np.float32)
    want = {}
    for k in range(18):
        r0, c0 = 3 + 7 * (k // 6), 3 + 15 * (k % 6)
        hh, ww = int(rng.integers(3, 5)), int(rng.integers(4, 7))
        rr, cc = np.meshgrid(np.arange(hh), np.arange(ww), indexing="ij")
        pts = np.stack([0.3 * k + 0.004 * cc, -0.5 + 0.004 * rr, 2.5 + rng.normal(0, 1e-3, size=rr.shape)], axis=2).astype(np.float32)
        xyz[r0:r0 + hh, c0:c0 + ww] = pts
        nrm[r0:r0 + hh, c0:c0 + ww] = (0, 0, -1)
        ev = np.linalg.eigvalsh(np.cov(pts.reshape(-1, 3).astype(np.float64).T, bias=True))
        want[r0 * W + c0] = (hh * ww, abs(ev[0]) / ev.sum())
    st = Frame360Stages(_mk(hip_lib, 2))
    labels, planes = st.plane_fit(xyz, nrm, H, W, 10, 0.05, 0.05, 0.9, 0)
    assert sorted(p["root"] for p in planes) == sorted(want)
    for p in planes:
        n, curv = want[p["root"]]
        assert p["count"] == n
        assert abs(p["curvature"] - curv) < 2e-3 * curv, (p["root"], p["curvature"], curv)


def test_plane_fit_refuses_sums_beyond_their_range(hip_lib):
    """64-bit sums of 2^-28 m^2 terms hold N r^2 < 3.4e10 m^2 (a whole 4096 x 2048 frame as ONE region 64 m away); beyond, a sum wraps.
    One region of 32768 points 1100 m away (3.96e10): error -8, not a plane with a nonsense covariance."""
    from rgbd360_amd.register import Frame360Stages, Rgbd360Error
    H, W = 128, 256
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    xyz = np.stack([0.5 * (cc - W / 2), 0.5 * (rr - H / 2), np.full(rr.shape, 1100.0)], axis=2).astype(np.float32)
    nrm = np.zeros((H, W, 3), np.float32)
    nrm[..., 2] = -1
    st = Frame360Stages(_mk(hip_lib, 2))
    with pytest.raises(Rgbd360Error, match=r"\(-8\).*out of range"):
        st.plane_fit(xyz, nrm, H, W, 40, 0.05, 0.05, 0.9, 0)
    xyz[..., 2] = 900.0                                               # 2.65e10: inside
    labels, planes = st.plane_fit(xyz, nrm, H, W, 40, 0.05, 0.05, 0.9, 0)
    assert len(planes) == 1 and planes[0]["count"] == H * W and abs(planes[0]["centroid"][2] - 900.0) < 1e-3


@pytest.mark.parametrize("W,seed,trans,rot", [(512, 943, 0.1, 3.0), (512, 840, 0.3, 10.0), (256, 471, 0.1, 25.0)])
def test_plane_regions_when_a_loose_threshold_links_the_whole_room(hip_lib, oracle_mod, W, seed, trans, rot):
    """An angular threshold of 0.1 rad on a small frame: the smoothed normals turn the room's corners by less than that per pixel, walls,
    floor and ceiling become ONE region of ~90 % of the frame (a component every merge level takes part in), which the curvature test
    refuses; what is left are slivers just above min_inliers.  Labels, planes and their order equal the CPU checker's (the draws are those
    tests/tools/hull_soak.py met with seed 101: trials 0, 6, 3)."""
    from rgbd360_amd.register import Frame360Stages
    H = W // 2
    depth = synth.make_pair(W, H, seed=seed, trans=trans, rot_deg=rot)[1][1]
    st = Frame360Stages(_mk(hip_lib, 2))
    out = st.frame_planes(depth, convention=2, angular_threshold=0.1, min_inliers=40)
    xyz = oracle_mod.sphere_cloud(depth, 2)
    assert np.array_equal(np.asarray(out["xyz"]).reshape(-1, 3), np.asarray(xyz).reshape(-1, 3))
    nrm, _ = oracle_mod.f360_normals(xyz, H, W, 0.05, 8.0, 1)
    nrm, nrm_dev = np.asarray(nrm).reshape(-1, 3), np.asarray(out["normals"]).reshape(-1, 3)
    ok = ~np.isnan(nrm[:, 0])
    assert np.array_equal(np.isnan(nrm_dev[:, 0]), ~ok) and np.abs(nrm_dev[ok] - nrm[ok]).max() <= 1.2e-7
    # (the segmentation is checked on the device's own normal map: a last-bit difference of a normal may move a comparison)
    labels_ref, planes_ref = oracle_mod.f360_plane_segment(xyz, nrm_dev, H, W, 40, 0.1, 0.05, 0.001, 1)
    labels_ref = np.asarray(labels_ref).reshape(-1)
    assert np.array_equal(np.asarray(out["labels"]).reshape(-1), labels_ref)
    counts = np.bincount(labels_ref[labels_ref >= 0])
    assert counts.max() > 0.75 * labels_ref.size                     # the scenario: one region holds the room
    assert [p["root"] for p in out["planes"]] == [p["root"] for p in planes_ref]
    assert [p["count"] for p in out["planes"]] == [p["count"] for p in planes_ref]
    assert all(p["count"] < 64 for p in planes_ref)                  # ... and no plane comes out of it


@pytest.mark.parametrize("W,H", [(250, 101), (700, 37), (65, 70), (24, 15), (513, 129), (1028, 37), (1500, 70), (1920, 33), (1026, 20), (4100, 18)])
def test_plane_regions_ragged_sizes(hip_lib, oracle_mod, W, H):
    """The tiled / hierarchical component passes on sizes that are not multiples of their tiles (256 x 4 link tile, 4 rows per
    run block, 64 x 64 merge tiles, 512-pixel wave strips; from 1024 columns on -- rows of whole dwords -- four waves share a row of the
    run pass, 1026 columns take the one-wave kernel again): labels, counts and plane roots equal the oracle's."""
    from rgbd360_amd.register import Frame360Stages
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=W + H)
    d = dA.copy()
    d[H // 3:H // 3 + 3, W // 4:W // 4 + 9] = 0                        # a hole
    d[H // 2:, W // 2:W // 2 + W // 8] = (d[H // 2:, W // 2:W // 2 + W // 8] * 0.8).astype(d.dtype)   # a depth step down to the last row
    xyz = oracle_mod.sphere_cloud(d, 2)
    nrm, _ = oracle_mod.f360_normals(xyz, H, W, 0.05, 4.0, 1)
    st = Frame360Stages(_mk(hip_lib, 2))
    labels, planes = st.plane_fit(xyz, nrm, H, W, 10, 0.08, 0.05, 0.01, 1)
    labels_ref, planes_ref = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 10, 0.08, 0.05, 0.01, 1)
    assert np.array_equal(labels, labels_ref)
    assert (labels_ref >= 0).any()
    assert [p["root"] for p in planes] == [p["root"] for p in planes_ref]
    assert [p["count"] for p in planes] == [p["count"] for p in planes_ref]
    # the device normal map on the same ragged size (32 x 16 tiles, halo 6)
    nrm_dev = st.normals(xyz, H, W, 0.05, 4.0, 1)
    assert np.array_equal(np.isnan(nrm_dev[:, 0]), np.isnan(nrm[:, 0]))
    ok = ~np.isnan(nrm[:, 0])
    if ok.any():
        assert np.abs(nrm_dev[ok] - nrm[ok]).max() <= 1.2e-7


def _noisy_scene(oracle_mod, W, H, seed):
    """A range panorama whose planes have something to grow into: measurement noise on bands of rows / columns (regions there fall
    under min_inliers or fail the curvature test), a hole and a depth step."""
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=seed)
    rng = np.random.default_rng(seed)
    d = dA.astype(np.float32)
    for _ in range(6):
        r0, c0 = int(rng.integers(0, H - H // 8)), int(rng.integers(0, W - W // 8))
        hh, ww = int(rng.integers(3, H // 8)), int(rng.integers(3, W // 8))
        d[r0:r0 + hh, c0:c0 + ww] += rng.normal(0, 12.0, size=(hh, ww)).astype(np.float32)       # 12 mm noise: inside 2 cm of the wall
    d[H // 3:H // 3 + 3, W // 4:W // 4 + 9] = 0
    d[H // 2:, W // 2:W // 2 + W // 8] *= 0.8
    return np.clip(d, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize("W,H,seed", [(512, 256, 3), (250, 101, 5), (700, 37, 7), (65, 70, 9)])
def test_plane_refinement_matches_oracle(hip_lib, oracle_mod, W, H, seed):
    """segmentAndRefine's refinement (Frame360.h:977): PCL's two sequential raster passes (oracle, literal) against the device's
    Jacobi sweeps -- refined labels identical pixel for pixel, inlier counts identical, extent descriptors to float rounding; the
    planes' centroid / normal / d / curvature stay those of `segment`."""
    from rgbd360_amd.register import Frame360Stages
    d = _noisy_scene(oracle_mod, W, H, seed)
    xyz = oracle_mod.sphere_cloud(d, 2)
    nrm, _ = oracle_mod.f360_normals(xyz, H, W, 0.05, 4.0, 1)
    st = Frame360Stages(_mk(hip_lib, 2))
    labels0, planes0 = st.plane_fit(xyz, nrm, H, W, 30, 0.06, 0.05, 0.002, 1)
    st.set_refinement(True, 0.02)
    labels1, planes1 = st.plane_fit(xyz, nrm, H, W, 30, 0.06, 0.05, 0.002, 1)
    stats = st.refinement_stats()
    # the oracle refines the device's own `segment` result: the test isolates the refinement step
    labels_ref, planes_ref, changed = oracle_mod.f360_plane_refine(xyz, H, W, labels0, planes0, 0.02)
    assert changed > 0 and stats["pixels_relabelled"] == changed, (changed, stats)
    assert np.array_equal(labels1, labels_ref)
    assert [p["root"] for p in planes1] == [p["root"] for p in planes0] == [p["root"] for p in planes_ref]
    assert [p["count"] for p in planes1] == [p["count"] for p in planes_ref]
    assert sum(p["count"] for p in planes1) == sum(p["count"] for p in planes0) + changed
    for a, b, c in zip(planes1, planes_ref, planes0):
        assert np.array_equal(a["centroid"], c["centroid"]) and np.array_equal(a["normal"], c["normal"]) and a["d"] == c["d"]
        assert abs(a["area_moment"] - b["area"]) <= 1e-4 * max(b["area"], 1e-3) and abs(a["elongation"] - b["elongation"]) <= 1e-3 * b["elongation"]
    # only non-plane pixels change, and every new inlier lies within the threshold of its plane
    grown = labels1 != labels0
    plane_roots = {p["root"] for p in planes0}
    assert not np.isin(labels0[grown], list(plane_roots)).any() and np.isin(labels1[grown], list(plane_roots)).all()
    model = {p["root"]: (p["normal"], p["d"]) for p in planes0}
    pts = xyz.reshape(H, W, 3)[grown]
    dist = np.array([abs(float(np.dot(model[l][0], p) + model[l][1])) for l, p in zip(labels1[grown], pts)])
    assert (dist < 0.02 + 1e-6).all()
    # switching it off again restores plain `segment`
    st.set_refinement(False)
    labels2, planes2 = st.plane_fit(xyz, nrm, H, W, 30, 0.06, 0.05, 0.002, 1)
    assert np.array_equal(labels2, labels0) and [p["count"] for p in planes2] == [p["count"] for p in planes0]


def test_frame_planes_recovers_the_room_walls(hip_lib, oracle_mod):
    """Functional known answer for the chained device pipeline (range image -> cloud -> normals -> regions): the six walls
    of the synthetic room come out within 1 degree / 1 cm (SURVEY.md 8c bar for the PCL-based rows)."""
    from rgbd360_amd.register import Frame360Stages
    (rgbA, dA), _, _ = synth.make_pair(512, 256, seed=5)
    out = Frame360Stages(_mk(hip_lib, 3)).frame_planes(dA, convention=2, angular_threshold=0.03)
    big = [p for p in out["planes"] if p["count"] > 1000]
    cam = synth.CAM_A
    truth = []           # (normal towards the camera, distance) of every wall, camera at CAM_A with identity orientation
    for ax in range(3):
        for sgn, bound in ((-1.0, synth.ROOM_HI[ax]), (1.0, synth.ROOM_LO[ax])):
            n = np.zeros(3)
            n[ax] = sgn
            truth.append((n, abs(bound - cam[ax])))
    found = 0
    for n_true, dist in truth:
        hits = [p for p in big if np.dot(p["normal"], n_true) > np.cos(np.radians(1.0)) and abs(p["d"] - dist) < 0.01]
        found += bool(hits)
    assert found == 6, [(p["count"], p["normal"], p["d"]) for p in big]
    assert out["labels"].shape == (256, 512) and np.isfinite(out["normals"]).any()
    # hull-true areas (Frame360.h:1025-1031): every wall of the 3 x 6 x 8 m room lies inside the 6 m depth band, so its region's hull
    # is the wall minus the margin where no normal exists (half the 8-pixel smoothing window + the depth-edge band: ~0.15-0.3 m per
    # side at 512 x 256) -- metric, where the density-weighted moment rectangle reads the 48 m2 floor as ~16 m2
    true_area = {0: 48.0, 1: 24.0, 2: 18.0}      # by the axis of the wall's normal: floor / ceiling, y walls (8 x 3), z walls (6 x 3)
    best = {}
    for p in big:
        ax = int(np.argmax(np.abs(p["normal"])))
        if abs(p["normal"][ax]) < 0.999:
            continue
        assert p["area"] <= true_area[ax] * 1.001 and p["hull_points"] >= 3, (ax, p["area"], p["area_moment"])      # metric: never more than the wall
        best[ax] = max(best.get(ax, 0.0), p["area"])
    # the largest region of every orientation is most of its wall (at 512 x 256 a pixel at the far end of an 8 m wall seen at 36 degrees
    # covers 0.12 m of it: the ~6-pixel margin costs 0.7 m per end; the wall behind the camera is cut in two by the panorama's first /
    # last column).  Exact-hull comparisons: test_pbmap_registration_seeds_the_dense_alignment, the 2048 x 1024 and 4096 x 2048 tests.
    assert sorted(best) == [0, 1, 2] and all(best[ax] > 0.70 * true_area[ax] for ax in best), best


@pytest.mark.parametrize("W,H,n_pyr", [(480, 80, 3), (250, 101, 3), (1000, 37, 2)])
def test_ragged_sizes_match_oracle(hip_lib, oracle_mod, W, H, n_pyr):
    """Sizes that are neither powers of two nor multiples of the block span (the real rig gives 1920x320): odd rows and
    columns through the pyramids, partially filled last step of the fused pass, seam columns at cols/8."""
    pair = synth.make_pair(W, H, seed=21)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=n_pyr)
    for level in range(n_pyr):
        ora.prepare_level(level)
        assert reg.level_dims(level) == ora.level_dims(level)
        for name in ("gray_src", "depth_trg", "gx", "dgy"):
            assert np.array_equal(reg.plane(name, level), ora.plane(name, level)), (name, level)
        assert np.array_equal(reg.warp_indices(level, T), ora.warp_indices(level, T))
        e = reg.eval(level, T, 2)
        _, err2, nvalid = ora.error(level, T, 2)
        H_, g_, Hd, gd, nvis = ora.hessgrad(level, T, 2)
        assert e["n_valid"] == nvalid and e["n_visible"] == nvis
        assert abs(e["err2"] - err2) <= ERR2_RTOL * max(1.0, err2)
        assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
    rc = reg.alignFrames360(np.eye(4), 2)
    st, pose_ref = ora.align360(np.eye(4), 2)
    assert rc == st
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV
    assert reg.num_iterations == list(ora.result.iters)[:n_pyr]
    _assert_libm_oracle_agrees(reg, ora, 2, n_pyr, status=st)


def test_non_default_parameters_match_oracle(hip_lib, oracle_mod, small_pair):
    """The setters of RPI.h:224-269 reach the kernels: depth band, standard deviations, no seam mask, 2 levels."""
    (rgbA, dA), (rgbB, dB), T = small_pair
    from rgbd360_amd.register import RegisterPhotoICP
    reg = RegisterPhotoICP()
    reg.setNumPyr(2); reg.setMinDepth(1.6); reg.setMaxDepth(4.5); reg.setGrayVariance(3.0 / 255); reg.setDepthVariance(0.1)
    reg.setMaskSeams(False)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=2, min_depth=1.6, max_depth=4.5, sigma_photo=3.0 / 255, sigma_depth=0.1, mask_seams=0,
                            math_mode=1, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    for level in range(2):
        ora.prepare_level(level)
        assert np.array_equal(reg.plane("gx", level), ora.plane("gx", level))          # unmasked seams
        assert np.array_equal(reg.plane("depth_src", level), ora.plane("depth_src", level))
        la, lb = reg.lut(level), ora.lut(level)
        assert np.array_equal(la[:, 0] != -10000, lb[:, 0] != -10000) and (lb[:, 0] == -10000).any()
    e = reg.eval(0, T, 2)
    _, err2, nvalid = ora.error(0, T, 2)
    assert e["n_valid"] == nvalid and abs(e["err2"] - err2) <= ERR2_RTOL * err2
    rc = reg.alignFrames360(np.eye(4), 2)
    st, pose_ref = ora.align360(np.eye(4), 2)
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rc == st == 0 and rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV
    _assert_libm_oracle_agrees(reg, ora, 2, 2)


def _fake_rig(seed=3, rows=60, cols=80):
    """8 random sensor images + rig-like extrinsics (45-degree steps about the up axis, small offsets)."""
    rng = np.random.default_rng(seed)
    rgb8 = rng.integers(0, 256, size=(8, rows, cols, 3), dtype=np.uint8)
    d8 = rng.integers(400, 6000, size=(8, rows, cols)).astype(np.uint16)
    d8[:, :5, :7] = 0
    Rt = []
    for s in range(8):
        R = synth.rodrigues([1.0, 0.0, 0.0], np.radians(45.0 * s + rng.uniform(-1, 1)))
        Rt.append(np.linalg.inv(synth.make_pose(R, rng.normal(size=3) * 0.03)).astype(np.float32))
    K = (rows * 262.5 / 240, rows * 262.5 / 240, cols / 2 - 0.5, rows / 2 - 0.5)
    return rgb8, d8, np.stack(Rt), K


def test_stitch_sphere_bit_exact(hip_lib, oracle_mod):
    """Frame360::stitchSphericalImage on the device equals the CPU restatement byte for byte."""
    from rgbd360_amd.register import stitch_sphere
    rgb8, d8, Rt, K = _fake_rig()
    reg = _mk(hip_lib, 3)
    a, b = stitch_sphere(reg, rgb8, d8, Rt, K)
    a_ref, b_ref = oracle_mod.stitch_sphere(rgb8, d8, Rt, K)
    assert a.shape == (int(60 * 8 * 0.5 * 60.0 / 180), 480, 3)
    assert (b_ref > 0).mean() > 0.3
    assert np.array_equal(a, a_ref) and np.array_equal(b, b_ref)


def test_concurrent_contexts_give_identical_poses(hip_lib):
    """rgbd360_align360_begin / _finish: several alignments in flight on one GPU (one context and stream per pair)
    produce exactly the poses of the one-at-a-time schedule."""
    from rgbd360_amd.batch import align_sequence, align_sequence_concurrent
    frames = {k: synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(6)}
    poses, status, iters = align_sequence(_mk(hip_lib, 3), lambda k: frames[k], 0, 5, 2)
    regs = [_mk(hip_lib, 3) for _ in range(3)]
    p2, s2, i2 = align_sequence_concurrent(regs, lambda k: frames[k], 0, 5, 2)
    assert np.array_equal(poses, p2) and np.array_equal(status, s2) and np.array_equal(iters, i2)
    # begin without finish, then a fresh begin: the context recovers
    r = regs[0]
    r.setTargetFrame(*frames[0]); r.setSourceFrame(*frames[1])
    r.alignFrames360_begin(np.eye(4), 2)
    assert r.alignFrames360_finish() == 0
    assert np.array_equal(r.getOptimalPose(), poses[0])


def test_native_batch_entry_equals_pairwise_alignment(hip_lib):
    """rgbd360_align360_batch (one C call for a whole sequence) returns exactly the poses of the pair-by-pair schedule,
    for 1, 2 and 3 sub-chunks in flight, and for an occlusion mode."""
    from rgbd360_amd.batch import align_sequence
    frames = [synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(6)]
    poses, status, iters = align_sequence(_mk(hip_lib, 3), lambda k: frames[k], 0, 5, 2)
    reg = _mk(hip_lib, 3)
    for k in (1, 2, 3):
        p2, s2, i2 = reg.alignSequence(frames, method=2, n_inflight=k)
        assert np.array_equal(poses, p2) and np.array_equal(status, s2) and np.array_equal(iters, i2), k
    from rgbd360_amd.batch import align_sequence_native
    pn, sn, inn = align_sequence_native(reg, lambda k: frames[k], 1, 4, 2, n_inflight=2)
    assert np.array_equal(pn, poses[1:4]) and np.array_equal(sn, status[1:4]) and np.array_equal(inn, iters[1:4])
    p3, s3, i3 = reg.alignSequence(frames, method=2, occlusion=2, n_inflight=2)
    one = _mk(hip_lib, 3)
    one.setTargetFrame(*frames[2]); one.setSourceFrame(*frames[3])
    assert one.alignFrames360(np.eye(4), 2, 2) == s3[2]
    assert np.array_equal(one.getOptimalPose(), p3[2])
    # frames already in HBM (device pointers from the same HIP runtime the library links): the same poses, no PCIe traffic in the call
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return p.value
    rgb_d = [to_device(f[0]) for f in frames]
    dep_d = [to_device(f[1]) for f in frames]
    p4, s4, i4 = reg.alignSequenceDev(rgb_d, dep_d, 128, 256, 0, method=2, n_inflight=2)
    for q in rgb_d + dep_d:
        hip.hipFree(C.c_void_p(q))
    assert np.array_equal(poses, p4) and np.array_equal(status, s4) and np.array_equal(iters, i4)
    # degenerate inputs
    assert reg.alignSequence(frames[:1])[0].shape == (0, 4, 4)
    from rgbd360_amd.register import Rgbd360Error
    with pytest.raises(Rgbd360Error):
        reg.alignSequence(frames, n_inflight=0)


@pytest.mark.gpu
def test_native_batch_threads_blank_frame_and_odd_spans(hip_lib):
    """The threaded sequence entry with a frame that has no valid depth in the middle (the pair that has it as source ends
    NO_VALID_PIXELS, the others are unaffected), uneven sub-chunk spans, more sub-chunks than pairs, float depth and row-padded host images."""
    from rgbd360_amd.batch import align_sequence
    frames = [synth.render(synth.trajectory_pose(k, 11), 256, 128, 11) for k in range(8)]
    frames[4] = (frames[4][0], np.zeros_like(frames[4][1]))
    poses, status, iters = align_sequence(_mk(hip_lib, 3), lambda k: frames[k], 0, 7, 2)
    assert status[3] != 0 and (np.delete(status, [3]) == 0).all()      # a blank TARGET still aligns photometrically
    reg = _mk(hip_lib, 3)
    for k in (1, 2, 3, 5, 7, 12):
        p2, s2, i2 = reg.alignSequence(frames, method=2, n_inflight=k)
        assert np.array_equal(poses, p2) and np.array_equal(status, s2) and np.array_equal(iters, i2), k
    # float32 metres + row-padded views (cv::Mat ROI style): steps are taken from the first frame, so pad all alike
    def padded(f):
        rgb = np.zeros((128, 300, 3), np.uint8); rgb[:, :256] = f[0]
        d = np.zeros((128, 290), np.float32); d[:, :256] = f[1].astype(np.float32) * np.float32(0.001)
        return rgb, d
    pads = [padded(f) for f in frames]
    import ctypes as C
    from rgbd360_amd._lib import Result
    n = len(frames) - 1
    rgb_ptrs = (C.c_void_p * len(frames))(*[q[0].ctypes.data for q in pads])
    dep_ptrs = (C.c_void_p * len(frames))(*[q[1].ctypes.data for q in pads])
    out = np.zeros(n * 16, np.float32)
    res = (Result * n)()
    rc = reg._L.rgbd360_align360_batch(reg._ctx(), len(frames), rgb_ptrs, 300 * 3, dep_ptrs, 290 * 4, 1, 128, 256, None, 2, 0, 3,
                                       out.ctypes.data_as(C.POINTER(C.c_float)), res)
    assert rc == 0
    f32 = [(f[0], f[1].astype(np.float32) * np.float32(0.001)) for f in frames]
    p3, s3, i3 = align_sequence(_mk(hip_lib, 3), lambda k: f32[k], 0, 7, 2)
    for j in range(n):
        assert res[j].status == s3[j]
        assert np.array_equal(out[16 * j:16 * j + 16].reshape(4, 4).T, p3[j]), j


@pytest.mark.parametrize("W,H,n_pyr", [(250, 101, 3), (480, 80, 3), (1000, 37, 2), (66, 18, 1), (130, 34, 2)])
def test_lockstep_engine_ragged_sizes(hip_lib, W, H, n_pyr):
    """The sequence engine's fused frame set-up (one tiled launch per level: records + next level's planes) and slot-batched passes
    on sizes that are no multiple of the 64 x 16 tile, of 4 columns or of 2: the poses of rgbd360_align360_batch are bit-identical
    to the pair-by-pair path (whose set-up kernels work per pixel), for host frames and for float depth."""
    from rgbd360_amd.batch import align_sequence
    frames = [synth.render(synth.trajectory_pose(k, 7), W, H, 7) for k in range(5)]
    poses, status, iters = align_sequence(_mk(hip_lib, n_pyr), lambda k: frames[k], 0, 4, 2)
    reg = _mk(hip_lib, n_pyr)
    for k in (1, 3, 4):
        p2, s2, i2 = reg.alignSequence(frames, method=2, n_inflight=k)
        assert np.array_equal(status, s2) and np.array_equal(iters, i2) and np.array_equal(poses, p2), k
    for m in (0, 1):
        pm, sm, im = align_sequence(_mk(hip_lib, n_pyr), lambda k: frames[k], 0, 4, m)
        p3, s3, i3 = reg.alignSequence(frames, method=m, n_inflight=4)
        assert np.array_equal(sm, s3) and np.array_equal(im, i3) and np.array_equal(pm, p3), m
    f32 = [(f[0], f[1].astype(np.float32) * np.float32(0.001)) for f in frames]
    pf, sf, itf = align_sequence(_mk(hip_lib, n_pyr), lambda k: f32[k], 0, 4, 2)
    p4, s4, i4 = reg.alignSequence(f32, method=2, n_inflight=2)
    assert np.array_equal(sf, s4) and np.array_equal(itf, i4) and np.array_equal(pf, p4)


# ---- occlusion-aware variants (RPI.h:3232-4249, sequential semantics) ---------------------------
_occluder_pair = synth.add_occluder
_occ_poses = synth.occlusion_test_poses


@pytest.mark.parametrize("occlusion", [1, 2])
@pytest.mark.parametrize("method", [0, 1, 2])
def test_eval_occlusion_parity(hip_lib, oracle_mod, small_pair, method, occlusion):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, _occluder_pair(small_pair))
    conflicts = 0
    for level in range(3):
        for pose in _occ_poses(T):
            e = reg.eval(level, pose, method, occlusion)
            _, sp, sd, n_p, n_d = ora.error_occ(level, pose, method, occlusion)
            H, g, Hd, gd, nvis = ora.hessgrad_occ(level, pose, method, occlusion)
            assert list(e["n_split"]) == [n_p, n_d], (level, list(e["n_split"]), n_p, n_d)     # integer work: exact
            assert e["n_visible"] == nvis
            assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp)), (e["err2_split"], sp)
            assert abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd)), (e["err2_split"], sd)
            scale_h = max(np.abs(Hd).max(), 1e-30)
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * scale_h
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.sqrt(scale_h))
            # the variants must actually differ from the plain pass somewhere, or the test proves nothing
            plain = reg.eval(level, pose, method, 0)
            conflicts += int(plain["n_visible"] != e["n_visible"]) + int(list(plain["n_split"]) != list(e["n_split"]))
    assert conflicts > 0


@pytest.mark.parametrize("occlusion,method", [(1, 2), (2, 0), (2, 1), (2, 2)])
def test_align_occlusion_matches_oracle(hip_lib, oracle_mod, small_pair, occlusion, method):
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, _occluder_pair(small_pair))
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    st, pose_ref = ora.align360(np.eye(4), method, occlusion)
    assert rc == st == 0
    assert reg.num_iterations == list(ora.result.iters)[:3]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    assert abs(reg.avResidual - ora.result.err_final) <= 1e-5 * max(1.0, ora.result.err_final)
    assert abs(reg.SSO - ora.result.sso) < 1e-6
    # the alignment still lands near the true motion (the billboard is a moving outlier for the room)
    rot, trans = synth.pose_error(reg.getOptimalPose(), T)
    assert rot < 0.02 and trans < 0.05, (rot, trans)
    # and against the reference-faithful libm oracle (sequential semantics of RPI.h:3232-4249): north-star tolerance
    _assert_libm_oracle_agrees(reg, ora, method, 3, occlusion)


def test_occlusion_fused_schedule_equals_the_three_launch_one(hip_lib, small_pair):
    """Round 4: the occlusion-aware alignments run {k_occ_build_fs, k_eval_occ} per iteration (the solve of the previous pass in the
    prologue of the build, the pass gated by the state the build writes).  Same poses, iteration counts, residuals and status, bit
    for bit, as {k_occ_build, k_eval_occ, k_solve} (rgbd360_debug_set_schedule(ctx, 1, 0), rgbd360_hip_diag.h) -- from the identity and
    from a guess, twice in a row on one context (the generation-tagged heads and the double-buffered state carry over)."""
    (rgbA, dA), (rgbB, dB), T = _occluder_pair(small_pair)
    out = {}
    for fused in ("1", "0"):
        reg = _mk(hip_lib, 3)
        reg.debug_set_schedule(fused_solve=True, fused_occ=fused == "1")
        reg.setTargetFrame(rgbA, dA)
        reg.setSourceFrame(rgbB, dB)
        rows = []
        guess = synth.make_pose(synth.rodrigues(np.array([0.0, 1.0, 0.0]), 0.004), np.array([0.01, 0.0, -0.005]))
        for occlusion, method in ((1, 2), (2, 0), (2, 1), (2, 2), (1, 0)):
            for g in (np.eye(4), guess, np.eye(4)):
                rc = reg.alignFrames360(g, method, occlusion)
                rows.append((rc, list(reg.num_iterations), reg.getOptimalPose().copy(), reg.avResidual, reg.SSO))
        out[fused] = rows
    assert len(out["1"]) == len(out["0"]) == 15
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0] and a[1] == b[1], (a, b)
        assert np.array_equal(a[2], b[2]), (a, b)
        assert (a[3] == b[3] or (a[3] != a[3] and b[3] != b[3])) and a[4] == b[4], (a, b)


def test_occlusion1_single_modality_returns_guess(hip_lib, oracle_mod, small_pair):
    """errorPhotoICP_sphereOcc1 adds avPhotoResidual + avDepthResidual: with PHOTO (or DEPTH) only, the unused term is
    0/0, the error is NaN and the reference's loop never runs -- the guess comes back."""
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair)
    guess = synth.make_pose(synth.rodrigues(np.array([0.0, 0.0, 1.0]), 0.01), np.array([0.01, 0.0, 0.0]))
    rc = reg.alignFrames360(guess, 0, 1)
    st, pose_ref = ora.align360(guess, 0, 1)
    assert rc == st == 2
    assert np.allclose(reg.getOptimalPose(), guess.astype(np.float32)) and np.allclose(pose_ref, guess.astype(np.float32))
    assert reg.num_iterations == [0, 0, 0]


# ---- the HIP path against the committed golden fixtures (tests/golden/, device-arithmetic entries) -------------------
def _golden():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "pair_256x128.npz"))
    return z, json.load(open(os.path.join(here, "golden", "oracle_256x128.json")))


@pytest.mark.parametrize("method", [0, 1, 2])
def test_hip_matches_golden_fixture(hip_lib, method):
    z, j = _golden()
    ref = j["runs"]["math1/method%d" % method]
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(z["rgbA"], z["dA"])
    reg.setSourceFrame(z["rgbB"], z["dB"])
    rc = reg.alignFrames360(np.eye(4), method)
    assert rc == ref["status"] and reg.num_iterations == ref["iters"]
    rot, trans = synth.pose_error(reg.getOptimalPose(), np.array(ref["pose"]))
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV
    assert abs(reg.SSO - ref["sso"]) < 1e-6
    g = ref["at_gt_level1"]
    e = reg.eval(1, z["T_gt"], method)
    assert e["n_valid"] == g["n_valid"] and e["n_visible"] == g["n_visible"]
    assert abs(e["err2"] - g["err2"]) <= ERR2_RTOL * max(1.0, g["err2"])
    assert np.abs(e["H64"] - np.array(g["H64"])).max() <= HG_RTOL * np.abs(np.array(g["H64"])).max()


@pytest.mark.parametrize("occlusion,method", [(1, 2), (2, 0), (2, 1), (2, 2)])
def test_hip_occlusion_matches_golden_fixture(hip_lib, occlusion, method):
    z, j = _golden()
    ref = j["occlusion"]["math1/occ%d/method%d" % (occlusion, method)]
    (rgbA, dA), (rgbB, dB), T = synth.add_occluder(((z["rgbA"], z["dA"]), (z["rgbB"], z["dB"]), z["T_gt"]))
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    assert rc == ref["status"] and reg.num_iterations == ref["iters"]
    rot, trans = synth.pose_error(reg.getOptimalPose(), np.array(ref["pose"]))
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV
    g = ref["at_probe_level1"]
    e = reg.eval(1, synth.occlusion_test_poses(T)[2], method, occlusion)
    assert list(e["n_split"]) == [g["n_photo"], g["n_depth"]] and e["n_visible"] == g["n_visible"]
    assert np.abs(e["H64"] - np.array(g["H64"])).max() <= HG_RTOL * np.abs(np.array(g["H64"])).max()


# ---- pinhole single-sensor alignment (RPI.h:4254-4512) -------------------------------------------
def _pinhole_ctx(hip_lib, oracle_mod, math_mode=1, depth_f32=False):
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77, depth_f32=depth_f32)
    reg = _mk(hip_lib, 3, setMaskSeams=False)
    reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    return reg, ora, T


def test_pinhole_warp_indices_bit_exact(hip_lib, oracle_mod):
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    for level in range(3):
        for pose in _poses(T):
            a, b = reg.warp_indices_pinhole(level, pose), ora.warp_indices_pinhole(level, pose)
            assert (b[:, 0] >= 0).mean() > 0.5
            assert np.array_equal(a, b), (level, int((a != b).any(axis=1).sum()))


@pytest.mark.parametrize("method", [0, 1, 2])
def test_pinhole_eval_parity(hip_lib, oracle_mod, method):
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    for level in range(3):
        for pose in _poses(T):
            e = reg.eval_pinhole(level, pose, method)
            _, sp, sd, n_p, n_d = ora.error_pinhole(level, pose, method)
            H, g, Hd, gd, nrows = ora.hessgrad_pinhole(level, pose, method)
            assert list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nrows           # integer work: exact
            assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp))
            assert abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd))
            scale_h = np.abs(Hd).max()
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * scale_h
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.sqrt(scale_h))


@pytest.mark.parametrize("method,depth_f32", [(0, False), (1, False), (2, False), (2, True)])
def test_pinhole_align_matches_oracle(hip_lib, oracle_mod, method, depth_f32):
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod, depth_f32=depth_f32)
    rc = reg.alignFrames(np.eye(4), method)
    st, pose_ref = ora.align_pinhole(np.eye(4), method)
    assert rc == st
    assert rc == (2 if method == 0 else 0)          # PHOTO only: x / nValidDepthPts = NaN in the reference, guess returned
    assert reg.num_iterations == list(ora.result.iters)[:3]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV, (rot, trans)
    if method != 0:
        assert abs(reg.avResidual - ora.result.err_final) <= 1e-3 * max(1.0, ora.result.err_final)
        assert np.allclose(reg.getHessian(), np.asarray(list(ora.result.hessian)).reshape(6, 6).T, rtol=1e-4,
                           atol=1e-4 * np.abs(reg.getHessian()).max())
        # and against the reference-faithful libm / roundf oracle: the north-star tolerance
        ora0 = oracle_mod.Oracle(n_pyr=3, math_mode=0, reduce_mode=0, mask_seams=0)
        (rgbA, dA), (rgbB, dB), _, K = synth.make_pinhole_pair(320, 240, seed=77, depth_f32=depth_f32)
        ora0.set_camera(*K); ora0.set_target(rgbA, dA); ora0.set_source(rgbB, dB)
        st0, pose0 = ora0.align_pinhole(np.eye(4), method)
        assert st0 == 0
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose0)
        assert rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)


def test_pinhole_argument_errors(hip_lib):
    from rgbd360_amd.register import Rgbd360Error
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    reg = _mk(hip_lib, 3, setMaskSeams=False)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    with pytest.raises(Rgbd360Error):
        reg.alignFrames(np.eye(4), 2)                 # no camera matrix yet
    reg.setCameraMatrix(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]]))
    with pytest.raises(Rgbd360Error):
        reg.alignFrames(np.eye(4), 2, occlusion=3)
    assert reg.alignFrames(np.eye(4), 2) == 0
    seams = _mk(hip_lib, 3)                           # default params mask the panorama seams: refused for a pinhole image
    seams.setCameraMatrix(K); seams.setTargetFrame(rgbA, dA); seams.setSourceFrame(rgbB, dB)
    with pytest.raises(Rgbd360Error):
        seams.alignFrames(np.eye(4), 2)


@pytest.mark.parametrize("method", [1, 2])
def test_hip_pinhole_matches_golden_fixture(hip_lib, method):
    z, j = _golden()
    G = j["pinhole"]
    ref = G["runs"]["math1/method%d" % method]
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    reg = _mk(hip_lib, 3, setMaskSeams=False)
    reg.setCameraMatrix(K); reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    rc = reg.alignFrames(np.eye(4), method)
    assert rc == ref["status"] and reg.num_iterations == ref["iters"]
    rot, trans = synth.pose_error(reg.getOptimalPose(), np.array(ref["pose"]))
    assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV, (rot, trans)
    g = ref["at_gt_level1"]
    e = reg.eval_pinhole(1, T, method)
    assert list(e["n_split"]) == [g["n_photo"], g["n_depth"]] and e["n_rows"] == g["n_rows"]
    assert np.abs(e["H64"] - np.array(g["H64"])).max() <= HG_RTOL * np.abs(np.array(g["H64"])).max()


# ---- pinhole occlusion-aware passes + salient-pixel list (RPI.h:1107-2030, 401-425, 590-690) ---------------------------------
def _pinhole_probe_pose(T):
    back = np.eye(4)
    back[2, 3] = 0.6            # the rendered motion pushed 0.6 m along the optical axis: up to four source pixels per target pixel
    return back @ T


@pytest.mark.parametrize("occ", [1, 2])
@pytest.mark.parametrize("method", [0, 1, 2])
def test_pinhole_occlusion_eval_parity(hip_lib, oracle_mod, method, occ):
    """errorPhotoICP_Occ1/2 + calcHessGrad_Occ1/2 at one pose: counts (integer work, the z-buffer's accept chain included) exact,
    sums within the float tolerances of the plain pass."""
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    longest = 0
    for level in range(3):
        for pose in list(_poses(T)) + [_pinhole_probe_pose(T)]:
            idx = ora.warp_indices_pinhole(level, pose)
            v = idx[:, 0] >= 0
            longest = max(longest, int(np.unique(idx[v, 0] * 4096 + idx[v, 1], return_counts=True)[1].max()))
            e = reg.eval_pinhole(level, pose, method, occ)
            _, sp, sd, n_p, n_d = ora.error_pinhole_occ(level, pose, method, occ)
            H, g, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pose, method, occ)
            assert list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nvis, (level, list(e["n_split"]), n_p, n_d, e["n_rows"], nvis)
            assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp))
            assert abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd))
            scale_h = max(np.abs(Hd).max(), 1e-30)
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * scale_h
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.sqrt(scale_h))
            if method == 1:
                assert not e["H64"].any()          # as written: both row sums test the photometric residual
    assert longest >= 3


@pytest.mark.parametrize("push", [3.0, 20.0, 200.0])
def test_pinhole_occlusion_long_arrival_lists(hip_lib, oracle_mod, push):
    """Poses that pile many source pixels on one target pixel (a zoom-out, a push, a collapse of the image onto a few pixels): lists
    longer than a target's slot row are visited by scanning the key array over the box of their arrivals -- the same sequential
    z-buffer semantics, counts exact, whatever the list length.  Both passes of both occlusion modes, every level."""
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    P = np.eye(4)
    P[2, 3] = push
    pose = P @ T
    longest = 0
    for level in range(3):
        idx = ora.warp_indices_pinhole(level, pose)
        v = idx[:, 0] >= 0
        longest = max(longest, int(np.unique(idx[v, 0] * 4096 + idx[v, 1], return_counts=True)[1].max()))
        for method, occ in ((2, 1), (2, 2), (0, 1)):
            e = reg.eval_pinhole(level, pose, method, occ)
            _, sp, sd, n_p, n_d = ora.error_pinhole_occ(level, pose, method, occ)
            H, g, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pose, method, occ)
            assert list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nvis, (level, method, occ, list(e["n_split"]), n_p, n_d, e["n_rows"], nvis)
            assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp)) and abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd))
            scale_h = max(np.abs(Hd).max(), 1e-30)
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * scale_h
            assert np.abs(e["g64"] - gd).max() <= HG_RTOL * max(np.abs(gd).max(), 1e-3 * np.sqrt(scale_h))
    assert longest > 8          # beyond the slot rows: the box scans ran


def test_pinhole_occlusion_eval_occlusion_zero_is_the_plain_pass(hip_lib, oracle_mod):
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    a, b = reg.eval_pinhole(1, T, 2), reg.eval_pinhole(1, T, 2, 0)
    assert np.array_equal(a["H64"], b["H64"]) and list(a["n_split"]) == list(b["n_split"])


@pytest.mark.parametrize("method,occ", [(2, 1), (2, 2), (0, 1), (1, 1)])
def test_pinhole_occlusion_align_matches_oracle(hip_lib, oracle_mod, method, occ):
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    rc = reg.alignFrames(np.eye(4), method, occ)
    st, pose_ref = ora.align_pinhole(np.eye(4), method, occ)
    assert rc == st
    # only PHOTO_DEPTH under occlusion 1 optimises anything: a single modality is 0 / 0 in the other average, and occlusion 2's error
    # gate (target depth against the point's INVERSE depth) leaves no depth residual on the fine levels -- as the source is written
    assert rc == (0 if (method, occ) == (2, 1) else 2)
    assert reg.num_iterations == list(ora.result.iters)[:3]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV, (rot, trans)
    assert abs(reg.SSO - ora.result.sso) <= 1e-6
    if rc == 0:
        assert sum(reg.num_iterations) >= 3
        assert abs(reg.avResidual - ora.result.err_final) <= 1e-3 * max(1.0, ora.result.err_final)
        assert np.allclose(reg.getHessian(), np.asarray(list(ora.result.hessian)).reshape(6, 6).T, rtol=1e-4, atol=1e-4 * np.abs(reg.getHessian()).max())
        r0, t0 = synth.pose_error(np.eye(4), T)
        r1, t1 = synth.pose_error(reg.getOptimalPose(), T)
        assert r1 < 0.5 * r0 and t1 < 0.5 * t0
        ora0 = oracle_mod.Oracle(n_pyr=3, math_mode=0, reduce_mode=0, mask_seams=0)       # the reference-faithful libm / roundf oracle
        (rgbA, dA), (rgbB, dB), _, K = synth.make_pinhole_pair(320, 240, seed=77)
        ora0.set_camera(*K); ora0.set_target(rgbA, dA); ora0.set_source(rgbB, dB)
        st0, pose0 = ora0.align_pinhole(np.eye(4), method, occ)
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose0)
        assert st0 == 0 and rot <= ROT_TOL and trans <= TRANS_TOL, (rot, trans)


def test_hip_pinhole_occlusion_matches_golden_fixture(hip_lib):
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pinhole_occ.json")) as f:
        G = json.load(f)
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    reg = _mk(hip_lib, 3, setMaskSeams=False)
    reg.setCameraMatrix(K); reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    for occ in (1, 2):
        for method in (0, 1, 2):
            ref = G["occ"]["math1/occ%d/method%d" % (occ, method)]
            rc = reg.alignFrames(np.eye(4), method, occ)
            assert rc == ref["status"] and reg.num_iterations == ref["iters"]
            rot, trans = synth.pose_error(reg.getOptimalPose(), np.array(ref["pose"]))
            assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV
            for name, pp, level in (("at_gt_level1", T, 1), ("at_probe_level0", _pinhole_probe_pose(T), 0)):
                g = ref[name]
                e = reg.eval_pinhole(level, pp, method, occ)
                assert list(e["n_split"]) == [g["n_photo"], g["n_depth"]] and e["n_rows"] == g["n_visible"]
                assert np.abs(e["H64"] - np.array(g["H64"])).max() <= HG_RTOL * max(np.abs(np.array(g["H64"])).max(), 1e-30)


def test_pinhole_saliency_list_mode(hip_lib, oracle_mod):
    """useSaliency(true): the error sums run over vSalientPixels (interior pixels with a salient TARGET gradient, used as source indices),
    the normal equations over every pixel."""
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    plain = reg.eval_pinhole(1, T, 2)
    reg.useSaliency(True)
    ora.use_saliency(True, 0.01)
    for level in range(3):
        for pose in _poses(T):
            for method in (0, 1, 2):
                e = reg.eval_pinhole(level, pose, method)
                _, sp, sd, n_p, n_d = ora.error_pinhole_salient(level, pose, method)
                H, g, Hd, gd, nrows = ora.hessgrad_pinhole(level, pose, method)
                assert list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nrows
                assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp)) and abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd))
                assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
    sal = reg.eval_pinhole(1, T, 2)
    assert sal["n_split"][0] < plain["n_split"][0] and np.array_equal(sal["H64"], plain["H64"])
    rc = reg.alignFrames(np.eye(4), 2)
    st, pose_ref = ora.align_pinhole(np.eye(4), 2)
    assert rc == st == 0 and reg.num_iterations == list(ora.result.iters)[:3]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV
    reg.useSaliency(False)
    back = reg.eval_pinhole(1, T, 2)
    assert list(back["n_split"]) == list(plain["n_split"])


def test_pinhole_occlusion_long_lists_are_walked_exactly(hip_lib, oracle_mod):
    """A pose that collapses the whole image onto a handful of target pixels (lists of thousands of source pixels): the sorted walk is
    exact and bounded whatever the list lengths."""
    reg, ora, T = _pinhole_ctx(hip_lib, oracle_mod)
    far = np.eye(4)
    far[2, 3] = 400.0
    idx = ora.warp_indices_pinhole(0, far)
    v = idx[:, 0] >= 0
    assert np.unique(idx[v, 0] * 4096 + idx[v, 1], return_counts=True)[1].max() > 2000
    for occ in (1, 2):
        e = reg.eval_pinhole(0, far, 2, occ)
        _, sp, sd, n_p, n_d = ora.error_pinhole_occ(0, far, 2, occ)
        nvis = ora.hessgrad_pinhole_occ(0, far, 2, occ)[4]
        assert list(e["n_split"]) == [n_p, n_d] and e["n_rows"] == nvis


@pytest.mark.parametrize("method", [0, 2])
def test_control_flow_matches_oracle_over_many_scenarios(hip_lib, oracle_mod, method):
    """The device-resident Gauss-Newton state machine (accept / reject, termination tests, level hand-over, speculative solve)
    against the oracle's restatement of alignFrames360 on pairs that exercise different paths: tiny and large motions, wrong
    initial guesses, iteration caps of 1 / 2 / 10, loose and tight tolerances, 2 to 4 pyramid levels."""
    from rgbd360_amd.register import RegisterPhotoICP
    rng = np.random.default_rng(2024 + method)
    checked = set()
    for case in range(14):
        trans = float(rng.choice([0.0, 0.01, 0.05, 0.12, 0.3]))
        rot_deg = float(rng.choice([0.0, 0.5, 2.0, 5.0]))
        n_pyr = int(rng.choice([2, 3, 4]))
        max_iters = int(rng.choice([1, 2, 10]))
        tol_res = float(rng.choice([1e-3, 1e-5, 5e-2]))
        pair = synth.make_pair(192, 96, seed=500 + case, trans=trans, rot_deg=rot_deg)
        (rgbA, dA), (rgbB, dB), T = pair
        guess = np.eye(4) if case % 3 else synth.make_pose(synth.rodrigues(rng.normal(size=3), 0.02), rng.normal(size=3) * 0.03)
        reg = RegisterPhotoICP()
        reg.setNumPyr(n_pyr)
        reg._p.max_iters = max_iters
        reg._p.tol_residual = tol_res
        reg.setTargetFrame(rgbA, dA)
        reg.setSourceFrame(rgbB, dB)
        ora = oracle_mod.Oracle(n_pyr=n_pyr, max_iters=max_iters, tol_residual=tol_res, math_mode=1, reduce_mode=1)
        ora.set_target(rgbA, dA)
        ora.set_source(rgbB, dB)
        rc = reg.alignFrames360(guess, method)
        st, pose_ref = ora.align360(guess, method)
        ctx = (case, trans, rot_deg, n_pyr, max_iters, tol_res)
        assert rc == st, ctx
        assert reg.num_iterations == list(ora.result.iters)[:n_pyr], (ctx, reg.num_iterations, list(ora.result.iters)[:n_pyr])
        rot, trans_e = synth.pose_error(reg.getOptimalPose(), pose_ref)
        assert rot <= 2e-5 and trans_e <= 2e-5, (ctx, rot, trans_e)       # several accepted steps: rounding differences add up
        checked.add((rc, tuple(min(i, 3) for i in reg.num_iterations)))
    assert len(checked) >= 5          # the scenarios did take different paths through the loop


def test_frame_planes_dev_equals_host_variant(hip_lib):
    """rgbd360_frame_planes_dev (depth in HBM, maps left in HBM) returns the plane list of the host-buffer entry point, and its
    device maps hold the same bytes."""
    import ctypes as C
    from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    (rgbA, dA), _, _ = synth.make_pair(256, 128, seed=5)
    st = Frame360Stages(RegisterPhotoICP())
    ref = st.frame_planes(dA, convention=2, angular_threshold=0.03)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), dA.nbytes) == 0 and hip.hipMemcpy(p, dA.ctypes.data_as(C.c_void_p), dA.nbytes, 1) == 0
    out = st.frame_planes_dev(p.value, 128, 256, 0, convention=2, angular_threshold=0.03)
    assert [(q["count"], q["root"]) for q in out["planes"]] == [(q["count"], q["root"]) for q in ref["planes"]]
    assert all(np.allclose(a["normal"], b["normal"]) and abs(a["d"] - b["d"]) < 1e-6 for a, b in zip(out["planes"], ref["planes"]))
    labels = np.empty(128 * 256, np.int32)
    nrm = np.empty((128 * 256, 3), np.float32)
    assert hip.hipMemcpy(labels.ctypes.data_as(C.c_void_p), C.c_void_p(out["labels_ptr"]), labels.nbytes, 2) == 0     # DeviceToHost
    assert hip.hipMemcpy(nrm.ctypes.data_as(C.c_void_p), C.c_void_p(out["normals_ptr"]), nrm.nbytes, 2) == 0
    hip.hipFree(p)
    assert np.array_equal(labels, np.asarray(ref["labels"]).reshape(-1))
    assert np.array_equal(np.nan_to_num(nrm), np.nan_to_num(ref["normals"]))


def test_eight_pyramid_levels(hip_lib, oracle_mod):
    """The deepest pyramid the ABI allows (n_pyr = 8, job tables of the batched set-up kernels full): planes bit-exact on every
    level, alignment identical to the oracle's."""
    pair = synth.make_pair(1024, 512, seed=21)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=8)
    for level in range(8):
        ora.prepare_level(level)
        for name in ("gray_src", "gray_trg", "depth_src", "depth_trg", "gx", "gy", "dgx", "dgy"):
            a, b = reg.plane(name, level), ora.plane(name, level)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, level)
        la, lb = reg.lut(level), ora.lut(level)
        valid = lb[:, 0] != -10000
        assert np.array_equal(la[:, 0] != -10000, valid) and np.array_equal(la[valid].view(np.uint32), lb[valid].view(np.uint32))
    rc = reg.alignFrames360(np.eye(4), 2)
    st, pose_ref = ora.align360(np.eye(4), 2)
    assert rc == st
    assert reg.num_iterations == list(ora.result.iters)[:8]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= 2e-5 and trans <= 2e-5, (rot, trans)


@pytest.mark.parametrize("trans,rot_deg", [(0.06, 2.0), (0.3, 10.0)])
def test_pbmap_registration_seeds_the_dense_alignment(hip_lib, oracle_mod, trans, rot_deg):
    """The reference's keyframe check (KFsphere_SLAM.cpp:129-163): planes of both frames -> RegisterPbMap -> the pose seeds
    alignFrames360 -> the dense pose must agree with the PbMap pose (`isApprox(.., 1e-1)`).  Device planes (chained
    Frame360 kernels) through the library's host matcher against oracle planes through the numpy restatement: same extent
    descriptors, same interpretation, same pose; the seeded dense alignment follows the oracle's accept / reject sequence."""
    from oracle import pbmap_ref
    from rgbd360_amd import pbmap
    from rgbd360_amd.register import Frame360Stages
    W, H = 512, 256
    pair = synth.make_pair(W, H, seed=5, trans=trans, rot_deg=rot_deg)
    (rgbA, dA), (rgbB, dB), T = pair
    st = Frame360Stages(_mk(hip_lib, 3))
    dev, ora_planes = [], []
    for d in (dA, dB):
        dev.append(st.frame_planes(d, convention=2, angular_threshold=0.03)["planes"])
        xyz = oracle_mod.sphere_cloud(d, 2)
        nrm, _ = oracle_mod.f360_normals(xyz, H, W, 0.05, 8.0, 1)
        labels_o, planes_o = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 40, 0.03, 0.05, 0.001, 1)
        # the checker's records get their hull fields from the EXACT hull of every inlier (Qhull): area, polygon mass centre
        for p_o in planes_o:
            area_h, center_h, nv = oracle_mod.f360_hull_stats(xyz, labels_o, p_o)
            p_o.update(area_moment=p_o["area"], hull_area_exact=area_h, center_hull=center_h.astype(np.float32), hull_points=nv)
        ora_planes.append(planes_o)
    for got, want in zip(dev, ora_planes):          # the extent descriptors of the plane records
        assert [p["root"] for p in got] == [p["root"] for p in want]
        for a, b in zip(got, want):
            # hull stage: the 256-direction inscribed polygon against the exact hull -- never larger, at most 0.3 % smaller, polygon mass
            # centre within 5 mm, for every region that can enter a PbMap (Frame360.h:1031,1037: area >= 0.12 m2, elongation <= 6; a
            # one-pixel-wide streak of 56 far pixels -- elongation 31 -- is all gentle arc: most of its hull vertices turn by less than
            # a direction step, and it reads 2.4 % low)
            # (a column of 43 collinear pixels -- elongation 2800 -- has no hull: hull_points 0, area = the moment rectangle's)
            assert (a["hull_points"] >= 3 or a["elongation"] > 6.0) and a["area"] <= max(b["hull_area_exact"] * (1 + 1e-5), a["area_moment"])
            if b["hull_area_exact"] > 0.12 and a["elongation"] <= 6.0:
                assert a["area"] >= 0.997 * b["hull_area_exact"], (a["area"], b["hull_area_exact"], a["elongation"])
                assert np.abs(a["center_hull"] - b["center_hull"]).max() < 5e-3
            b["area"] = b["hull_area_exact"]          # what the numpy restatement of the matcher works with below
            # (the device sums its moments in 2^-24 m fixed point: the near-zero in-plane eigenvalue of a sliver region, and with
            #  it the sliver's area, moves by ~1e-4 m2; such regions are far below min_area_plane and never matched)
            assert abs(a["area_moment"] - b["area_moment"]) <= (1e-4 * b["area_moment"] if b["area_moment"] > 0.12 else 1e-3)
            if b["area_moment"] > 0.12:              # (slivers have an ill-defined in-plane aspect)
                assert abs(a["elongation"] - b["elongation"]) <= 1e-3 * b["elongation"]
                if b["elongation"] > 1.5:
                    assert abs(abs(np.dot(a["ppal_dir"], b["ppal_dir"])) - 1) < 1e-4
    reg360 = pbmap.RegisterRGBD360(odometry_config=True)
    good = reg360.RegisterPbMap(dev[0], dev[1], 25, pbmap.ODOMETRY_6DoF)
    want = pbmap_ref.register_planes(ora_planes[0], ora_planes[1], 25, pbmap_ref.ODOMETRY_6DoF)
    assert good and want["status"] == 0
    assert reg360.getMatchedPlanes() == want["match"] and len(want["match"]) >= 5
    rot, tr = synth.pose_error(reg360.getPose(), want["pose"])
    assert rot < 1e-5 and tr < 2e-5, (rot, tr)                  # (the fit weighs by area: device hull areas are up to 0.3 % below the exact ones)
    rot, tr = synth.pose_error(reg360.getPose(), T)            # planes alone: within 0.1 degree / 1 cm of the truth
    assert rot < math.radians(0.1) and tr < 0.01, (rot, tr)
    # dense alignment seeded with the PbMap pose (target = reference frame, source = the other one)
    reg, ora, _ = _pair_ctx(hip_lib, oracle_mod, pair)
    guess = reg360.getPose()
    rc = reg.alignFrames360(guess, 2)
    st_o, pose_ref = ora.align360(guess, 2)
    assert rc == st_o == 0
    assert reg.num_iterations == list(ora.result.iters)[:3]
    rot, tr = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and tr <= POSE_TOL_DEV, (rot, tr)
    rot, tr = synth.pose_error(reg.getOptimalPose(), T)
    assert rot < 2e-3 and tr < 5e-3, (rot, tr)
    assert np.allclose(reg.getOptimalPose(), guess, atol=1e-1)  # the reference's validity test of the keyframe link


@pytest.mark.parametrize("rows,cols,sigma_s,sigma_r", [(120, 160, 10.0, 0.05), (97, 131, 7.0, 0.03), (240, 320, 10.0, 0.05)])
def test_bilateral_filter_bit_exact(hip_lib, oracle_mod, rows, cols, sigma_s, sigma_r):
    """pcl::FastBilateralFilter (Frame360.h:493-499) on the device against the oracle: same grid, same float operations, integer cell
    sums -> the filtered cloud is bit-identical; an all-invalid cloud passes through."""
    from rgbd360_amd.register import Frame360Stages
    from tests.test_oracle_cpu import _noisy_pinhole_cloud
    xyz, _ = _noisy_pinhole_cloud(rows, cols, seed=rows + cols)
    st = Frame360Stages(_mk(hip_lib, 2))
    got = st.bilateral_filter(xyz, rows, cols, sigma_s, sigma_r)
    want = oracle_mod.fast_bilateral(xyz, rows, cols, sigma_s, sigma_r)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    assert np.abs(np.nan_to_num(got[:, 2]) - np.nan_to_num(xyz.reshape(-1, 3)[:, 2])).max() > 1e-3      # it did filter
    allnan = np.full((rows, cols, 3), np.nan, np.float32)
    assert np.isnan(st.bilateral_filter(allnan, rows, cols, sigma_s, sigma_r)).all()
    # smoothed cloud -> normals / regions: the wall behind the box comes out as one region with the filter, in pieces without
    nrm = st.normals(got, rows, cols, 0.02, 8.0, 0)
    _, planes_f = st.plane_fit(got, nrm, rows, cols, 40, 0.0398, 0.02, 0.0013, 0)
    nrm0 = st.normals(xyz, rows, cols, 0.02, 8.0, 0)
    _, planes_0 = st.plane_fit(xyz, nrm0, rows, cols, 40, 0.0398, 0.02, 0.0013, 0)
    assert max([p["count"] for p in planes_f], default=0) > 2 * max([p["count"] for p in planes_0], default=1)


def test_cloud_planes_chain_equals_the_stages(hip_lib, oracle_mod):
    """rgbd360_cloud_planes (cloud uploaded once: bilateral filter -> normal map -> regions -> plane.transform(Rt)) against the
    oracle's stages composed on the host: same regions (roots, counts), plane parameters to float32 rounding in the rig frame."""
    from rgbd360_amd.register import Frame360Stages
    from tests.test_oracle_cpu import _noisy_pinhole_cloud
    rows, cols = 120, 160
    xyz, _ = _noisy_pinhole_cloud(rows, cols, seed=9, noise=0.006)
    Rt = synth.make_pose(synth.rodrigues(np.array([0.2, 1.0, -0.3]), 0.7), np.array([0.05, -0.02, 0.11]))
    st = Frame360Stages(_mk(hip_lib, 2))
    got = st.cloud_planes(xyz, rows, cols, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, 0, Rt)
    filt = oracle_mod.fast_bilateral(xyz, rows, cols, 10.0, 0.05)
    nrm, _w = oracle_mod.f360_normals(filt, rows, cols, 0.02, 8.0, 0)
    want = oracle_mod.f360_plane_segment(filt, nrm, rows, cols, 40, 0.0398, 0.02, 0.0013, 0, max_planes=512)[1]
    assert len(want) >= 2 and [p["root"] for p in got] == [p["root"] for p in want]
    assert [p["count"] for p in got] == [p["count"] for p in want]
    R, t = Rt[:3, :3], Rt[:3, 3]
    for a, b in zip(got, want):
        n, c = R @ b["normal"].astype(np.float64), R @ b["centroid"].astype(np.float64) + t
        if n @ c > 0:
            n = -n
        assert np.allclose(a["centroid"], c, atol=2e-5) and abs(a["d"] + n @ c) < 1e-4
        if b["curvature"] > 1e-9:
            assert float(a["normal"] @ n) > 1 - 1e-5
        assert abs(a["area_moment"] - b["area"]) <= 1e-3 * max(b["area"], 0.1)
    # without the filter and without Rt the call is the plain normal map + regions of the sensor cloud
    plain = st.cloud_planes(xyz, rows, cols, 0.0, 0.0, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, 0, None)
    nrm0, _w = oracle_mod.f360_normals(xyz, rows, cols, 0.02, 8.0, 0)
    want0 = oracle_mod.f360_plane_segment(xyz, nrm0, rows, cols, 40, 0.0398, 0.02, 0.0013, 0, max_planes=512)[1]
    assert [(p["root"], p["count"]) for p in plain] == [(p["root"], p["count"]) for p in want0]


@pytest.mark.parametrize("rows,cols,step", [(240, 320, 2), (240, 320, 1), (121, 163, 2), (96, 128, 4)])
def test_sensor_cloud_bit_exact(hip_lib, oracle_mod, rows, cols, step):
    """CloudRGBD::getPointCloud + DownsampleRGBD::downsamplePointCloud on the device against the oracle (bit-exact), including
    blocks without a valid depth, depths outside (min, max), odd sizes (the last row / column of an odd image is dropped like
    the reference's integer division does) and a strided depth image."""
    from rgbd360_amd.register import Frame360Stages
    rng = np.random.default_rng(rows * 7 + cols + step)
    big = rng.uniform(200, 11000, (rows, cols + 5)).astype(np.uint16)
    big[rng.random(big.shape) < 0.25] = 0
    big[8:16, 16:48] = 0                                             # whole blocks without depth (for every step)
    d = big[:, 2:2 + cols]                                           # row stride != 2 * cols
    st = Frame360Stages(_mk(hip_lib, 2))
    got = st.sensor_cloud(d, step, 0.3, 10.0)
    want = oracle_mod.sensor_cloud(np.ascontiguousarray(d), step, 0.3, 10.0)
    assert got.shape == (rows // step, cols // step, 3)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).any() and np.isfinite(want).any()
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))


@pytest.mark.parametrize("convention", [0, 1, 2])
@pytest.mark.parametrize("depth_f32", [False, True])
@pytest.mark.parametrize("depth_mode", [0, 1])
def test_frame_planes_equals_the_stages_in_every_mode(hip_lib, convention, depth_f32, depth_mode):
    """rgbd360_frame_planes forms the cloud inside the depth-edge kernel, which round 4 compiles once per {convention, depth type,
    depth mode} (k_f360_edge_bits<true, SPEC>): the cloud, the normal map and the labels of every combination equal those of the staged
    route -- rgbd360_sphere_cloud (k_sphere_cloud), rgbd360_normal_map and rgbd360_plane_fit on that cloud (k_f360_edge_bits<false>) --
    bit for bit."""
    from rgbd360_amd.register import Frame360Stages
    (_, dA), _, _ = synth.make_pair(512, 256, seed=21)
    d = dA.astype(np.float32) * np.float32(0.001) if depth_f32 else dA
    if convention == 1:
        d = d.copy()
        d[40:60, 100:140] = 16000 if not depth_f32 else 16.0      # beyond the 15 m the second convention accepts: NaN points
    reg = _mk(hip_lib, 2)
    st = Frame360Stages(reg)
    kw = dict(max_depth_change_factor=0.05, normal_smoothing_size=8.0, min_inliers=40, angular_threshold=0.03, distance_threshold=0.05,
              max_curvature=0.0013, depth_mode=depth_mode)
    one = st.frame_planes(d, convention=convention, **kw)
    xyz = reg.sphere_cloud(d, convention)
    assert np.array_equal(np.isnan(one["xyz"]), np.isnan(xyz)) and np.array_equal(np.nan_to_num(one["xyz"]), np.nan_to_num(xyz))
    nrm = st.normals(xyz, 256, 512, 0.05, 8.0, depth_mode)
    assert np.array_equal(np.isnan(one["normals"]), np.isnan(nrm)) and np.array_equal(np.nan_to_num(one["normals"]), np.nan_to_num(nrm))
    labels, planes = st.plane_fit(xyz, nrm, 256, 512, 40, 0.03, 0.05, 0.0013, depth_mode)
    assert np.array_equal(one["labels"].reshape(-1), np.asarray(labels).reshape(-1))
    assert [(p["root"], p["count"]) for p in one["planes"]] == [(p["root"], p["count"]) for p in planes]
    assert np.isfinite(nrm).any()


def test_sensor_planes_equals_the_two_calls(hip_lib):
    """rgbd360_sensor_planes (depth image in, planes out, the cloud never leaves the device) = rgbd360_sensor_cloud + rgbd360_cloud_planes."""
    from rgbd360_amd.register import Frame360Stages
    (_, dA), _, _, _ = synth.make_pinhole_pair(320, 240, seed=3)
    Rt = synth.make_pose(synth.rodrigues(np.array([1.0, 0.2, 0.1]), 0.5), np.array([0.1, 0.2, -0.05]))
    st = Frame360Stages(_mk(hip_lib, 2))
    cloud = st.sensor_cloud(dA, 2, 0.3, 10.0)
    two = st.cloud_planes(cloud, 120, 160, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, 0, Rt)
    one = st.sensor_planes(dA, 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, Rt)
    assert len(one) == len(two) >= 2
    for a, b in zip(one, two):
        assert (a["root"], a["count"]) == (b["root"], b["count"])
        assert np.array_equal(a["normal"], b["normal"]) and np.array_equal(a["centroid"], b["centroid"]) and a["d"] == b["d"]


def test_forced_iterations_in_lockstep_equal_single_pair(hip_lib):
    """rgbd360_forced_iters_batch (bench.py's iteration_lockstep block): P pairs per launch reach the single pair's pose, bit for bit."""
    (rgbA, dA), (rgbB, dB), _ = synth.make_pair(512, 256, seed=77)
    reg = _mk(hip_lib, 3)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    for method in (0, 2):
        one = reg.forced_iters(0, np.eye(4), method, 6)
        many = reg.forced_iters_batch(5, (rgbA, dA), (rgbB, dB), 0, np.eye(4), method, 6)
        assert many["status"] == one["status"]
        for k in range(5):
            assert np.array_equal(many["poses"][k], one["pose"])
    reg.close()


@pytest.mark.parametrize("occlusion", [1, 2])
def test_eval_occlusion_parity_full_size(hip_lib, oracle_mod, occlusion):
    """The occlusion-aware pass at 2048x1024, level 0: near the poles a couple of hundred source pixels collapse onto one target
    pixel -- the case the candidate-run lists of k_occ_build exist for.  Counts and numVisible exact against the sequential oracle
    at the identity, at the true pose and at a pose that compresses the image."""
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, _occluder_pair(synth.make_pair(2048, 1024, seed=77)), n_pyr=4)
    for pose in (np.eye(4), T, _occ_poses(T)[3]):
        e = reg.eval(0, pose, 2, occlusion)
        _, sp, sd, n_p, n_d = ora.error_occ(0, pose, 2, occlusion)
        H, g, Hd, gd, nvis = ora.hessgrad_occ(0, pose, 2, occlusion)
        assert list(e["n_split"]) == [n_p, n_d]
        assert e["n_visible"] == nvis
        assert abs(e["err2_split"][0] - sp) <= ERR2_RTOL * max(1.0, abs(sp))
        assert abs(e["err2_split"][1] - sd) <= ERR2_RTOL * max(1.0, abs(sd))
    reg.close()


def test_occlusion_sequence_with_the_default_inflight_equals_pairwise(hip_lib):
    """Occlusion-aware sequences take the per-context route, whose number of contexts is capped below n_inflight (a fifth busy
    hardware queue is time-sliced: DESIGN.md 3.3): with the default n_inflight every pair still gets the pose of the pairwise call."""
    frames = [synth.render(synth.trajectory_pose(k, 11), 256, 128, 11) for k in range(12)]
    reg = _mk(hip_lib, 3)
    for occ in (1, 2):
        p, s, i = reg.alignSequence(frames, method=2, occlusion=occ)
        one = _mk(hip_lib, 3)
        for j in range(len(frames) - 1):
            one.setTargetFrame(*frames[j]); one.setSourceFrame(*frames[j + 1])
            assert one.alignFrames360(np.eye(4), 2, occ) == s[j]
            assert np.array_equal(one.getOptimalPose(), p[j]) and list(one.num_iterations) == list(i[j]), (occ, j)
        one.close()
    reg.close()


_RECOMPUTE_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
out = {}
for W, H, n_pyr, depth_f32 in ((256, 128, 3, False), (200, 100, 2, False), (328, 164, 2, True), (1024, 512, 4, False)):
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=99, depth_f32=depth_f32)
    reg = RegisterPhotoICP(); reg.setNumPyr(n_pyr)
    reg.setTargetFrame(rgbA, dA); reg.setSourceFrame(rgbB, dB)
    for method in (0, 1, 2):
        for level in range(n_pyr):
            e = reg.eval(level, T, method)
            out["e_%%d_%%d_%%d_%%d" %% (W, H, method, level)] = np.concatenate([e["H64"].ravel(), e["g64"], [e["err2"], e["n_valid"], e["n_visible"]]])
        rc = reg.alignFrames360(np.eye(4), method)
        out["p_%%d_%%d_%%d" %% (W, H, method)] = np.concatenate([reg.getOptimalPose().ravel(), [rc], reg.num_iterations])
        f = reg.forced_iters(0, np.eye(4), method, 6)
        out["f_%%d_%%d_%%d" %% (W, H, method)] = f["pose"].ravel()
np.savez(sys.argv[1], **out)
"""


def test_recompute_form_of_the_pass_is_bit_identical_to_the_record_form(hip_lib, tmp_path):
    """The per-pixel pass reads its source points either as {x, y, z, I} records (16 B per pixel: the reference's LUT_xyz_sphere,
    precomputed) or re-forms them per pixel from depth + angle tables (8 B per pixel; levels of RGBD360_RECOMPUTE_MIN_PX pixels and
    more, 4 Mpx by default).  Both forms must give the same float64 sums, counts, poses and iteration counts BIT FOR BIT -- the point
    is formed by the same float operations in the same order -- at power-of-two and ragged sizes (rows of 200 / 328 pixels start
    inside a wave), u16 and float depth, every method and level, through the two-launch pass (eval), the fused launch (align, forced)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_RECOMPUTE_CHILD % dict(root=root))
    res = {}
    for tag, min_px in (("records", str(1 << 30)), ("recompute", "0")):
        path = tmp_path / (tag + ".npz")
        env = dict(os.environ, RGBD360_RECOMPUTE_MIN_PX=min_px)
        p = subprocess.run([sys.executable, str(script), str(path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["records"]) == set(res["recompute"]) and len(res["records"]) > 40
    for k in res["records"]:
        a, b = res["records"][k], res["recompute"][k]
        assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a, b.view(np.uint64) if b.dtype == np.float64 else b), k


@pytest.mark.parametrize("W,H,refine", [(512, 256, False), (512, 256, True), (1000, 500, False)])
def test_plane_colour_descriptors_match_oracle(hip_lib, oracle_mod, W, H, refine):
    """Frame360.h:1045-1046 (plane.calcPlaneHistH(); plane.calcMainColor2(); mrpt::pbmap, third-party, unpinned): normalised-colour mean
    / deviation, mean intensity and the 74-bin saturated-hue histogram of every planar region, from the frame's colour panorama.
    The per-pixel arithmetic is integer on both sides, so the device's sums equal the checker's exactly and the float descriptors agree
    to their last rounding; with segmentAndRefine's refinement on, the colour is that of the GROWN inlier sets."""
    from rgbd360_amd.register import Frame360Stages
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=31)
    rgb = rgbA.copy()
    rgb[: H // 4, :, 0] //= 3                       # tint parts of the frame so that regions differ in colour, and paint a dark
    rgb[H // 2:, :, 2] = 255 - rgb[H // 2:, :, 2]   # and a grey patch for the two special histogram bins
    rgb[H // 3: H // 3 + 20, 40:200] = (20, 25, 18)
    rgb[H // 3 + 30: H // 3 + 50, 40:200] = (128, 126, 131)
    st = Frame360Stages(_mk(hip_lib, 3))
    st.set_refinement(refine, 0.02)
    st.set_color_image(rgb)
    kw = dict(convention=2, angular_threshold=0.03 if W <= 512 else 0.02, min_inliers=40 if W <= 512 else 120)
    out = st.frame_planes(dA, **kw)
    assert len(out["planes"]) >= 5
    _, want = oracle_mod.f360_plane_colour(out["labels"], rgb, out["planes"])
    for p, w in zip(out["planes"], want):
        assert p["color_count"] == w["color_count"] > 0 and p["color_count"] <= p["count"]
        assert np.abs(p["color_nrgb"] - w["color_nrgb"]).max() <= 1e-7 and abs(p["color_nrgb"].sum() - 1.0) < 1e-3
        assert np.abs(p["color_dev"] - w["color_dev"]).max() <= 2e-6
        assert abs(p["intensity"] - w["intensity"]) <= 1e-4 * max(1.0, w["intensity"])
        assert np.abs(p["hist_h"] - w["hist_h"]).max() <= 1e-7 and abs(p["hist_h"].sum() - 1.0) < 1e-5
    assert any(p["hist_h"][72] > 0.01 for p in out["planes"]) and any(p["hist_h"][73] > 0.01 for p in out["planes"])
    # the dominant colour (calcMainColor2's mean shift over ~2000 samples of the region, integer arithmetic): sample count, the samples it
    # ends on and the mode itself are EXACT against the numpy restatement; on the painted frame it differs from the mean somewhere
    modes = oracle_mod.f360_plane_colour_mode(out["labels"], rgb, out["planes"])
    for p, w in zip(out["planes"], modes):
        assert w["color_mode_count"] >= 0 and p["color_mode_count"] == w["color_mode_count"] > 0
        assert p["color_mode_count"] <= min(p["color_count"], 4096) and p["color_mode_count"] >= min(p["color_count"], 1400)
        assert np.array_equal(p["color_mode"], w["color_mode"]) and p["intensity_mode"] == w["intensity_mode"]
        assert p["color_concentration"] == w["color_concentration"] and 0.5 <= p["color_concentration"] + 0.5 / p["color_mode_count"] and p["color_concentration"] <= 1.0
    assert any(np.abs(p["color_mode"] - p["color_nrgb"]).max() > 5e-3 for p in out["planes"])
    if refine:
        assert st.refinement_stats()["pixels_relabelled"] >= 0
    # without a colour image (or with one of another geometry) the planes come back colourless, everything else unchanged
    st.set_color_image(None)
    bare = st.frame_planes(dA, **kw)
    assert all(q["color_count"] == 0 and not q["hist_h"].any() for q in bare["planes"])
    assert [q["root"] for q in bare["planes"]] == [q["root"] for q in out["planes"]]
    st.set_color_image(rgb[: H // 2])
    assert all(q["color_count"] == 0 for q in st.frame_planes(dA, **kw)["planes"])


def test_dominant_colour_survives_a_poster_on_the_wall(hip_lib, oracle_mod):
    """calcMainColor2 (Frame360.h:1046) takes the mean-shift MODE of a plane's colour, not its mean: a wall with a poster is the same wall
    whether the poster fills 30 % of the view of it or 12 %.  Two views of the synthetic room whose far wall carries a saturated poster of
    different extent: the means of that wall's normalised colour differ by more than the matcher's colour_threshold (0.07,
    configLocaliser_spherical.ini:19), the dominant colours by far less -- and RegisterPbMap matches the wall through its colour test
    where the means alone would have refused it."""
    from rgbd360_amd import pbmap
    from rgbd360_amd.register import Frame360Stages
    W, H = 512, 256
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
    views = []
    for share in (0.30, 0.12):
        rgb = rgbA.copy()
        st = Frame360Stages(_mk(hip_lib, 3))
        base = st.frame_planes(dA, convention=2, angular_threshold=0.03, min_inliers=40)
        big = max(base["planes"], key=lambda p: p["count"])                       # the largest region: paint a poster over `share` of its pixels
        rr, cc = np.nonzero(base["labels"] == big["root"])
        order = np.lexsort((rr, cc))                                               # column-major: a compact block of the region
        k = int(share * len(order))
        rgb[rr[order[:k]], cc[order[:k]]] = (250, 30, 20)
        st.set_color_image(rgb)
        out = st.frame_planes(dA, convention=2, angular_threshold=0.03, min_inliers=40)
        views.append((out, [p for p in out["planes"] if p["root"] == big["root"]][0], rgb))
    (outA, wallA, rgbA2), (outB, wallB, rgbB2) = views
    assert np.abs(wallA["color_nrgb"] - wallB["color_nrgb"]).max() > 0.07                 # the means: another colour
    assert np.abs(wallA["color_mode"] - wallB["color_mode"]).max() < 0.02                 # the dominant colours: the same wall
    assert 0.5 <= wallA["color_concentration"] <= 0.8 and wallB["color_concentration"] > wallA["color_concentration"]
    for out, rgb in ((outA, rgbA2), (outB, rgbB2)):                                         # exact against the restatement
        for p, w in zip(out["planes"], oracle_mod.f360_plane_colour_mode(out["labels"], rgb, out["planes"])):
            assert p["color_mode_count"] == w["color_mode_count"] and np.array_equal(p["color_mode"], w["color_mode"])
    res = pbmap.register_planes(outA["planes"], outB["planes"], 0, 0)
    ia = [i for i, p in enumerate(outA["planes"]) if p["root"] == wallA["root"]][0]
    ib = [i for i, p in enumerate(outB["planes"]) if p["root"] == wallB["root"]][0]
    assert res["status"] == 0 and res["match"].get(ia) == ib
    # the same records stripped of their dominant colour fall back on the means -- and the wall is refused
    strip = lambda planes: [dict(p, color_mode_count=0) for p in planes]
    res2 = pbmap.register_planes(strip(outA["planes"]), strip(outB["planes"]), 0, 0)
    assert res2["match"].get(ia) != ib


def test_plane_colour_of_a_downsampled_sensor_cloud(hip_lib, oracle_mod):
    """The per-sensor route (Frame360.h:479-503: CloudRGBD::getPointCloud + DownsampleRGBD, step 2): a cloud pixel takes the colour of the
    CENTRE pixel of its 2 x 2 block (DownsampleRGBD.h:240, 285-287)."""
    from rgbd360_amd.register import Frame360Stages
    rows, cols, step = 240, 320, 2
    (rgb, depth), _, _, _ = synth.make_pinhole_pair(cols, rows, seed=3)       # a pinhole view of the synthetic room
    rgb = rgb.copy()
    rgb[:, : cols // 2, 0] = 230                                  # the left half of the view is reddish
    st = Frame360Stages(_mk(hip_lib, 2))
    st.set_color_image(rgb, step=step)
    r2, c2 = rows // step, cols // step
    xyz = st.sensor_cloud(depth, step, 0.3, 10.0)
    nrm = st.normals(xyz, r2, c2, 0.02, 8.0, 0)
    labels, planes = st.plane_fit(xyz, nrm, r2, c2, 40, 0.0398, 0.02, 0.0013, 0)
    assert len(planes) >= 2
    _, want = oracle_mod.f360_plane_colour(labels, rgb, planes, step=step)
    for p, w in zip(planes, want):
        assert p["color_count"] == w["color_count"] > 0
        assert np.abs(p["color_nrgb"] - w["color_nrgb"]).max() <= 1e-7 and np.abs(p["hist_h"] - w["hist_h"]).max() <= 1e-7
    # the centre-of-block rule: a checker that sampled the block's first pixel instead would disagree
    _, other = oracle_mod.f360_plane_colour(labels, np.roll(rgb, (1, 1), axis=(0, 1)), planes, step=step)
    assert any(np.abs(p["color_nrgb"] - w["color_nrgb"]).max() > 1e-5 for p, w in zip(planes, other))
    # the one-call route (no smoothing, same thresholds) reports the same colour records
    one = st.sensor_planes(depth, step, 0.3, 10.0, 0.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013)
    assert [q["color_count"] for q in one] == [q["color_count"] for q in planes]


def test_colour_constraint_on_device_planes(hip_lib):
    """The chain the reference's keyframe link runs (Frame360 planes with their colour -> RegisterPbMap): two views of the synthetic room
    with their own colour panoramas match as they do without colour (same albedo seen from two poses: the descriptors agree well inside
    the .ini thresholds); the SAME geometry with the second view's colour channels rotated (every wall another colour) matches nothing
    -- walls of equal shape and different colour are not paired -- while a 15 % darker second view still matches everything."""
    from rgbd360_amd import pbmap
    from rgbd360_amd.register import Frame360Stages
    W, H = 512, 256
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=5, trans=0.06, rot_deg=2.0)
    rgbA, rgbB = rgbA.copy(), rgbB.copy()
    for rgb in (rgbA, rgbB):                 # the generator's walls are grey-ish: tint them (a colour per octant of the panorama)
        for k in range(8):
            rgb[:, k * W // 8:(k + 1) * W // 8, k % 3] = np.minimum(255, rgb[:, k * W // 8:(k + 1) * W // 8, k % 3].astype(int) + 90).astype(np.uint8)
    st = Frame360Stages(_mk(hip_lib, 3))

    def planes(depth, rgb):
        st.set_color_image(rgb)
        return st.frame_planes(depth, convention=2, angular_threshold=0.03)["planes"]
    pa, pb = planes(dA, rgbA), planes(dB, rgbB)
    assert all(p["color_count"] > 0 for p in pa + pb)
    par = pbmap.default_params(True)
    with_colour = pbmap.register_planes(pa, pb, 0, pbmap.ODOMETRY_6DoF, par)
    par.use_color = 0
    without = pbmap.register_planes(pa, pb, 0, pbmap.ODOMETRY_6DoF, par)
    assert with_colour["status"] == without["status"] == 0 and with_colour["match"] == without["match"] and len(without["match"]) >= 4
    par.use_color = 1
    rot, tr = synth.pose_error(with_colour["pose"], T)
    assert rot < 2e-3 and tr < 5e-3, (rot, tr)
    repainted = planes(dB, np.ascontiguousarray(rgbB[:, :, [1, 2, 0]]))
    r = pbmap.register_planes(pa, repainted, 0, pbmap.ODOMETRY_6DoF, par)
    assert len(r["match"]) < len(without["match"]) and r["status"] != 0, r["match"]
    # (15 %: the dominant colour's intensity -- the mean R + G + B of the samples the mode ends on, ~600 on the tinted walls -- may move by
    # the odometry profile's intensity_threshold of 100 at most)
    darker = planes(dB, (rgbB * 0.85).astype(np.uint8))
    assert pbmap.register_planes(pa, darker, 0, pbmap.ODOMETRY_6DoF, par)["match"] == without["match"]


# ---- what the reference's float32 branch lets through: non-finite and out-of-range depth (RPI.h:316-319: level 0 is the caller's image
#      AS IS; the LUT's range gate :4573 drops such SOURCE pixels; a TARGET pixel reaches isfinite(depth2), :2714 / :3064) -------------------
def _spoiled_pair(W, H, seed, ramps, isolated=True):
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=seed, depth_f32=True)
    return (rgbA, synth.spoil_depth(dA, 1, ramps, isolated)), (rgbB, synth.spoil_depth(dB, 2, ramps, isolated)), T


@pytest.mark.parametrize("ramps", [False, True])
def test_nonfinite_float_depth_planes_and_counts(hip_lib, oracle_mod, ramps):
    """A 256 x 128 float32-metres pair whose depth images carry NaN, +Inf, -Inf, negative values, values beyond maxDepth and zeros (patches
    and isolated pixels) in BOTH frames.  Level 0 keeps them (bit for bit, NaN payloads included), the higher levels average only
    the valid ones; gradients next to them follow the comparisons of RPI.h:365-398 (NaN: never monotone; Inf: a finite one-sided
    difference); the LUT's gate invalidates such source pixels; the pass counts exactly the pixels the oracle counts, the target's NaN
    and Inf pixels failing isfinite(depth2)."""
    pair = _spoiled_pair(256, 128, 77, ramps)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair)
    assert not np.isfinite(pair[0][1]).all() and (pair[0][1] < 0).any() and (pair[0][1] > 6.0).any()
    for level in range(3):
        ora.prepare_level(level)
        for name in ("depth_src", "depth_trg", "dgx", "dgy", "gx", "gy"):
            a, b = reg.plane(name, level), ora.plane(name, level)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (name, level)
            assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), (name, level)
        if level > 0:
            assert np.isfinite(reg.plane("depth_trg", level)).all()              # means of valid pixels only (RPI.h:337-346)
        la, lb = reg.lut(level), ora.lut(level)
        valid = lb[:, 0] != -10000
        assert np.array_equal(la[:, 0] != -10000, valid) and np.isfinite(lb[valid]).all()
        assert np.array_equal(la[valid].view(np.uint32), lb[valid].view(np.uint32))
    assert (~np.isfinite(reg.plane("depth_trg", 0))).sum() > 100
    for level in range(3):
        for pose in _poses(T)[:3]:
            for method in (0, 1, 2):
                e = reg.eval(level, pose, method)
                rms, err2, nvalid = ora.error(level, pose, method)
                H, g, Hd, gd, nvis = ora.hessgrad(level, pose, method)
                assert e["n_valid"] == nvalid and e["n_visible"] == nvis, (level, method)
                if np.isfinite(err2):
                    assert abs(e["err2"] - err2) <= ERR2_RTOL * max(1.0, abs(err2))
                    assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
                else:                       # a salient gradient on non-positive target depth: the reference's residual is NaN (RPI.h:2721-2723)
                    assert ramps and method != 0 and level == 0 and not np.isfinite(e["err2"])
            for occ in (1, 2):
                e = reg.eval(level, pose, 2, occ)
                _r, sp, sd, n_p, n_d = ora.error_occ(level, pose, 2, occ)
                assert list(e["n_split"]) == [n_p, n_d], (level, occ, list(e["n_split"]), n_p, n_d)
                for got, want in zip(e["err2_split"], (sp, sd)):
                    assert (abs(got - want) <= ERR2_RTOL * max(1.0, abs(want))) if np.isfinite(want) else not np.isfinite(got), (level, occ, got, want)


@pytest.mark.parametrize("ramps", [False, True])
@pytest.mark.parametrize("method,occlusion", [(0, 0), (1, 0), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2)])
def test_nonfinite_float_depth_alignment(hip_lib, oracle_mod, method, occlusion, ramps):
    """The same pair through rgbd360_align360: status, iterations per level and pose as the oracle's.  With the ramps the sums of the
    depth modalities are NaN at level 0: `diff_error > tol_residual` is false (RPI.h:4611), the level ends without a step and the call
    reports what the reference's NaN error amounts to -- handled like the oracle, not as a silent number."""
    pair = _spoiled_pair(256, 128, 77, ramps)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair)
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    st, pose_ref = ora.align360(np.eye(4), method, occlusion)
    assert rc == st, (rc, st)
    assert reg.num_iterations == list(ora.result.iters)[:3], (reg.num_iterations, list(ora.result.iters)[:3])
    pose = reg.getOptimalPose()
    assert np.isfinite(pose).all()
    rot, trans = synth.pose_error(pose, pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    _assert_libm_oracle_agrees(reg, ora, method, 3, occlusion, status=st)
    if st == 0 and method != 1:
        rot_gt, trans_gt = synth.pose_error(pose, T)
        assert rot_gt < 3e-3 and trans_gt < 6e-3, (rot_gt, trans_gt)              # the spoiled tenth of the image does not derail it


def test_nonfinite_float_depth_full_size_sequence_and_planes(hip_lib, oracle_mod):
    """2048 x 1024 with the same kinds of values: alignment vs the oracle (iterations, exact counts, pose), the sequence engine on
    [A, B, A] bit-identical to the pairwise calls, and rgbd360_frame_planes on the spoiled target: cloud, normal map and region labels
    as the oracle's stages give them (NaN / Inf ranges are points the normal map's finite test drops)."""
    # with isolated pixels of every kind (7 x 1024 of them) a few fall side by side and form ramps: the depth sums of level 0 are NaN,
    # the level ends without a step (status 2, iterations [0, 1, 1, 5]) -- the device follows the oracle there too
    noisy = _spoiled_pair(2048, 1024, 1234, False)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, noisy, n_pyr=4)
    for method, want in ((0, 0), (2, 2)):
        rc = reg.alignFrames360(np.eye(4), method)
        st, pose_ref = ora.align360(np.eye(4), method)
        assert rc == st == want and reg.num_iterations == list(ora.result.iters)[:4], (method, rc, st)
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
        assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    assert reg.num_iterations[0] == 0
    pair = _spoiled_pair(2048, 1024, 1234, False, isolated=False)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=4)
    for method in (0, 2):
        rc = reg.alignFrames360(np.eye(4), method)
        st, pose_ref = ora.align360(np.eye(4), method)
        assert rc == st == 0 and reg.num_iterations == list(ora.result.iters)[:4]
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
        assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
        e = reg.eval(0, pose_ref, method)
        _, err2, nvalid = ora.error(0, pose_ref, method)
        _H, _g, Hd, gd, nvis = ora.hessgrad(0, pose_ref, method)
        assert e["n_valid"] == nvalid and e["n_visible"] == nvis
        assert abs(e["err2"] - err2) <= ERR2_RTOL * err2 and np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()
        _assert_libm_oracle_agrees(reg, ora, method, 4)
    pose_pair = reg.getOptimalPose()
    frames = [pair[0], pair[1], pair[0]]
    seq = _mk(hip_lib, 4)
    poses, status, iters = seq.alignSequence(frames, 2)
    assert status[0] == 0 and np.array_equal(poses[0], pose_pair) and list(iters[0]) == reg.num_iterations
    back = _mk(hip_lib, 4)
    back.setTargetFrame(*pair[1]); back.setSourceFrame(*pair[0])
    assert back.alignFrames360(np.eye(4), 2) == status[1] == 0 and np.array_equal(back.getOptimalPose(), poses[1])
    # Frame360 stages on the spoiled range image
    from rgbd360_amd.register import Frame360Stages
    dA = pair[0][1]
    out = Frame360Stages(reg).frame_planes(dA, convention=2, angular_threshold=0.03)
    xyz = oracle_mod.sphere_cloud(dA, 2)
    assert np.array_equal(np.isnan(xyz), np.isnan(out["xyz"])) and np.array_equal(np.nan_to_num(xyz, posinf=1e30, neginf=-1e30), np.nan_to_num(out["xyz"], posinf=1e30, neginf=-1e30))
    nrm, _ = oracle_mod.f360_normals(xyz, 1024, 2048, 0.05, 8.0, 1)
    ok = ~np.isnan(nrm[:, 0])
    assert np.array_equal(np.isnan(out["normals"][:, 0]), ~ok)
    assert np.abs(out["normals"][ok] - nrm[ok]).max() <= 1.2e-7
    labels, planes = oracle_mod.f360_plane_segment(xyz, out["normals"], 1024, 2048, 40, 0.03, 0.05, 0.001, 1)
    assert np.array_equal(np.asarray(out["labels"]).reshape(-1), np.asarray(labels).reshape(-1))
    assert [(p["root"], p["count"]) for p in out["planes"]] == [(p["root"], p["count"]) for p in planes] and len(planes) >= 6


def test_hull_polygon_keeps_its_sense_when_the_rig_origin_lies_behind_the_plane(hip_lib):
    """rgbd360_sensor_planes with an Rt that puts the rig's origin on the far side of the walls the sensor sees (t = 6 m along the optical
    axis): every normal is turned round to face the new origin (Frame360.h:989-993) and the record's polygon must still run
    counter-clockwise seen from the side the normal points to (include/rgbd360_hip.h) -- the vertex order is reversed with the normal."""
    from rgbd360_amd.register import Frame360Stages
    (_, dA), _, _, _ = synth.make_pinhole_pair(320, 240, seed=3)
    st = Frame360Stages(_mk(hip_lib, 2))
    near = st.sensor_planes(dA, 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, np.eye(4))
    far = st.sensor_planes(dA, 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, synth.make_pose(np.eye(3), np.array([0.0, 0.0, -6.0])))
    assert len(near) == len(far) >= 2
    flipped = checked = 0
    for a, b in zip(near, far):
        assert (a["root"], a["count"]) == (b["root"], b["count"]) and b["d"] >= 0
        if a["hull_points"] >= 3 and a["area"] > 0.05:      # (a sliver of a few cm2 at 3 m: its area is float32 rounding)
            _check_hull_polygon(a)
            _check_hull_polygon(b)
            checked += 1
        flipped += int(float(a["normal"] @ b["normal"]) < -0.99)
    assert flipped >= 1 and checked >= 2


@pytest.mark.parametrize("method", [0, 2])
def test_two_launch_schedule_equals_the_fused_one_on_alignments(hip_lib, small_pair, method):
    """rgbd360_debug_set_schedule(ctx, 0, 1): every Gauss-Newton iteration as a {k_eval, k_solve} pair instead of ONE k_eval_fs launch
    (the library's default, the kernel `value` is made of): poses, iteration counts, Hessians and residuals bit-identical, from the
    identity and from a guess (until round 6 an environment variable, RGBD360_FUSED_SOLVE, switched this; tests/tools/fused_soak.py is
    the long version)."""
    (rgbA, dA), (rgbB, dB), T = small_pair
    rows = {}
    for fused in (True, False):
        reg = _mk(hip_lib, 3)
        reg.debug_set_schedule(fused_solve=fused, fused_occ=True)
        reg.setTargetFrame(rgbA, dA)
        reg.setSourceFrame(rgbB, dB)
        out = []
        for g in (np.eye(4), np.asarray(T), np.eye(4)):
            rc = reg.alignFrames360(g, method)
            out.append((rc, list(reg.num_iterations), reg.getOptimalPose().copy(), reg.getHessian().copy(), reg.avResidual))
        rows[fused] = out
    for a, b in zip(rows[True], rows[False]):
        assert a[0] == b[0] == 0 and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4], (a, b)


def test_per_context_sequence_route_equals_the_lockstep_engines(hip_lib):
    """rgbd360_debug_set_sequence_route(ctx, 1, 3): a plain sequence over the per-context route (one context per sub-chunk of pairs, what
    the occlusion-aware sequences always use) gives the lock-step engines' poses bit for bit (tests/tools/engine_soak.py: the long version)."""
    frames = [synth.render(synth.trajectory_pose(k, 7), 256, 128, 7) for k in range(6)]
    reg = _mk(hip_lib, 3)
    p1, s1, i1 = reg.alignSequence(frames, method=2, n_inflight=4)
    reg.debug_set_sequence_route(True, 3)
    p0, s0, i0 = reg.alignSequence(frames, method=2, n_inflight=3)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1) and np.array_equal(i0, i1) and not s1.any()


# ---- the spherical warp in the reference's own arithmetic (rgbd360_set_index_arithmetic(ctx, 1), csrc/libm_f32.h) ----------------------
def test_libm_restatement_on_the_device_equals_the_c_library(hip_lib):
    """asinf / atanf / roundf / atan2f as csrc/libm_f32.h restates them (glibc 2.35's fdlibm float code, operation for operation), evaluated
    by the DEVICE, against this process's C library: every float of four 2^24-wide windows (around 0.5 and 1 where asinf changes its branch,
    the arctangent's reduction ranges, the index range of roundf) and 2^26 drawn pairs.  (The host compile of the same header is checked
    against the library on every float by tools/libm_f32_check.cpp.)"""
    reg = _mk(hip_lib, 2)
    for first in (0x3e800000, 0x3f800000 - (1 << 23), 0x40000000, 0xbf000000, 0x44000000, 0x32000000 - (1 << 23)):
        bad = reg.selftest_libm(first, 1 << 24)
        assert bad == (0, 0, 0, 0), (hex(first), bad)


@pytest.mark.parametrize("W,H", [(256, 128), (2048, 1024), (4096, 2048)])
def test_warp_indices_in_the_reference_arithmetic_are_bit_exact(hip_lib, oracle_mod, W, H):
    """rgbd360_set_index_arithmetic(ctx, 1): every target index and the visibility of every source pixel equal the oracle's math_mode 0
    (the C library's asinf / atan2f / roundf on the CPU) -- at the identity, the true motion and perturbed poses, on every level.  In the
    default arithmetic 8e-5 of them differ by one pixel (test_warp_indices_against_the_reference_arithmetic_full_size)."""
    pair = synth.make_pair(W, H, seed=11)
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, pair, n_pyr=3, math_mode=0)
    reg.set_index_arithmetic(1)
    flips_default = 0
    for level in range(3):
        for P in _poses(T):
            got = reg.warp_indices(level, P)
            want = ora.warp_indices(level, P)
            assert np.array_equal(got, want), (level, int((got != want).any(axis=1).sum()))
    reg.set_index_arithmetic(0)
    got = reg.warp_indices(0, T)
    flips_default = int((got != ora.warp_indices(0, T)).any(axis=1).sum())
    if W >= 2048:
        assert 0 < flips_default < 2e-4 * W * H          # what the default definition leaves


@pytest.mark.parametrize("method", [0, 1, 2])
def test_eval_counts_in_the_reference_arithmetic_are_exact(hip_lib, oracle_mod, small_pair, method):
    """With the reference's warp arithmetic the pass's pixel counts equal the libm oracle's exactly, its sums to float rounding."""
    reg, ora, T = _pair_ctx(hip_lib, oracle_mod, small_pair, math_mode=0)
    reg.set_index_arithmetic(1)
    for level in range(3):
        for P in _poses(T):
            e = reg.eval(level, P, method)
            rms, err2, nvalid = ora.error(level, P, method)
            H, g, Hd, gd, nvis = ora.hessgrad(level, P, method)
            assert e["n_valid"] == nvalid and e["n_visible"] == nvis, (level, e["n_valid"], nvalid, e["n_visible"], nvis)
            assert abs(e["err2"] - err2) <= ERR2_RTOL * max(1.0, abs(err2)), (e["err2"], err2)
            assert np.abs(e["H64"] - Hd).max() <= HG_RTOL * np.abs(Hd).max()


@pytest.mark.parametrize("method,occlusion", [(0, 0), (2, 0), (2, 1), (1, 2)])
def test_alignment_in_the_reference_arithmetic_follows_the_libm_oracle(hip_lib, oracle_mod, small_pair, method, occlusion):
    """The whole alignment with the reference's warp arithmetic against the libm oracle with float64 sums (math_mode 0, reduce_mode 1): same
    status, same accept / reject sequence, pose within the device tolerance -- the comparison that costs 1e-4 rad in the default arithmetic
    wherever an index flips."""
    pair = synth.add_occluder(small_pair) if occlusion else small_pair
    (rgbA, dA), (rgbB, dB), T = pair
    reg = _mk(hip_lib, 3)
    reg.set_index_arithmetic(1)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=0, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    st, pose_ref = ora.align360(np.eye(4), method, occlusion)
    assert rc == st == 0
    assert reg.num_iterations == list(ora.result.iters)[:3]
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)
    # the sequence entry follows the context's setting
    poses, status, iters = reg.alignSequence([(rgbA, dA), (rgbB, dB)], method=method, occlusion=occlusion)
    assert status[0] == 0 and list(iters[0]) == reg.num_iterations
    rot, trans = synth.pose_error(poses[0], pose_ref)
    assert rot <= POSE_TOL_DEV and trans <= POSE_TOL_DEV, (rot, trans)


@pytest.mark.parametrize("depth_f32", [False, True])
def test_pinhole_warp_and_alignment_in_the_reference_arithmetic(hip_lib, oracle_mod, depth_f32):
    """rgbd360_set_index_arithmetic(ctx, 1) on the pinhole path (RPI.h:701-708 as compiled: no fused multiply-adds, 1.0 / Z in double,
    roundf): target indices bit-equal to the oracle's math_mode 0 on every level, the Levenberg-Marquardt alignment on the libm oracle's
    accept / reject sequence (float64 sums), plain and occlusion-aware."""
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77, depth_f32=depth_f32)
    reg = _mk(hip_lib, 3, setMaskSeams=False)
    reg.set_index_arithmetic(1)
    reg.setCameraMatrix(K)
    reg.setTargetFrame(rgbA, dA)
    reg.setSourceFrame(rgbB, dB)
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=0, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    for occlusion in (0, 1, 2):
        rc = reg.alignFrames(np.eye(4), 2, occlusion)           # (builds the per-alignment source records the index kernel reads)
        st, pose_ref = ora.align_pinhole(np.eye(4), 2, occlusion)
        assert rc == st and (occlusion != 0 or rc == 0), (occlusion, rc, st)      # (the occlusion-aware runs of this pair may end without residuals: both sides alike)
        assert reg.num_iterations == list(ora.result.iters)[:3], (occlusion, reg.num_iterations, list(ora.result.iters)[:3])
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
        assert rot <= PINHOLE_ROT_TOL_DEV and trans <= PINHOLE_TRANS_TOL_DEV, (occlusion, rot, trans)
    for level in range(3):
        for P in _poses(T):
            assert np.array_equal(reg.warp_indices_pinhole(level, P), ora.warp_indices_pinhole(level, P)), level


def test_reference_arithmetic_reaches_sibling_contexts_engines_and_the_multi_handle(hip_lib, oracle_mod):
    """rgbd360_set_index_arithmetic is a property of the context every route of it inherits: the sibling contexts of the per-context sequence
    route (occlusion-aware sequences), the lock-step engines, and the devices of a multi-GPU handle -- a 5-frame sequence gives, pair by
    pair, the poses of single alignments in the same arithmetic (bit for bit), and these differ from the default arithmetic's."""
    from rgbd360_amd.multi import MultiGpuSequence
    frames = [synth.render(synth.trajectory_pose(k, 3), 256, 128, 3) for k in range(5)]
    reg = _mk(hip_lib, 3)
    out = {}
    for mode in (0, 1):
        reg.set_index_arithmetic(mode)
        single = []
        for k in range(4):
            reg.setTargetFrame(*frames[k])
            reg.setSourceFrame(*frames[k + 1])
            assert reg.alignFrames360(np.eye(4), 2, 2) == 0
            single.append(reg.getOptimalPose().copy())
        poses_occ, st_occ, _ = reg.alignSequence(frames, method=2, occlusion=2, n_inflight=4)        # per-context route: three siblings
        poses_eng, st_eng, _ = reg.alignSequence(frames, method=2, occlusion=0, n_inflight=4)        # lock-step engines
        assert list(st_occ) == [0] * 4 and list(st_eng) == [0] * 4
        for k in range(4):
            assert np.array_equal(poses_occ[k], single[k]), (mode, k)
        plain = []
        for k in range(4):
            reg.setTargetFrame(*frames[k])
            reg.setSourceFrame(*frames[k + 1])
            reg.alignFrames360(np.eye(4), 2, 0)
            plain.append(reg.getOptimalPose().copy())
            assert np.array_equal(poses_eng[k], plain[k]), (mode, k)
        out[mode] = (single, plain)
    assert any(not np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1]))      # the two arithmetics are not the same function
    m = MultiGpuSequence(n_gpus=1, n_pyr=3)
    try:
        m.set_index_arithmetic(1)
        poses_m, st_m, _ = m.align_sequence(frames, method=2)
        for k in range(4):
            assert np.array_equal(poses_m[k], out[1][1][k]), k
    finally:
        m.close()
