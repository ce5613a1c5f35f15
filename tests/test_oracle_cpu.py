"""CPU tests of the oracle (no GPU): golden fixtures, an independent numpy restatement, analytic known answers."""
import json
import os
import sys
import zlib

import numpy as np
import pytest

from rgbd360_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(HERE, "golden", "pair_256x128.npz"))
    j = json.load(open(os.path.join(HERE, "golden", "oracle_256x128.json")))
    return z, j


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def test_generator_reproduces_golden_inputs(golden, small_pair):
    z, _ = golden
    (rgbA, dA), (rgbB, dB), T = small_pair
    assert np.array_equal(rgbA, z["rgbA"]) and np.array_equal(dA, z["dA"])
    assert np.array_equal(rgbB, z["rgbB"]) and np.array_equal(dB, z["dB"])
    assert np.allclose(T, z["T_gt"], atol=1e-15)


def test_oracle_planes_match_golden(golden, oracle_mod):
    z, j = golden
    ora = oracle_mod.Oracle(n_pyr=3)
    ora.set_target(z["rgbA"], z["dA"])
    ora.set_source(z["rgbB"], z["dB"])
    for level in range(3):
        ora.prepare_level(level)
        for name in oracle_mod.PLANES:
            p = ora.plane(name, level)
            assert crc(p) == j["planes"]["%s/%d" % (name, level)]["crc32"], (name, level)
        lut = ora.lut(level)
        assert crc(lut[lut[:, 0] != -10000]) == j["lut"][str(level)]["crc32_valid_xyz"]


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("method", [0, 1, 2])
def test_oracle_alignment_matches_golden(golden, oracle_mod, math_mode, method):
    z, j = golden
    ref = j["runs"]["math%d/method%d" % (math_mode, method)]
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1)
    ora.set_target(z["rgbA"], z["dA"])
    ora.set_source(z["rgbB"], z["dB"])
    st, pose = ora.align360(np.eye(4), method)
    assert st == ref["status"]
    assert list(ora.result.iters)[:3] == ref["iters"]
    rot, trans = synth.pose_error(pose, np.array(ref["pose"]))
    assert rot < 1e-6 and trans < 1e-6
    tr = ora.trace()
    assert len(tr) == len(ref["trace"])
    for a, b in zip(tr, ref["trace"]):
        assert (a["level"], a["it"], a["accepted"], a["n_valid"]) == (b["level"], b["it"], b["accepted"], b["n_valid"])
        assert abs(a["new_error"] - b["new_error"]) < 1e-7
    e = ora.error(1, z["T_gt"], method)
    g = ref["at_gt_level1"]
    assert e[2] == g["n_valid"] and abs(e[1] - g["err2"]) < 1e-9 * g["err2"]
    H, gg, Hd, gd, nvis = ora.hessgrad(1, z["T_gt"], method)
    assert nvis == g["n_visible"]
    assert np.allclose(Hd, np.array(g["H64"]), rtol=1e-9)


def test_alignment_recovers_ground_truth(oracle_mod, small_pair):
    """Known answer: the recovered pose approaches the synthetic ground truth (resolution-limited), for every cost."""
    (rgbA, dA), (rgbB, dB), T = small_pair
    for method in (0, 1, 2):
        ora = oracle_mod.Oracle(n_pyr=3)
        ora.set_target(rgbA, dA)
        ora.set_source(rgbB, dB)
        st, pose = ora.align360(np.eye(4), method)
        assert st == 0
        rot0, trans0 = synth.pose_error(np.eye(4), T)
        rot, trans = synth.pose_error(pose, T)
        assert rot < 0.05 * rot0 and trans < 0.08 * trans0, (method, rot, trans)


def test_identical_frames_have_zero_gradient(oracle_mod, small_pair):
    (rgbA, dA), _, _ = small_pair
    ora = oracle_mod.Oracle(n_pyr=3, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbA, dA)
    H, g, Hd, gd, nvis = ora.hessgrad(0, np.eye(4), 2)
    # identity warp of identical frames: photo residuals vanish exactly; depth residuals only carry the
    # float32 rounding of |p| vs the stored depth
    assert np.abs(gd).max() < 1e-3 * np.sqrt(np.abs(Hd).max())
    rms, err2, nvalid = ora.error(0, np.eye(4), 0)
    assert err2 == 0.0 and nvalid > 0


def test_jacobian_sign_canary(oracle_mod, small_pair):
    """A Gauss-Newton step from the identity must reduce the error (a sign slip in J_T flips the update)."""
    (rgbA, dA), (rgbB, dB), T = small_pair
    ora = oracle_mod.Oracle(n_pyr=3, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    for method in (0, 2):
        e0 = ora.error(2, np.eye(4), method)[0]
        H, g, _, _, _ = ora.hessgrad(2, np.eye(4), method)
        st, pose1, upd = oracle_mod.gn_step(H, g, 1.0, np.eye(4))
        assert st == 0
        e1 = ora.error(2, pose1, method)[0]
        assert e1 < e0
        assert synth.pose_error(pose1, T)[0] < synth.pose_error(np.eye(4), T)[0]


@pytest.mark.parametrize("method", [0, 1, 2])
def test_oracle_agrees_with_independent_numpy_restatement(oracle_mod, method):
    """Two separately written restatements of the reference (C++ per-pixel loops vs vectorised numpy) must agree.
    numpy's float32 sin/asin/atan2 differ from libm by an ulp now and then, hence tolerances, not equality."""
    import np_restatement as NP
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(128, 64, seed=99)
    ora = oracle_mod.Oracle(n_pyr=2, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    fr = NP.Frames(rgbA, dA, rgbB, dB, n_pyr=2)
    for level in range(2):
        ora.prepare_level(level)
        for name, mine in (("gray_trg", fr.gray_t), ("gray_src", fr.gray_s), ("depth_trg", fr.dep_t),
                           ("depth_src", fr.dep_s), ("gx", fr.gx), ("gy", fr.gy), ("dgx", fr.dgx), ("dgy", fr.dgy)):
            a, b = ora.plane(name, level), mine[level]
            assert np.array_equal(a, b), (name, level, np.abs(a - b).max())
        x, y, z, valid = fr.lut(level)
        lut = ora.lut(level)
        assert np.array_equal(lut[:, 0] != -10000, valid)
        assert np.allclose(lut[valid], np.stack([x, y, z], 1)[valid], rtol=3e-7, atol=1e-7)
        for pose in (np.eye(4), T):
            rms, err2, nvalid = ora.error(level, pose, method)
            H, g, Hd, gd, nvis = ora.hessgrad(level, pose, method)
            e2, nv, Hn, gn, nvisn = NP.error_and_hg(fr, level, pose, method)
            assert abs(nv - nvalid) <= max(3, 2e-3 * nvalid)
            assert abs(nvisn - nvis) <= 3
            assert abs(e2 - err2) <= 5e-3 * err2
            assert np.abs(Hn - Hd).max() <= 5e-3 * np.abs(Hd).max()
            assert np.abs(gn - gd).max() <= 5e-3 * max(np.abs(gd).max(), 1e-2 * np.sqrt(np.abs(Hd).max()))


def test_device_polynomials_are_within_three_ulp_of_exact(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-1, 1, 20000), [0.0, 1.0, -1.0, 0.5, -0.5, 0.49999997, 0.99999994]]).astype(np.float32)
    got = np.array([L.oracle_asinf_poly(float(x)) for x in xs], np.float32)
    ref = np.arcsin(xs.astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - ref) / np.maximum(ulp, 1e-45)) <= 3.0
    ys = rng.normal(size=20000).astype(np.float32)
    zs = rng.normal(size=20000).astype(np.float32)
    got = np.array([L.oracle_atan2f_poly(float(y), float(z)) for y, z in zip(ys, zs)], np.float32)
    ref = np.arctan2(ys.astype(np.float64), zs.astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - ref) / ulp) <= 3.0
    # signed-zero column seam: atan2(-0, -z) = -pi, atan2(+0, -z) = +pi (SURVEY.md 3.4 gotcha 3)
    assert L.oracle_atan2f_poly(-0.0, -1.0) == pytest.approx(-np.pi) and L.oracle_atan2f_poly(0.0, -1.0) == pytest.approx(np.pi)


def test_libm_and_device_arithmetic_modes_agree_on_pose(oracle_mod, small_pair):
    (rgbA, dA), (rgbB, dB), T = small_pair
    poses = []
    for mm in (0, 1):
        ora = oracle_mod.Oracle(n_pyr=3, math_mode=mm, reduce_mode=1)
        ora.set_target(rgbA, dA)
        ora.set_source(rgbB, dB)
        st, pose = ora.align360(np.eye(4), 2)
        assert st == 0
        poses.append(pose)
        if mm == 0:
            idx0 = ora.warp_indices(0, T)
        else:
            idx1 = ora.warp_indices(0, T)
    rot, trans = synth.pose_error(poses[0], poses[1])
    assert rot < 1e-5 and trans < 1e-5
    # the two arithmetic definitions disagree on the nearest pixel only where the warp lands within an ulp of a
    # half-integer: a vanishing fraction of pixels
    assert (idx0 != idx1).any(axis=1).mean() < 2e-3


def test_huber_rank_inverse_exp_against_numpy(oracle_mod):
    import ctypes as C
    L = oracle_mod.lib()
    k = np.float32(6 / 255)
    for e in (0.0, 0.01, -0.01, 0.0235, 0.05, -0.3):
        w = L.oracle_weight_huber(float(e), float(k))
        ea = abs(np.float32(e))
        ref = 1.0 if ea < k else np.sqrt(2 * k * ea - k * k) / ea
        assert w == pytest.approx(ref, rel=1e-6)
    rng = np.random.default_rng(3)
    A = rng.normal(size=(40, 6))
    H = (A.T @ A).astype(np.float32)
    Hc = np.ascontiguousarray(H.T.reshape(36))
    assert L.oracle_rank6(Hc.ctypes.data_as(C.c_void_p)) == 6
    inv = np.zeros(36, np.float32)
    assert L.oracle_inverse6(Hc.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p)) == 0
    assert np.allclose(inv.reshape(6, 6).T, np.linalg.inv(H.astype(np.float64)), rtol=2e-4, atol=1e-6)
    Hbad = H.copy()
    Hbad[:, 5] = Hbad[:, 4]
    Hbad[5, :] = Hbad[4, :]
    Hb = np.ascontiguousarray(Hbad.T.reshape(36))
    assert L.oracle_rank6(Hb.ctypes.data_as(C.c_void_p)) == 5
    from scipy.linalg import expm
    v = np.array([0.01, -0.02, 0.03, 0.02, -0.01, 0.035])
    E = np.zeros(16)
    L.oracle_se3_pseudo_exp(v.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p))
    E = E.reshape(4, 4).T
    W = np.array([[0, -v[5], v[4]], [v[5], 0, -v[3]], [-v[4], v[3], 0]])
    assert np.allclose(E[:3, :3], expm(W), atol=1e-14)
    assert np.allclose(E[:3, 3], v[:3])     # pseudo-exponential: translation copied verbatim


def test_no_valid_pixels_status(oracle_mod, small_pair):
    (rgbA, dA), (rgbB, dB), _ = small_pair
    ora = oracle_mod.Oracle(n_pyr=3)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, np.zeros_like(dB))
    st, pose = ora.align360(np.eye(4), 2)
    assert st == 2 and np.allclose(pose, np.eye(4))


def test_frame360_oracle_recovers_the_room_walls(oracle_mod):
    """Rows a14/a15 (PCL restatement): on the synthetic room the segmented planes are the six walls within 1 degree / 1 cm."""
    W, H = 512, 256
    (rgbA, dA), _, _ = synth.make_pair(W, H, seed=5)
    xyz = oracle_mod.sphere_cloud(dA, 2)
    assert np.allclose(np.linalg.norm(xyz, axis=1), dA.ravel() * 1e-3, atol=1e-5)
    nrm, win = oracle_mod.f360_normals(xyz, H, W, 0.05, 8.0, 1)
    ok = np.isfinite(nrm[:, 0])
    assert ok.mean() > 0.8 and np.allclose(np.linalg.norm(nrm[ok], axis=1), 1, atol=1e-5)
    labels, planes = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 40, 0.03, 0.05, 0.001, 1)
    big = [p for p in planes if p["count"] > 1000]
    cam = synth.CAM_A
    for ax in range(3):
        for sgn, bound in ((-1.0, synth.ROOM_HI[ax]), (1.0, synth.ROOM_LO[ax])):
            n = np.zeros(3)
            n[ax] = sgn
            hits = [p for p in big if np.dot(p["normal"], n) > np.cos(np.radians(1.0)) and abs(p["d"] - abs(bound - cam[ax])) < 0.01]
            assert hits, (ax, sgn)
    # labels are region roots: the smallest pixel index of each region
    for p in big:
        assert labels.ravel()[p["root"]] == p["root"] and (labels == p["root"]).sum() == p["count"]


def test_frame360_distance_map_is_a_chamfer_transform(oracle_mod):
    xyz = np.zeros((40 * 60, 3), np.float32)
    xyz[:, 2] = 2.0
    img = xyz.reshape(40, 60, 3)
    img[20, 30, 2] = 3.0                    # one depth spike -> depth-change pixels around it
    d = oracle_mod.f360_distance_map(xyz, 40, 60, 0.05, 0)
    assert d[20, 30] == 0 and d[20, 31] == 0 and d[21, 30] == 0
    # the depth-change set is the plus shape around the spike; (25,36) is 5 diagonal steps from (20,31)
    assert d[20, 36] == pytest.approx(5.0) and d[25, 36] == pytest.approx(5 * 1.4, abs=1e-5) and d[26, 36] == pytest.approx(8.0, abs=1e-5)


def test_analytic_warp_jacobian_matches_finite_differences(oracle_mod):
    """The restated jacobianProj23 * jacobianT36 (RPI.h:2994-3026) is the derivative of the continuous spherical
    projection (c', r') = (theta' k, h - phi' k) under the left perturbation exp(delta) * p' (hard parts)."""
    import ctypes as C
    from scipy.linalg import expm
    L = oracle_mod.lib()
    L.oracle_warp_jacobian.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_warp_jacobian.restype = None
    nCols = 2048
    k = nCols / (2 * 3.14159265359)
    rng = np.random.default_rng(4)

    def project(p):
        d = np.linalg.norm(p)
        return np.array([(np.arctan2(p[1], p[2]) + np.pi) * k, -np.arcsin(p[0] / d) * k])      # (c', r' - h)

    for _ in range(20):
        p = rng.normal(size=3) * 2.0
        if abs(p[2]) < 0.2 or np.hypot(p[1], p[2]) < 0.3:
            continue
        pf = p.astype(np.float32)
        Jw0 = np.zeros(6, np.float32)
        Jw1 = np.zeros(6, np.float32)
        L.oracle_warp_jacobian(pf.ctypes.data_as(C.c_void_p), nCols, Jw0.ctypes.data_as(C.c_void_p), Jw1.ctypes.data_as(C.c_void_p))
        J = np.stack([Jw0, Jw1]).astype(np.float64)
        Jfd = np.zeros((2, 6))
        eps = 1e-6
        for j in range(6):
            dlt = np.zeros(6)
            dlt[j] = eps
            W = np.array([[0, -dlt[5], dlt[4]], [dlt[5], 0, -dlt[3]], [-dlt[4], dlt[3], 0]])
            pp = expm(W) @ pf.astype(np.float64) + dlt[:3]
            pm = expm(-W) @ pf.astype(np.float64) - dlt[:3]
            Jfd[:, j] = (project(pp) - project(pm)) / (2 * eps)
        assert np.allclose(J, Jfd, rtol=2e-4, atol=2e-3 * np.abs(Jfd).max()), (p, J, Jfd)


# ---- occlusion-aware variants (sequential semantics of RPI.h:3232-4249) ---------------------------------------------
@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("occlusion,method", [(1, 2), (2, 0), (2, 1), (2, 2)])
def test_oracle_occlusion_matches_golden(golden, oracle_mod, math_mode, occlusion, method):
    z, j = golden
    ref = j["occlusion"]["math%d/occ%d/method%d" % (math_mode, occlusion, method)]
    (rgbA, dA), (rgbB, dB), T = synth.add_occluder(((z["rgbA"], z["dA"]), (z["rgbB"], z["dB"]), z["T_gt"]))
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    st, pose = ora.align360(np.eye(4), method, occlusion)
    assert st == ref["status"] and list(ora.result.iters)[:3] == ref["iters"]
    rot, trans = synth.pose_error(pose, np.array(ref["pose"]))
    assert rot < 1e-6 and trans < 1e-6
    assert abs(ora.result.err_final - ref["err_final"]) < 1e-9
    probe = synth.occlusion_test_poses(T)[2]
    e = ora.error_occ(1, probe, method, occlusion)
    g = ref["at_probe_level1"]
    assert (e[3], e[4]) == (g["n_photo"], g["n_depth"])
    assert abs(e[1] - g["sum_photo"]) <= 1e-9 * max(1.0, g["sum_photo"]) and abs(e[2] - g["sum_depth"]) <= 1e-9 * max(1.0, g["sum_depth"])
    H, gg, Hd, gd, nvis = ora.hessgrad_occ(1, probe, method, occlusion)
    assert nvis == g["n_visible"]
    assert np.allclose(Hd, np.array(g["H64"]), rtol=1e-9)


def test_occlusion_invariants(oracle_mod, small_pair):
    """Relations between the occlusion passes and the plain pass that follow from the reference source alone."""
    (rgbA, dA), (rgbB, dB), T = synth.add_occluder(small_pair)
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=0, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    for pose in synth.occlusion_test_poses(T):
        # Occ1's H,g pass indexes its z-buffer by the source pixel, so it rejects nothing (RPI.h:3473-3475); for PHOTO only
        # the trailing store is reached by exactly the pixels the plain pass keeps: identical normal equations
        H1, g1, Hd1, gd1, nv1 = ora.hessgrad_occ(1, pose, 0, 1)
        H0, g0, Hd0, gd0, nv0 = ora.hessgrad(1, pose, 0)
        assert nv1 == nv0
        assert np.allclose(Hd1, Hd0, rtol=1e-12) and np.allclose(gd1, gd0, rtol=1e-10, atol=1e-9)
        # with PHOTO_DEPTH a non-salient depth gradient also drops the photometric row (RPI.h:3575-3600): H can only lose
        Hd1b = ora.hessgrad_occ(1, pose, 2, 1)[2]
        Hd0b = ora.hessgrad(1, pose, 2)[2]
        assert np.all(np.diag(Hd1b) <= np.diag(Hd0b) * (1 + 1e-12))
        # Occ1 error: a target pixel holds one residual, so the sums cannot exceed the plain ones; the counters count
        # every z-buffer update, so they lie between the number of distinct targets and the plain counts
        rms0, err2_0, n0 = ora.error(1, pose, 2)
        e1 = ora.error_occ(1, pose, 2, 1)
        assert e1[1] + e1[2] <= err2_0 * (1 + 1e-12) and e1[3] + e1[4] <= n0
        # Occ2: the gate only removes pixels; distinct targets <= accepted pixels <= visible pixels
        e2 = ora.error_occ(1, pose, 2, 2)
        nv2 = ora.hessgrad_occ(1, pose, 2, 2)[4]
        assert nv2 <= e2[3] == e2[4] <= nv0
    # single modality under Occ1: avPhoto + avDepth has a 0/0 term -> NaN, no iteration, guess returned
    st, pose = ora.align360(np.eye(4), 0, 1)
    assert st == 2 and np.allclose(pose, np.eye(4)) and list(ora.result.iters)[:3] == [0, 0, 0]


def test_occlusion_zbuffer_known_answer(oracle_mod):
    """Hand-checkable case: two flat frames 2 m away, the source carrying a near patch that a sideways pose moves in
    front of far pixels.  Every number below follows from counting pixels."""
    W, H = 64, 32
    rgb = np.zeros((H, W, 3), np.uint8)
    rgb[..., :] = (np.arange(W)[None, :, None] * 4) % 256            # horizontal ramp: salient everywhere inside
    d_far = np.full((H, W), 2000, np.uint16)
    ora = oracle_mod.Oracle(n_pyr=1, math_mode=0, reduce_mode=1, mask_seams=0)
    ora.set_target(rgb, d_far)
    ora.set_source(rgb, d_far)
    e_plain = ora.error(0, np.eye(4), 2)
    e1 = ora.error_occ(0, np.eye(4), 2, 1)
    e2 = ora.error_occ(0, np.eye(4), 2, 2)
    # identity pose, identical frames: one source pixel per target pixel, nothing gated, nothing occluded
    assert e1[3] + e1[4] == e_plain[2]
    assert e2[3] == e2[4] == ora.hessgrad(0, np.eye(4), 2)[4] == ora.hessgrad_occ(0, np.eye(4), 2, 2)[4]
    # a source whose left half is 1 m closer: under Occ2 every pixel of that half fails |Dtrg - dist| <= 0.3
    d_src = d_far.copy()
    d_src[:, : W // 2] = 1000
    ora.set_source(rgb, d_src)
    e2 = ora.error_occ(0, np.eye(4), 2, 2)
    assert e2[3] == H * W // 2
    assert ora.hessgrad_occ(0, np.eye(4), 2, 2)[4] == H * W // 2


# ---- pinhole single-sensor path (RPI.h:4254-4512) ------------------------------------------------------------------------
def test_se3_exp_matches_matrix_exponential(oracle_mod):
    """CPose3D::exp(., false) restatement against scipy's expm of the 4x4 twist, through all three branches of the
    small-angle series (theta^2 < 1e-8, < 1e-6, general)."""
    import scipy.linalg
    rng = np.random.default_rng(3)
    for scale in (1.0, 3e-3, 5e-4, 1e-5, 0.0):
        v = rng.normal(size=6) * np.array([0.3, 0.3, 0.3, scale, scale, scale])
        X = np.zeros((4, 4))
        X[:3, :3] = np.array([[0, -v[5], v[4]], [v[5], 0, -v[3]], [-v[4], v[3], 0]])
        X[:3, 3] = v[:3]
        # below theta^2 = 1e-8 MRPT truncates the translation coupling after the first-order term: error ~ theta^2 |u| / 6
        tol = 1e-13 if (v[3:] ** 2).sum() >= 1e-8 else 1e-10
        assert np.abs(oracle_mod.se3_exp(v) - scipy.linalg.expm(X)).max() < tol


def test_pinhole_lut_and_projection_roundtrip(oracle_mod):
    """Known answer: back-projecting the source depth with K and re-projecting with the identity pose lands every valid
    pixel on itself (RPI.h:4277-4300 against :701-708)."""
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(160, 120, seed=5)
    for mm in (0, 1):
        ora = oracle_mod.Oracle(n_pyr=2, math_mode=mm, reduce_mode=1, mask_seams=0)
        ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
        for level in (0, 1):
            rows, cols = ora.level_dims(level)
            idx = ora.warp_indices_pinhole(level, np.eye(4))
            rr, cc = np.divmod(np.arange(rows * cols), cols)
            valid = ora.lut_pinhole(level)[:, 0] != -10000
            assert valid.mean() > 0.9
            assert np.array_equal(idx[valid, 0], rr[valid]) and np.array_equal(idx[valid, 1], cc[valid])
            lut = ora.lut_pinhole(level)
            z = ora.plane("depth_src", level).reshape(-1)
            assert np.array_equal(lut[:, 2], z)
            s = 2.0 ** -level
            assert np.allclose(lut[valid, 0], (cc[valid] - K[2] * s) * z[valid] / (K[0] * s), rtol=1e-5, atol=1e-6)


def test_pinhole_alignment_improves_on_the_guess(oracle_mod):
    """Functional: PHOTO_DEPTH moves the identity guess towards the rendered motion; PHOTO only reproduces the reference's
    NaN (both averages are divided by nValidDepthPts, RPI.h:742-743) and returns the guess."""
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    ora = oracle_mod.Oracle(n_pyr=3, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    st, pose = ora.align_pinhole(np.eye(4), 2)
    r0, t0 = synth.pose_error(np.eye(4), T)
    r1, t1 = synth.pose_error(pose, T)
    assert st == 0 and sum(list(ora.result.iters)[:3]) >= 2
    assert t1 < 0.6 * t0 and r1 < 0.6 * r0, (r0, t0, r1, t1)
    st, pose = ora.align_pinhole(np.eye(4), 0)
    assert st == 2 and np.allclose(pose, np.eye(4))
    # the cost the driver minimises is far lower at the rendered motion than at the guess, on every level
    for level in range(3):
        assert ora.error_pinhole(level, T, 2)[0] < 0.4 * ora.error_pinhole(level, np.eye(4), 2)[0]


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("method", [0, 1, 2])
def test_oracle_pinhole_matches_golden(golden, oracle_mod, math_mode, method):
    import zlib
    _, j = golden
    G = j["pinhole"]
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
    assert [crc(rgbA), crc(dA), crc(rgbB), crc(dB)] == G["crc32_inputs"]          # the generator reproduces the recorded inputs
    ref = G["runs"]["math%d/method%d" % (math_mode, method)]
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    st, pose = ora.align_pinhole(np.eye(4), method)
    assert st == ref["status"] and list(ora.result.iters)[:3] == ref["iters"]
    rot, trans = synth.pose_error(pose, np.array(ref["pose"]))
    assert rot < 1e-6 and trans < 1e-6
    tr = ora.trace()
    assert [(t["level"], t["it"], t["accepted"], t["n_valid"]) for t in tr] == [(t["level"], t["it"], t["accepted"], t["n_valid"]) for t in ref["trace"]]
    g = ref["at_gt_level1"]
    e = ora.error_pinhole(1, T, method)
    assert (e[3], e[4]) == (g["n_photo"], g["n_depth"])
    assert abs(e[1] - g["sum_photo"]) <= 1e-9 * max(1.0, g["sum_photo"]) and abs(e[2] - g["sum_depth"]) <= 1e-9 * max(1.0, g["sum_depth"])
    H, gg, Hd, gd, nrows = ora.hessgrad_pinhole(1, T, method)
    assert nrows == g["n_rows"] and np.allclose(Hd, np.array(g["H64"]), rtol=1e-9)


@pytest.mark.parametrize("occlusion", [1, 2])
def test_numpy_restatement_agrees_on_occlusion_passes(oracle_mod, occlusion):
    """The occlusion error passes against the independent numpy restatement (group-by-target prefix maxima instead of a
    z-buffer swept in pixel order)."""
    sys.path.insert(0, HERE)
    import np_restatement as NP
    (rgbA, dA), (rgbB, dB), T = synth.add_occluder(synth.make_pair(128, 64, seed=99))
    ora = oracle_mod.Oracle(n_pyr=2, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    fr = NP.Frames(rgbA, dA, rgbB, dB, n_pyr=2)
    conflicts = 0
    for level in range(2):
        for pose in synth.occlusion_test_poses(T):
            for method in (0, 1, 2):
                _, sp, sd, n_p, n_d = ora.error_occ(level, pose, method, occlusion)
                a, b, c, d = NP.occ_error(fr, level, pose, method, occlusion)
                if method in (0, 2):
                    assert abs(c - n_p) <= max(3, 2e-3 * n_p), (level, method, c, n_p)
                    assert abs(a - sp) <= 5e-3 * max(sp, 1.0)
                if method in (1, 2):
                    assert abs(d - n_d) <= max(3, 2e-3 * n_d), (level, method, d, n_d)
                    assert abs(b - sd) <= 5e-3 * max(sd, 1.0)
            if occlusion == 2:
                nv = ora.hessgrad_occ(level, pose, 2, 2)[4]
                assert abs(NP.occ2_visible(fr, level, pose) - nv) <= 3
                conflicts += int(nv != ora.hessgrad(level, pose, 2)[4])
            else:
                conflicts += int(ora.error_occ(level, pose, 2, 1)[3] != ora.error(level, pose, 0)[2])
    assert conflicts > 0            # the scene does exercise the z-buffer


def test_numpy_restatement_agrees_on_pinhole_error(oracle_mod):
    sys.path.insert(0, HERE)
    import np_restatement as NP
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(160, 120, seed=5)
    ora = oracle_mod.Oracle(n_pyr=2, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    fr = NP.Frames(rgbA, dA, rgbB, dB, n_pyr=2, mask=False)
    for level in range(2):
        for pose in (np.eye(4), T):
            for method in (0, 1, 2):
                _, sp, sd, n_p, n_d = ora.error_pinhole(level, pose, method)
                a, b, c, d = NP.pinhole_error(fr, level, pose, K, method)
                assert abs(c - n_p) <= max(3, 2e-3 * n_p) and abs(d - n_d) <= max(3, 2e-3 * n_d), (level, method, c, n_p, d, n_d)
                assert abs(a - sp) <= 5e-3 * max(sp, 1.0) and abs(b - sd) <= 5e-3 * max(sd, 1.0)


def _noisy_pinhole_cloud(rows=120, cols=160, seed=3, noise=0.01):
    """A wall 2.5 m in front of a pinhole camera (slanted), a nearer box, holes, Gaussian depth noise."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 131.25, cols / 2 - 0.5, rows / 2 - 0.5
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    z_true = 2.5 + 0.4 * (u - cx) / f
    z_true[40:70, 50:90] = 1.4                                   # a box in front of the wall: a 1 m depth step
    z = z_true + rng.normal(size=z_true.shape) * noise
    z[10:14, 100:130] = np.nan                                   # a hole
    z[rng.random(z.shape) < 0.02] = np.nan
    xyz = np.stack([(u - cx) * z / f, (v - cy) * z / f, z], -1).astype(np.float32)
    xyz[~np.isfinite(xyz[..., 2])] = np.nan
    return xyz, z_true


def test_fast_bilateral_smooths_within_surfaces_and_keeps_depth_steps(oracle_mod):
    """pcl::FastBilateralFilter restated (Frame360.h:493-499 set-up): noise on a surface shrinks several times, the 1 m step between
    box and wall stays a step, x / y and invalid points are untouched, a noiseless fronto-parallel plane is a fixed point."""
    xyz, z_true = _noisy_pinhole_cloud()
    rows, cols = z_true.shape
    out = oracle_mod.fast_bilateral(xyz, rows, cols, 10.0, 0.05).reshape(rows, cols, 3)
    ok = np.isfinite(xyz[..., 2])
    assert np.array_equal(np.isfinite(out[..., 0]), np.isfinite(xyz[..., 0]))            # x, y copied (NaN stays NaN)
    assert np.array_equal(out[..., :2][ok], xyz[..., :2][ok])
    inner = ok.copy()
    inner[:12] = inner[-12:] = False
    inner[:, :12] = inner[:, -12:] = False
    inner[36:74, 46:94] = False                                                          # away from the depth step
    e_in = (xyz[..., 2] - z_true)[inner].std()
    e_out = (out[..., 2] - z_true)[inner].std()
    assert e_out < 0.4 * e_in, (e_in, e_out)
    box = np.zeros_like(ok)
    box[44:66, 54:86] = True
    assert abs(np.nanmedian(out[..., 2][box & ok]) - 1.4) < 0.01                         # the box did not bleed into the wall
    flat = np.zeros((rows, cols, 3), np.float32)
    flat[..., 2] = 2.0
    same = oracle_mod.fast_bilateral(flat, rows, cols, 10.0, 0.05).reshape(rows, cols, 3)
    assert np.abs(same[..., 2] - 2.0).max() < 1e-6
    allnan = np.full((8, 9, 3), np.nan, np.float32)
    assert np.isnan(oracle_mod.fast_bilateral(allnan, 8, 9)).all()


def test_plane_refinement_grows_planes_into_their_noisy_border(oracle_mod):
    """oracle_f360_plane_refine (the `refine` half of segmentAndRefine, Frame360.h:977) on a hand-made scene: a wall z = 2 with a
    noisy patch (no normals there -> the patch is left out by `segment`) and a patch 5 cm in front of the wall.  The plane grows
    through the noisy patch (within 2 cm) and not into the offset one; the raster passes reach pixels that are several steps away
    from the original region in every direction; counts and extents follow, the plane model does not move."""
    H, W = 40, 60
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32) * 0.01 - 0.3, np.arange(H, dtype=np.float32) * 0.01 - 0.2)
    xyz = np.stack([xs, ys, np.full_like(xs, 2.0)], -1)
    rng = np.random.default_rng(0)
    xyz[10:20, 15:30, 2] += rng.uniform(-0.015, 0.015, size=(10, 15)).astype(np.float32)        # noisy, but within 2 cm
    xyz[25:32, 35:50, 2] = 1.95                                                                  # 5 cm off the wall
    xyz[5:8, 5:8] = np.nan
    nrm = np.zeros_like(xyz)
    nrm[..., 2] = -1.0
    nrm[10:20, 15:30] = np.nan                                                                   # no normals on the noisy patch
    nrm[25:32, 35:50] = np.nan
    labels, planes = oracle_mod.f360_plane_segment(xyz, nrm, H, W, 40, 0.05, 0.02, 0.01, 0)
    assert len(planes) == 1
    root = planes[0]["root"]
    assert (labels[10:20, 15:30] != root).all() and (labels[25:32, 35:50] != root).all()
    ref, planes2, changed = oracle_mod.f360_plane_refine(xyz, H, W, labels, planes, 0.02)
    assert changed == 10 * 15 and (ref[10:20, 15:30] == root).all()
    assert (ref[25:32, 35:50] != root).all() and (ref[5:8, 5:8] == -1).all()
    assert planes2[0]["count"] == planes[0]["count"] + changed
    assert np.array_equal(planes2[0]["normal"], planes[0]["normal"]) and planes2[0]["d"] == planes[0]["d"]
    # filling the hole moves mass towards the centre: the moment rectangle shrinks a little, towards the true 0.6 x 0.4 m wall
    assert 0.22 < planes2[0]["area"] < planes[0]["area"]
    # idempotent: a second refinement finds nothing left to grow
    ref2, _, changed2 = oracle_mod.f360_plane_refine(xyz, H, W, ref, planes2, 0.02)
    assert changed2 == 0 and np.array_equal(ref2, ref)


# ---- pinhole occlusion-aware passes and the salient-pixel list (RPI.h:1107-2030, 401-425, 590-690) ----------------------------
def _pinhole_occ_golden():
    import json
    with open(os.path.join(HERE, "golden", "pinhole_occ.json")) as f:
        return json.load(f)


def _probe_pose(T):
    back = np.eye(4)
    back[2, 3] = 0.6
    return back @ T


@pytest.mark.parametrize("math_mode", [0, 1])
def test_oracle_pinhole_occlusion_matches_golden(oracle_mod, math_mode):
    import zlib
    G = _pinhole_occ_golden()
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
    assert [crc(rgbA), crc(dA), crc(rgbB), crc(dB)] == G["crc32_inputs"]
    ora = oracle_mod.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    for occ in (1, 2):
        for method in (0, 1, 2):
            ref = G["occ"]["math%d/occ%d/method%d" % (math_mode, occ, method)]
            st, pose = ora.align_pinhole(np.eye(4), method, occ)
            assert st == ref["status"] and list(ora.result.iters)[:3] == ref["iters"]
            rot, trans = synth.pose_error(pose, np.array(ref["pose"]))
            assert rot < 1e-6 and trans < 1e-6
            assert abs(ora.result.sso - ref["sso"]) < 1e-6
            for name, pp, level in (("at_gt_level1", T, 1), ("at_probe_level0", _probe_pose(T), 0)):
                g = ref[name]
                e = ora.error_pinhole_occ(level, pp, method, occ)
                assert (e[3], e[4]) == (g["n_photo"], g["n_depth"])
                assert abs(e[1] - g["sum_photo"]) <= 1e-9 * max(1.0, g["sum_photo"]) and abs(e[2] - g["sum_depth"]) <= 1e-9 * max(1.0, g["sum_depth"])
                H, gg, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pp, method, occ)
                assert nvis == g["n_visible"] and np.allclose(Hd, np.array(g["H64"]), rtol=1e-9) and np.allclose(gd, np.array(g["g64"]), rtol=1e-9, atol=1e-12)
    ora.use_saliency(True, 0.01)
    S = G["salient"]["math%d" % math_mode]
    for level in range(3):
        v = ora.salient_pixels(level)
        assert len(v) == S["list_len"][level] and crc(v.astype(np.int32)) == S["list_crc"][level]
    for method in (1, 2):
        g = S["method%d" % method]
        e = ora.error_pinhole_salient(1, T, method)
        assert (e[3], e[4]) == (g["n_photo"], g["n_depth"]) and abs(e[1] - g["sum_photo"]) <= 1e-9 * max(1.0, g["sum_photo"])
        st, pose = ora.align_pinhole(np.eye(4), method, 0)
        assert st == g["status"] and list(ora.result.iters)[:3] == g["iters"]
        rot, trans = synth.pose_error(pose, np.array(g["pose"]))
        assert rot < 1e-6 and trans < 1e-6


def test_numpy_restatement_agrees_on_pinhole_occlusion_passes(oracle_mod):
    """errorPhotoICP_Occ1/2 and calcHessGrad_Occ1/2 against the independent numpy restatement (grouped by target pixel: prefix maxima
    of 1/Z instead of a z-buffer swept in pixel order), at the rendered motion and at a pose that piles up to four source pixels on a
    target pixel; counts exact, sums to float accumulation error."""
    sys.path.insert(0, HERE)
    import np_restatement as NP
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(160, 120, seed=5)
    ora = oracle_mod.Oracle(n_pyr=2, reduce_mode=1, mask_seams=0)
    ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
    fr = NP.Frames(rgbA, dA, rgbB, dB, n_pyr=2, mask=False)
    longest = 0
    for level in range(2):
        for pose in (T, _probe_pose(T)):
            idx = ora.warp_indices_pinhole(level, pose)
            v = idx[:, 0] >= 0
            longest = max(longest, np.unique(idx[v, 0] * 4096 + idx[v, 1], return_counts=True)[1].max())
            for occ in (1, 2):
                for method in (0, 1, 2):
                    _, sp, sd, n_p, n_d = ora.error_pinhole_occ(level, pose, method, occ)
                    H, g, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pose, method, occ)
                    a = NP.pinhole_occ(fr, level, pose, K, method, occ)
                    assert (n_p, n_d, nvis) == (a[2], a[3], a[6]), (level, occ, method)
                    assert abs(a[0] - sp) <= 1e-6 * max(sp, 1.0) and abs(a[1] - sd) <= 1e-6 * max(sd, 1.0)
                    assert np.abs(Hd - a[4]).max() <= 1e-6 * max(np.abs(Hd).max(), 1e-9) and np.abs(gd - a[5]).max() <= 1e-6 * max(np.abs(gd).max(), 1e-9)
                    if method == 1:
                        assert not Hd.any()          # both row sums test the PHOTO residual (RPI.h:1523, 1531): depth alone sums nothing
        assert np.array_equal(ora.salient_pixels(level), NP.pinhole_salient_list(fr, level))
    assert longest >= 3


def test_pinhole_occlusion_zbuffer_is_sequential(oracle_mod):
    """Known answers of the index-order sweep.  Pushing every point away along the optical axis shrinks the warped image, so several
    source pixels share a target pixel.  If the source depth GROWS with the pixel index, a list's later pixels are farther: only the first
    is accepted.  If it SHRINKS, every later pixel is closer: all are accepted, and each counts.  thres_sal_photo = 0 makes every target
    pixel salient (`|g| < 0` never holds), so the photo counter counts the accepted pixels; numVisiblePixels counts a target pixel's first
    arrival twice (RPI.h:1421-1430)."""
    rows, cols = 24, 32
    rng = np.random.default_rng(0)
    rgbA = rng.integers(0, 255, (rows, cols, 3)).astype(np.uint8)
    rgbB = rng.integers(0, 255, (rows, cols, 3)).astype(np.uint8)
    dA = np.full((rows, cols), 2000, np.uint16)
    ramp = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    K = (30.0, 30.0, 15.5, 11.5)
    P = np.eye(4)
    P[2, 3] = 2.5
    for math_mode in (0, 1):
        for growing in (True, False):
            dB = (1500 + ramp if growing else 1500 + ramp[::-1, ::-1]).astype(np.uint16)
            ora = oracle_mod.Oracle(n_pyr=1, math_mode=math_mode, reduce_mode=1, mask_seams=0, thres_sal_photo=0.0, thres_sal_depth=0.0)
            ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
            idx = ora.warp_indices_pinhole(0, P)
            v = idx[:, 0] >= 0
            per_target = np.unique(idx[v, 0] * cols + idx[v, 1], return_counts=True)[1]
            assert per_target.max() >= 3 and v.sum() > 2 * len(per_target)
            accepted = len(per_target) if growing else int(v.sum())
            _, sp, sd, n_p, n_d = ora.error_pinhole_occ(0, P, 2, 1)
            assert n_p == accepted and n_d == accepted
            nvis = ora.hessgrad_pinhole_occ(0, P, 2, 1)[4]
            assert nvis == accepted + len(per_target)
