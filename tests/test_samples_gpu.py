"""BASELINE.json configs[0] ON THE DEVICE: the one real RGB-D pair the reference ships (samples/sphere_images_1.bin vs _10.bin, the
sequence of Registration/RegisterPairRGBD360.cpp:56-142 and OdometryRGBD360.cpp:141-193) through the HIP path.

The sensor images and the rig's extrinsics travel as a data fixture (tests/golden/sample_pair.npz, written by
tests/golden/make_golden_samples.py from the reference's files in the build container); the expected values are
tests/golden/config1_samples.json (tests/tools/config1_samples.py --write: the CPU oracle on the same data) AND the oracle run
live beside the device.  What this pair has that the synthetic room does not: sensor noise, 21 % of the panorama without depth,
seams between the eight sensors, levels that end at the iteration limit ([10, 10, 10, 7] accepted steps per level).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rgbd360_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

ROT_TOL, TRANS_TOL = 1e-4, 1e-3     # north star: pose vs the reference's arithmetic (oracle math_mode 0)
POSE_TOL_DEV = 5e-6                 # vs the oracle in the device's arithmetic (same pixels; float32 rounding of rows and weights only)
# 30-37 accepted steps on sensor data: every step's 1e-7 rounding differences of the float32 sums go through an ill-conditioned
# solve (this pair: floor and ceiling dominate, the translation along the corridor is weakly observable).  Measured below.


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "config1_samples.json")))


@pytest.fixture(scope="module")
def sample():
    import config1_samples as c1
    fr = {idx: c1.frames(idx, "fixture") for idx in (1, 10)}
    Rt = c1.extrinsics("fixture")
    Rt_inv = np.stack(c1.load_extrinsics("fixture"))
    return dict(frames=fr, Rt=Rt, Rt_inv=Rt_inv, c1=c1)


@pytest.fixture(scope="module")
def panos(hip_lib, sample):
    """The two panoramas stitched ON THE DEVICE (rgbd360_stitch_sphere)."""
    from rgbd360_amd.register import RegisterPhotoICP, stitch_sphere
    reg = RegisterPhotoICP()
    out = []
    for idx in (1, 10):
        fr = sample["frames"][idx]
        out.append(stitch_sphere(reg, np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), sample["Rt_inv"]))
    return out


def test_sample_panoramas_stitched_on_the_device_have_the_committed_checksums(hip_lib, oracle_mod, sample, panos, gold):
    """Frame360::stitchSphericalImage (Frame360.h:386-405, 1099-1148) of both sample frames: byte for byte the panoramas the oracle
    stitches, whose CRCs were recorded from the reference's own files."""
    c1 = sample["c1"]
    assert panos[0][0].shape == (320, 1920, 3) and panos[0][1].shape == (320, 1920)
    got = {"rgb_1": c1.crc(panos[0][0]), "depth_1": c1.crc(panos[0][1]), "rgb_10": c1.crc(panos[1][0]), "depth_10": c1.crc(panos[1][1])}
    assert got == gold["crc32"]
    for (rgb, d), idx in zip(panos, (1, 10)):
        fr = sample["frames"][idx]
        a, b = oracle_mod.stitch_sphere(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), sample["Rt_inv"])
        assert np.array_equal(rgb, a) and np.array_equal(d, b)
        assert 0.75 < (d > 0).mean() < 0.85                      # a fifth of the sphere carries no depth


def _align_both(hip_lib, oracle_mod, panos, method, occlusion):
    from rgbd360_amd.register import RegisterPhotoICP
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    reg.setTargetFrame(*panos[0])
    reg.setSourceFrame(*panos[1])
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    ora = oracle_mod.Oracle(n_pyr=4, math_mode=1, reduce_mode=1)
    ora.set_target(*panos[0])
    ora.set_source(*panos[1])
    st, pose_dev = ora.align360(np.eye(4), method, occlusion)
    return reg, rc, ora, st, pose_dev


@pytest.mark.parametrize("method,occlusion", [(0, 0), (1, 0), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2)])
def test_sample_pair_alignment_matches_the_oracle(hip_lib, oracle_mod, panos, gold, method, occlusion):
    """rgbd360_align360 on the real pair, every cost function, plain and occlusion-aware: status, the accept / reject sequence
    (iterations per level: the photometric modes stop at the iteration limit on three levels) and the pose against the oracle in
    the device's arithmetic run live, and the pose against the committed record of the reference's arithmetic (libm asinf / atan2f /
    roundf, float32 accumulators) within the north star's 1e-4 rad / 1e-3 m."""
    reg, rc, ora, st, pose_dev = _align_both(hip_lib, oracle_mod, panos, method, occlusion)
    rec = gold["alignments"]["m%d_o%d" % (method, occlusion)]
    assert rc == st == rec["device"]["status"] == rec["libm"]["status"] == 0
    iters = list(reg.num_iterations)
    assert iters == list(ora.result.iters)[:4] == rec["device"]["iters"], (iters, list(ora.result.iters)[:4], rec["device"]["iters"])
    if (method, occlusion) == (0, 0) or (method, occlusion) == (2, 0):
        assert iters == [10, 10, 10, 7]                           # tests/golden/config1_samples.json, both arithmetic modes
    pose = reg.getOptimalPose()
    rot, trans = synth.pose_error(pose, pose_dev)
    rot_l, trans_l = synth.pose_error(pose, np.array(rec["libm"]["pose"]))
    rot_r, trans_r = synth.pose_error(pose, np.array(rec["device"]["pose"]))
    print(f"sample pair m{method} o{occlusion}: iters {iters} vs oracle(device arithmetic) {rot:.2e} rad {trans:.2e} m; "
          f"vs the committed libm record {rot_l:.2e} rad {trans_l:.2e} m (libm iters {rec['libm']['iters']})")
    assert rot <= POSE_TOL_DEV and trans <= 4 * POSE_TOL_DEV, (rot, trans)
    assert rot_r <= POSE_TOL_DEV and trans_r <= 4 * POSE_TOL_DEV, (rot_r, trans_r)      # the live oracle is the recorded one
    assert rot_l <= ROT_TOL and trans_l <= TRANS_TOL, (rot_l, trans_l)
    assert abs(reg.avResidual - ora.result.err_final) <= 1e-4 * max(1.0, ora.result.err_final)
    assert abs(reg.SSO - ora.result.sso) < 1e-5
    # the motion between frames 1 and 10 of a hand-held rig: a few decimetres, about a degree (plain modes)
    if occlusion == 0 and method != 1:
        rot0, trans0 = synth.pose_error(pose, np.eye(4))
        assert 0.05 < trans0 < 0.6 and rot0 < 0.1


@pytest.mark.parametrize("method,occlusion", [(0, 0), (1, 0), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2)])
def test_sample_pair_alignment_in_the_reference_arithmetic(hip_lib, oracle_mod, panos, gold, method, occlusion):
    """The same alignments with rgbd360_set_index_arithmetic(ctx, 1) -- the warp in the reference's own arithmetic (csrc/libm_f32.h) --
    against the COMMITTED record of the reference-faithful oracle (libm, float32 accumulators; generated where the reference's samples
    are): status, iterations per level, and the pose far inside the north-star tolerance; every warped index of the finest level at the
    recorded pose equals the live libm oracle's."""
    from rgbd360_amd.register import RegisterPhotoICP
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    reg.set_index_arithmetic(1)
    reg.setTargetFrame(*panos[0])
    reg.setSourceFrame(*panos[1])
    rc = reg.alignFrames360(np.eye(4), method, occlusion)
    rec = gold["alignments"]["m%d_o%d" % (method, occlusion)]["libm"]
    iters = list(reg.num_iterations)
    # live: the libm warp with float64 sums (the device's sums are float64 partials too)
    ora = oracle_mod.Oracle(n_pyr=4, math_mode=0, reduce_mode=1)
    ora.set_target(*panos[0])
    ora.set_source(*panos[1])
    st, pose_live = ora.align360(np.eye(4), method, occlusion)
    assert rc == st == rec["status"] == 0
    assert iters == list(ora.result.iters)[:4], (iters, list(ora.result.iters)[:4])
    rot, trans = synth.pose_error(reg.getOptimalPose(), pose_live)
    assert rot <= POSE_TOL_DEV and trans <= 4 * POSE_TOL_DEV, (rot, trans)
    # committed: libm warp + the reference's float32 accumulators.  Photo + depth with the z-buffer gate (2, 2) ends level 2 after 3 steps
    # there (and in the device-arithmetic record) and after 10 with the libm warp and float64 sums -- a step the accumulators decide; with
    # three levels at their iteration limit the two runs then end 9e-3 rad / 6e-2 m apart.  The device follows the float64-sum oracle to
    # 1e-9; a float32 accumulation in the reference's order is sequential by definition.
    rot_r, trans_r = synth.pose_error(reg.getOptimalPose(), np.array(rec["pose"]))
    print(f"sample pair m{method} o{occlusion}, reference arithmetic: iters {iters} (record {rec['iters']}), vs the live libm oracle {rot:.2e} rad {trans:.2e} m, "
          f"vs the committed libm record {rot_r:.2e} rad {trans_r:.2e} m")
    if iters == rec["iters"]:
        assert rot_r <= 2e-5 and trans_r <= 1e-4, (rot_r, trans_r)
    else:
        assert (method, occlusion) == (2, 2) and iters == [10, 10, 10, 7] and rec["iters"] == [10, 10, 3, 7], (iters, rec["iters"])
    P = np.array(rec["pose"])
    assert np.array_equal(reg.warp_indices(0, P), ora.warp_indices(0, P))


def test_sample_pair_occlusion1_single_modality_has_no_valid_pixel(hip_lib, oracle_mod, panos):
    """Occ1 counts a pixel only where BOTH residuals exist (RPI.h:3298-3340): photometric-only and depth-only runs end with status 2
    and the guess, on the device as in the oracle."""
    for method in (0, 1):
        reg, rc, ora, st, pose_dev = _align_both(hip_lib, oracle_mod, panos, method, 1)
        assert rc == st == 2
        assert np.array_equal(reg.getOptimalPose(), np.eye(4, dtype=np.float32))


@pytest.mark.parametrize("method", [0, 2])
def test_sample_pair_per_pass_sums_match_the_oracle(hip_lib, oracle_mod, panos, method):
    """One evaluation per level at the identity and at the recovered pose: counts exact (holes, seams, out-of-range depth), sums to
    float32 rounding."""
    reg, rc, ora, st, pose_dev = _align_both(hip_lib, oracle_mod, panos, method, 0)
    for level in range(4):
        for pose in (np.eye(4), pose_dev):
            e = reg.eval(level, pose, method)
            rms, err2, nvalid = ora.error(level, pose, method)
            H, g, Hd, gd, nvis = ora.hessgrad(level, pose, method)
            assert e["n_valid"] == nvalid and e["n_visible"] == nvis
            assert abs(e["err2"] - err2) <= 2e-6 * max(1.0, abs(err2))
            assert np.abs(e["H64"] - Hd).max() <= 2e-5 * np.abs(Hd).max()
            a, b = reg.warp_indices(level, pose), ora.warp_indices(level, pose)
            assert np.array_equal(a, b)


def test_sample_pair_in_the_sequence_engine_equals_the_pairwise_call(hip_lib, panos):
    """The lock-step sequence engine (rgbd360_align360_batch) on [frame 1, frame 10, frame 1]: poses bit-identical to rgbd360_align360
    pair by pair -- also where levels run to the iteration limit."""
    from rgbd360_amd.register import RegisterPhotoICP
    frames = [panos[0], panos[1], panos[0]]
    reg = RegisterPhotoICP()
    reg.setNumPyr(4)
    poses, status, iters = reg.alignSequence(frames, 2)
    for i in range(2):
        one = RegisterPhotoICP()
        one.setNumPyr(4)
        one.setTargetFrame(*frames[i])
        one.setSourceFrame(*frames[i + 1])
        assert one.alignFrames360(np.eye(4), 2) == status[i] == 0
        assert list(one.num_iterations) == list(iters[i])
        assert np.array_equal(one.getOptimalPose(), poses[i])
    # 1 -> 10 and 10 -> 1: inverse motions only as far as this pair constrains them (measured 0.012 rad / 0.14 m: floor and ceiling
    # dominate, the motion along the corridor is weakly observable -- the same reason the plane registration returns status 2)
    rot, trans = synth.pose_error(poses[0] @ poses[1], np.eye(4))
    print(f"sample pair forward o backward: {rot:.3e} rad {trans:.3e} m")
    assert rot < 0.05 and trans < 0.3, (rot, trans)


def _oracle_sensor_planes(O, depth, Rt):
    cloud = O.sensor_cloud(depth, 2, 0.3, 10.0)
    H, W, _ = cloud.shape
    xyz = O.fast_bilateral(cloud, H, W, 10.0, 0.05)
    nrm, _w = O.f360_normals(xyz, H, W, 0.02, 8.0, 0)
    planes = O.f360_plane_segment(xyz, nrm, H, W, 40, 0.0398, 0.02, 0.0013, 0, max_planes=512)[1]
    R, t = Rt[:3, :3], Rt[:3, 3]
    out = []
    for p in planes:
        n, c = R @ p["normal"].astype(np.float64), R @ p["centroid"].astype(np.float64) + t
        if n @ c > 0:
            n = -n
        q = dict(p)
        q.update(normal=n.astype(np.float32), centroid=c.astype(np.float32), d=np.float32(-n @ c),
                 ppal_dir=(R @ p["ppal_dir"].astype(np.float64)).astype(np.float32))
        out.append(q)
    return out


def test_sample_pair_plane_chain_equals_the_oracle_chain(hip_lib, oracle_mod, sample):
    """The plane side of RegisterPairRGBD360.cpp:60-110 on the real frames: rgbd360_sensor_planes x 8 (pinhole cloud down-sampled by 2,
    bilateral filter, normal map, regions, plane.transform(Rt): Frame360.h:479-499, 949-1075) against the oracle's stages per sensor
    -- same regions, plane parameters to float32 rounding -- then RegisterPbMap's matcher on both plane sets: same matches, and the
    verdict this pair deserves (floor, ceiling and one wall direction: the translation along that wall is not observable, status 2).
    groupPlanes + mergePlanes on the device planes (Frame360.h:615-639) pool the pieces and keep the verdict."""
    from rgbd360_amd import pbmap
    from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP
    from tests.test_gpu_parity import _check_hull_polygon
    st = Frame360Stages(RegisterPhotoICP())
    dev, ora = [], []
    worst_d, worst_n = {False: 0.0, True: 0.0}, {False: 0.0, True: 0.0}
    for idx in (1, 10):
        per_sensor_dev, flat_ora = [], []
        for s, (rgb, depth) in enumerate(sample["frames"][idx]):
            Rt = sample["Rt"][s]
            got = st.sensor_planes(depth, 2, 0.3, 10.0, 10.0, 0.05, 0.02, 8.0, 40, 0.0398, 0.02, 0.0013, Rt)
            want = _oracle_sensor_planes(oracle_mod, depth, Rt)
            assert [(p["root"], p["count"]) for p in got] == [(p["root"], p["count"]) for p in want], (idx, s)
            for a, b in zip(got, want):
                # slivers of a few dozen points (elongation 20-40: nearly a line; the reference drops them, elongation > 6,
                # Frame360.h:1041) have two small eigenvalues close together and a normal that turns by up to 1e-3 rad with the
                # last bit of the moments: their offset d = -n.c follows it (measured: 2.3e-3 m at 2.5 m from the rig)
                sliver = b["elongation"] > 10.0
                assert np.allclose(a["centroid"], b["centroid"], atol=5e-5), (idx, s, a["root"])
                worst_d[sliver] = max(worst_d[sliver], abs(float(a["d"]) - float(b["d"])))
                if b["curvature"] > 1e-9:
                    worst_n[sliver] = max(worst_n[sliver], 1.0 - float(a["normal"].astype(np.float64) @ b["normal"].astype(np.float64)))
                assert abs(a["area_moment"] - b["area"]) <= 1e-3 * max(b["area"], 0.1)
            for a in got:
                if a["hull_points"] >= 3 and a["area"] > 0.05:
                    _check_hull_polygon(a)
            per_sensor_dev.append(got)
            flat_ora += want
        dev.append(per_sensor_dev)
        ora.append(flat_ora)
    print("plane parameters, device vs oracle: |d| %.2e m, 1 - cos %.2e (regions), %.2e m, %.2e (slivers)" % (worst_d[False], worst_n[False], worst_d[True], worst_n[True]))
    assert worst_d[False] < 2e-4 and worst_n[False] < 2e-6
    assert worst_d[True] < 1e-2 and worst_n[True] < 1e-4
    flat_dev = [[p for lst in f for p in lst] for f in dev]
    assert min(len(f) for f in flat_dev) >= 20
    params = pbmap.default_params(True)
    # the matcher on the raw pieces, device planes with their hull-true areas replaced by the moment areas the oracle's records carry
    as_oracle = [[dict(p, area=p["area_moment"], hull_points=0, hull=None, color_count=0) for p in f] for f in flat_dev]
    r_dev = pbmap.register_planes(as_oracle[0], as_oracle[1], 25, pbmap.ODOMETRY_6DoF, params)
    r_ora = pbmap.register_planes(ora[0], ora[1], 25, pbmap.ODOMETRY_6DoF, params)
    assert r_dev["status"] == r_ora["status"] == 2
    assert r_dev["match"] == r_ora["match"] and len(r_dev["match"]) >= 12
    matched = [flat_dev[0][i] for i in r_dev["match"]]
    assert len([p for p in matched if abs(p["normal"][0]) > 0.95]) >= 6        # x is up in the rig frame: floor / ceiling pieces
    assert len([p for p in matched if abs(p["normal"][0]) < 0.3]) >= 3         # wall pieces
    # Frame360::getPlanes = per sensor the tail of getPlanesSensor (Frame360.h:1034-1068: small / narrow regions never stored, regions of
    # one surface pooled), then groupPlanes + mergePlanes: pieces of one surface seen by neighbouring sensors become single planes
    pooled = [[pbmap.pool_sensor_planes(lst) for lst in f] for f in dev]
    assert all(p["area"] >= 0.12 and p["elongation"] <= 6.0 for f in pooled for lst in f for p in lst)
    assert sum(len(lst) for f in pooled for lst in f) < sum(len(f) for f in flat_dev)
    frames = [pbmap.merge_planes(pbmap.group_planes(f)) for f in pooled]
    assert all(3 <= len(m) < len(f) for m, f in zip(frames, flat_dev))
    rm = pbmap.register_planes(frames[0], frames[1], 25, pbmap.ODOMETRY_6DoF, params)
    print("planes per frame", [len(f) for f in flat_dev], "after getPlanes", [len(m) for m in frames], "matched", len(r_dev["match"]), len(rm["match"]),
          "status", rm["status"])
    assert len(rm["match"]) >= 3 and rm["status"] in (0, 2)
    for i, j in rm["match"].items():
        # pooled planes of the two frames: same direction, offsets within the rig's motion plus what pooling different pieces moves them
        assert float(frames[0][i]["normal"] @ frames[1][j]["normal"]) > 0.99 and abs(frames[0][i]["d"] - frames[1][j]["d"]) < 0.3


def test_frame360_pair_example_runs_on_the_sample_pair(hip_lib, sample, tmp_path):
    """examples/frame360_pair.cpp = RegisterPairRGBD360.cpp's sequence through the Frame360 / Calib360 / RegisterRGBD360 adapters, on the
    sample frames written back into the reference's file formats: it loads, stitches, extracts planes, registers them and runs the
    dense alignment from the plane pose (the identity when the planes leave the translation open)."""
    from tests.test_cpp_adapter import build_frame360_pair
    c1 = sample["c1"]
    exe = build_frame360_pair(tmp_path)
    for idx in (1, 10):
        c1.write_bin(tmp_path / ("sphere_images_%d.bin" % idx), sample["frames"][idx])
    for s in range(8):
        np.savetxt(tmp_path / ("Rt_0%d.txt" % (s + 1)), sample["Rt"][s], fmt="%.17g")
    out = subprocess.check_output([exe, str(tmp_path / "sphere_images_1.bin"), str(tmp_path / "sphere_images_10.bin"), str(tmp_path), "2"],
                                  text=True).strip().splitlines()
    print("\n".join(out))
    c = out[0].replace(",", "").split()
    assert int(c[2]) == 8 * 160 * 120 and int(c[5]) > 500 and int(c[7]) > 0.3 * int(c[5]), out[0]
    w = out[1].split()
    assert int(w[1]) >= 3 and int(w[2]) >= 3 and int(w[4]) > int(w[1]), out[1]
    st = out[2].split()
    assert st[1] in ("0", "2"), out[2]
    assert out[7].split()[2] == "0", out[7]                                    # dense status
    Td = np.array([[float(x) for x in l.split()] for l in out[8:12]])
    assert np.allclose(Td[:3, :3] @ Td[:3, :3].T, np.eye(3), atol=1e-4)
    rot, trans = synth.pose_error(Td, np.eye(4))
    assert 0.02 < trans < 0.6 and rot < 0.1, (rot, trans)


QVGA_K = (262.5, 262.5, 159.5, 119.5)       # 525 * 320 / 640 (RegisterRGBD360.h:357-365), Calib360.h:74-77


@pytest.mark.parametrize("method", [1, 2])
def test_sample_pair_pinhole_alignment_per_sensor(hip_lib, oracle_mod, sample, method):
    """RegisterPhotoICP::alignFrames (RPI.h:4254-4512, the per-sensor use of MethodsRegisterRGBD360.cpp:336-348) on the eight REAL sensor
    image pairs of frames 1 and 10: a narrow field of view on noisy depth, several sensors facing a bare wall -- poorly conditioned,
    levels that stop at the iteration limit or never start.  Per sensor: same status and accept / reject sequence as the oracle in
    device arithmetic, exact per-pass counts at the result, pose within the pinhole tolerance of the synthetic tests scaled by what
    the problem's conditioning does to a 1e-7 rounding difference (measured and printed)."""
    from rgbd360_amd.register import RegisterPhotoICP
    worst = [0.0, 0.0]
    n_marginal = 0
    for s in range(8):
        (rgbA, dA), (rgbB, dB) = sample["frames"][1][s], sample["frames"][10][s]
        reg = RegisterPhotoICP()
        reg.setNumPyr(3)
        reg.setMaskSeams(False)
        reg.setCameraMatrix(QVGA_K)
        reg.setTargetFrame(rgbA, dA)
        reg.setSourceFrame(rgbB, dB)
        rc = reg.alignFrames(np.eye(4), method)
        ora = oracle_mod.Oracle(n_pyr=3, math_mode=1, reduce_mode=1, mask_seams=0)
        ora.set_camera(*QVGA_K)
        ora.set_target(rgbA, dA)
        ora.set_source(rgbB, dB)
        st, pose_ref = ora.align_pinhole(np.eye(4), method)
        iters = list(reg.num_iterations)
        rot, trans = synth.pose_error(reg.getOptimalPose(), pose_ref)
        print(f"sensor {s} method {method}: status {rc}/{st} iters {iters} / {list(ora.result.iters)[:3]} pose diff {rot:.2e} rad {trans:.2e} m")
        assert rc == st
        if iters != list(ora.result.iters)[:3]:
            # A different accept / reject sequence is only tolerated where the oracle's own decision was a coin toss: some step of the
            # diverging level changed the error by less than float32 rounding of the sums (2e-5 relative).  Met on sensor 6, depth only
            # (a bare wall in a 60-degree view): the oracle accepts a step that improves 1.2929930 -> 1.2929868 (5e-6) in device arithmetic,
            # rejects it with float32 accumulators ([10, 7, 0] / [10, 6, 0] / [10, 10, 0] / [10, 7, 0] over its four mode combinations);
            # the device, whose weights come from the hardware's 1-ulp rsq / rcp, lands on [10, 6, 0].
            lvl = next(l for l in range(3) if iters[l] != list(ora.result.iters)[l])
            marginal = [abs(t["error"] - t["new_error"]) / t["error"] for t in ora.trace() if t["level"] == lvl and t["it"] >= 0]
            print(f"   sensor {s}: sequences differ on level {lvl}; smallest relative error change of a step there {min(marginal):.1e}")
            assert min(marginal) < 2e-5, (s, iters, list(ora.result.iters)[:3], marginal)
            n_marginal += 1
            continue
        worst = [max(worst[0], rot), max(worst[1], trans)]
        for level in range(3):
            e = reg.eval_pinhole(level, pose_ref, method)
            _, sp, sd, n_p, n_d = ora.error_pinhole(level, pose_ref, method)
            assert list(e["n_split"]) == [n_p, n_d], (s, level)
    print(f"worst pose difference over the sensors with the oracle's sequence: {worst[0]:.2e} rad {worst[1]:.2e} m; coin-toss sequences: {n_marginal}")
    assert worst[0] <= 1e-4 and worst[1] <= 1e-3 and n_marginal <= 1, (worst, n_marginal)


@pytest.mark.parametrize("method", [0, 2])
def test_sample_pair_rig_dense_registration(hip_lib, oracle_mod, sample, method):
    """RegisterRGBD360::RegisterDensePhotoICP (RegisterRGBD360.h:344-520, the reference's three defects fixed) on the real 8-sensor pair with the
    reference's extrinsics: every level runs to the iteration limit ([10, 10, 10]); same sequence as the oracle, pose within the rig
    tolerance of the synthetic tests, and within the north star's of the reference-faithful arithmetic."""
    from rgbd360_amd.rig import RegisterDensePhotoICP
    Rt = sample["Rt"]
    f1, f10 = sample["frames"][1], sample["frames"][10]
    reg = RegisterDensePhotoICP(Rt, QVGA_K, n_pyr=3)
    reg.setTargetFrame(f1)
    reg.setSourceFrame(f10)
    ok = reg.align(np.eye(4), method)
    poses = {}
    for mm in ((1, 1), (0, 0)):
        rig = oracle_mod.RigOracle(Rt, QVGA_K, n_pyr=3, math_mode=mm[0], reduce_mode=mm[1])
        for s in range(8):
            rig.set_frame(s, True, *f1[s])
            rig.set_frame(s, False, *f10[s])
        st, pose = rig.align(np.eye(4), method)
        poses[mm] = (st, pose, list(rig.iters))
    st, pose_ref, iters_ref = poses[(1, 1)]
    rot, trans = synth.pose_error(reg.getPose(), pose_ref)
    rot0, trans0 = synth.pose_error(reg.getPose(), poses[(0, 0)][1])
    print(f"rig method {method}: ok {ok} iters {reg.num_iterations} / {iters_ref}; vs oracle (device arithmetic) {rot:.2e} rad {trans:.2e} m; vs libm {rot0:.2e} rad {trans0:.2e} m")
    assert ok and st == 0 and reg.num_iterations == iters_ref == [10, 10, 10]
    assert rot <= 5e-5 and trans <= 2e-4, (rot, trans)
    assert rot0 <= ROT_TOL and trans0 <= TRANS_TOL, (rot0, trans0)


@pytest.mark.parametrize("refine", [False, True])
def test_sample_panorama_frame_planes_equal_the_oracle_stages(hip_lib, oracle_mod, panos, refine):
    """Rows a13-a15 on REAL range data: rgbd360_frame_planes on the stitched 1920 x 320 panorama of frame 1 (sensor noise, a fifth of the
    pixels without depth, seams every 240 columns) with the reference's PCL parameters (Frame360.h:949-977) and the panorama's colours
    -- cloud bit-exact, normal map equal to the oracle's (same NaN pattern), region labels and (root, count) lists identical, colour
    descriptors to their integer sums; with segmentAndRefine's refinement the grown labels equal the oracle's raster passes."""
    from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP
    rgb, d = panos[0]
    H, W = d.shape
    st = Frame360Stages(RegisterPhotoICP())
    st.set_color_image(rgb)
    kw = dict(convention=2, max_depth_change_factor=0.02, normal_smoothing_size=8.0, min_inliers=80, angular_threshold=0.0398,
              distance_threshold=0.02, max_curvature=0.0013, depth_mode=1, max_planes=1024)
    plain = st.frame_planes(d, **kw)
    xyz = oracle_mod.sphere_cloud(d, 2)
    assert np.array_equal(np.isnan(xyz), np.isnan(plain["xyz"])) and np.array_equal(np.nan_to_num(xyz), np.nan_to_num(plain["xyz"]))
    nrm, _w = oracle_mod.f360_normals(xyz, H, W, 0.02, 8.0, 1)
    ok = ~np.isnan(nrm[:, 0])
    assert np.array_equal(np.isnan(plain["normals"][:, 0]), ~ok) and 0.2 < ok.mean() < 0.9
    assert np.abs(plain["normals"][ok] - nrm[ok]).max() <= 1.2e-7 and (plain["normals"][ok] == nrm[ok]).mean() > 0.9999
    labels, planes = oracle_mod.f360_plane_segment(xyz, plain["normals"], H, W, 80, 0.0398, 0.02, 0.0013, 1, max_planes=1024)
    assert np.array_equal(plain["labels"], labels)
    got_rc, want_rc = {(p["root"], p["count"]): p for p in plain["planes"]}, {(p["root"], p["count"]): p for p in planes}
    only = [(k, float(v["curvature"])) for k, v in got_rc.items() if k not in want_rc] + [(k, float(v["curvature"])) for k, v in want_rc.items() if k not in got_rc]
    print("regions on one side only (root, count), curvature:", only)
    # a region whose curvature sits on the filter's threshold (0.0013) to the last digits may pass on one side only: the device sums its
    # moments in 2^-24 m fixed point, the checker in floating point
    assert all(abs(c - 0.0013) < 2e-6 for _, c in only) and len(only) <= 2, only
    assert len(planes) >= 20
    out = plain
    if refine:
        st.set_refinement(True, 0.02)
        out = st.frame_planes(d, **kw)
        labels_ref, planes_ref, changed = oracle_mod.f360_plane_refine(xyz, H, W, plain["labels"], plain["planes"], 0.02)
        assert changed > 1000 and st.refinement_stats()["pixels_relabelled"] == changed
        assert np.array_equal(out["labels"], labels_ref)
        assert [(p["root"], p["count"]) for p in out["planes"]] == [(p["root"], p["count"]) for p in planes_ref]
    _, want = oracle_mod.f360_plane_colour(out["labels"], rgb, out["planes"])
    modes = oracle_mod.f360_plane_colour_mode(out["labels"], rgb, out["planes"])
    for p, w, m in zip(out["planes"], want, modes):
        assert p["color_count"] == w["color_count"] > 0
        assert np.abs(p["color_nrgb"] - w["color_nrgb"]).max() <= 1e-7 and np.abs(p["hist_h"] - w["hist_h"]).max() <= 1e-7
        assert p["color_mode_count"] == m["color_mode_count"] and np.array_equal(p["color_mode"], m["color_mode"])
    print(f"real panorama: {ok.mean():.2f} of the pixels carry a normal, {len(out['planes'])} planes, largest {max(p['count'] for p in out['planes'])} px"
          + (f", refinement grew {changed} px" if refine else ""))
