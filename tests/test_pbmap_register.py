"""RegisterRGBD360::RegisterPbMap (reference include/RegisterRGBD360.h:110-338): the host-side plane
matching + closed-form pose of the HIP library (`rgbd360_register_planes`, pure host C++ -- runs without a GPU) against
analytic known answers and against the independent numpy restatement oracle/pbmap_ref.py.  PARITY UNPINNED: the matcher
and the pose fit are MRPT code outside the reference tree (see the oracle's header)."""
import math
import os

import numpy as np
import pytest

from oracle import pbmap_ref as O
from rgbd360_amd import pbmap, synth

ROOM_LO, ROOM_HI = np.asarray(synth.ROOM_LO, float), np.asarray(synth.ROOM_HI, float)
# A box is symmetric: without odometry constraints a flipped interpretation explains it equally well (true of the
# reference too).  The visible part of every second wall is therefore made much smaller, which the unary area
# constraint (ratio 4 / 3) tells apart.
VISIBLE = [1.0, 0.2, 1.0, 0.22, 1.0, 0.18]


def room_planes(T_wc, rng=None, noise_n=0.0, noise_d=0.0):
    """The six walls of the synthetic room as seen from camera pose T_wc (camera -> world): analytic plane records."""
    R, c = T_wc[:3, :3], T_wc[:3, 3]
    size = ROOM_HI - ROOM_LO
    planes = []
    for ax in range(3):
        for sgn, bound in ((-1.0, ROOM_HI[ax]), (1.0, ROOM_LO[ax])):
            n_w = np.zeros(3)
            n_w[ax] = sgn                                   # towards the inside of the room
            centre_w = 0.5 * (ROOM_LO + ROOM_HI)
            centre_w[ax] = bound
            n_c = R.T @ n_w
            if rng is not None and noise_n > 0:
                n_c = n_c + rng.normal(size=3) * noise_n
                n_c /= np.linalg.norm(n_c)
            cen_c = R.T @ (centre_w - c)
            d = -float(n_c @ cen_c) + (rng.normal() * noise_d if rng is not None and noise_d > 0 else 0.0)
            dims = np.delete(size, ax)
            ppal_w = np.zeros(3)
            ppal_w[[k for k in range(3) if k != ax][int(np.argmax(dims))]] = 1.0
            planes.append(dict(centroid=cen_c.astype(np.float32), normal=n_c.astype(np.float32), d=np.float32(d),
                               curvature=np.float32(1e-5), count=1000, root=len(planes), area=np.float32(dims[0] * dims[1] * VISIBLE[len(planes)]),
                               elongation=np.float32(dims.max() / dims.min()), ppal_dir=(R.T @ ppal_w).astype(np.float32)))
    return planes


def clutter_plane(rng, root):
    n = rng.normal(size=3)
    n /= np.linalg.norm(n)
    cen = rng.uniform(-2, 2, size=3)
    if n @ cen > 0:
        n = -n
    return dict(centroid=cen.astype(np.float32), normal=n.astype(np.float32), d=np.float32(-n @ cen), curvature=np.float32(2e-4),
                count=200, root=root, area=np.float32(rng.uniform(0.2, 6.0)), elongation=np.float32(rng.uniform(1.0, 4.0)),
                ppal_dir=np.array([1, 0, 0], np.float32))


def motion(rng, trans, rot_deg):
    ax = rng.normal(size=3)
    t = rng.normal(size=3)
    return synth.make_pose(synth.rodrigues(ax, math.radians(rot_deg)), t / np.linalg.norm(t) * trans)


def params_dict(p):
    return {name: getattr(p, name) for name, _ in p._fields_ if name != "max_nodes"}


def test_default_params_are_the_reference_ini_values():
    for odo, fname in ((False, "configLocaliser_spherical.ini"), (True, "configLocaliser_sphericalOdometry.ini")):
        p, q = pbmap.default_params(odo), O.default_params(odo)
        for k, v in q.items():
            assert abs(getattr(p, k) - v) < 1e-6 * max(1.0, abs(v)), k
        path = os.path.join("/root/reference/config_files", fname)
        if not os.path.exists(path):          # the GPU box has no reference tree
            continue
        ini = {}
        for line in open(path):
            line = line.split("//")[0].strip()
            if "=" in line and not line.startswith("%"):
                k, v = line.split("=", 1)
                ini[k.strip()] = v.strip()
        for ours, theirs in (("dist_d", "dist_d"), ("angle_deg", "angle"), ("elongation_threshold", "elongation_threshold"),
                             ("area_threshold", "area_threshold"), ("dist_threshold", "dist_threshold"),
                             ("angle_threshold_deg", "angle_threshold"), ("height_threshold", "height_threshold"),
                             ("cos_normal_threshold", "cos_normal_threshold"), ("min_planes_recognition", "min_planes_recognition"),
                             ("color_threshold", "color_threshold"), ("intensity_threshold", "intensity_threshold")):
            assert abs(getattr(p, ours) - float(ini[theirs])) < 1e-6, (fname, ours)
        assert p.use_color == 1 and p.hue_threshold == 0.0 and 0.3 < float(ini["hue_threshold"]) < 0.5      # the hue test is opt-in (Frame360.h:673)


@pytest.mark.parametrize("mode,trans,rot", [(O.DEFAULT_6DoF, 0.8, 35.0), (O.ODOMETRY_6DoF, 0.06, 2.0), (O.DEFAULT_6DoF, 0.0, 0.0)])
def test_box_room_motion_is_recovered_exactly(mode, trans, rot):
    """Known answer: noiseless walls seen from two poses, target list shuffled -> all six matched, pose = ground truth."""
    rng = np.random.default_rng(11)
    for trial in range(5):
        T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
        T_wB = T_wA @ motion(rng, trans, rot)
        ref, trg = room_planes(T_wA), room_planes(T_wB)
        perm = rng.permutation(6)
        trg = [trg[k] for k in perm]
        r = pbmap.register_planes(ref, trg, 0, mode)
        assert r["status"] == 0
        assert r["match"] == {i: int(np.where(perm == i)[0][0]) for i in range(6)}
        rot_err, tr_err = synth.pose_error(r["pose"], np.linalg.inv(T_wA) @ T_wB)
        assert rot_err < 2e-6 and tr_err < 5e-6, (rot_err, tr_err)
        assert abs(r["area_matched"] - sum(float(p["area"]) for p in ref)) < 1e-3
        ev = np.linalg.eigvalsh(r["info"].astype(np.float64))
        assert ev.min() > 0 and np.allclose(r["info"], r["info"].T)


@pytest.mark.parametrize("mode", [O.DEFAULT_6DoF, O.PLANAR_3DoF, O.ODOMETRY_6DoF, O.PLANAR_ODOMETRY_3DoF])
def test_matches_numpy_restatement_on_noisy_cluttered_scenes(mode):
    """Branch-and-bound + Jacobi/cross-product pose fit (library) vs plain enumeration + numpy SVD / lstsq (oracle):
    same status, same interpretation, same pose and information matrix."""
    rng = np.random.default_rng(100 + mode)
    odo = mode in (O.ODOMETRY_6DoF, O.PLANAR_ODOMETRY_3DoF)
    statuses = set()
    for trial in range(24):
        T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
        if mode in (O.PLANAR_3DoF, O.PLANAR_ODOMETRY_3DoF):      # rotation about the up axis x, translation on the floor
            ang = math.radians(3.0 if odo else rng.uniform(-60, 60))
            t = np.array([0.0, *rng.normal(size=2)])
            M = synth.make_pose(synth.rodrigues(np.array([1.0, 0, 0]), ang), t / np.linalg.norm(t) * (0.08 if odo else 0.7))
        else:
            M = motion(rng, 0.08 if odo else rng.uniform(0.1, 1.0), 3.0 if odo else rng.uniform(0, 50))
        T_wB = T_wA @ M
        ref = room_planes(T_wA, rng, 0.004, 0.004)
        trg = room_planes(T_wB, rng, 0.004, 0.004)
        if trial % 3 == 1:                                       # a wall not seen in one of the frames
            ref.pop(int(rng.integers(6)))
        if trial % 4 == 2:                                       # only two wall directions left: translation unobservable
            ref = [p for p in ref if abs(p["normal"][0]) < 0.5]
        for k in range(int(rng.integers(0, 4))):
            ref.append(clutter_plane(rng, 100 + k))
        for k in range(int(rng.integers(0, 4))):
            trg.append(clutter_plane(rng, 200 + k))
        trg = [trg[k] for k in rng.permutation(len(trg))]
        p = pbmap.default_params(odo)
        got = pbmap.register_planes(ref, trg, 0, mode, p)
        want = O.register_planes(ref, trg, 0, mode, params_dict(p))
        statuses.add(want["status"])
        assert got["status"] == want["status"], trial
        assert got["match"] == want["match"], trial
        assert abs(got["area_matched"] - want["area_matched"]) < 1e-4 * max(1.0, want["area_matched"])
        if want["status"] == 0:
            rot_err, tr_err = synth.pose_error(got["pose"], want["pose"])
            assert rot_err < 2e-6 and tr_err < 5e-6, (trial, rot_err, tr_err)
            assert np.abs(got["info"] - want["info"]).max() < 1e-5 * np.abs(want["info"]).max()
            gt_rot, gt_tr = synth.pose_error(got["pose"], np.linalg.inv(T_wA) @ T_wB)
            assert gt_rot < math.radians(1.0) and gt_tr < 0.03, (trial, gt_rot, gt_tr)
        else:
            assert np.array_equal(got["pose"], np.eye(4, dtype=np.float32))
    assert 0 in statuses and len(statuses) > 1            # the scenario mix exercises a failure path too


def test_insufficient_and_unobservable_cases():
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ synth.default_motion(3, 0.3, 10.0)
    ref, trg = room_planes(T_wA), room_planes(T_wB)
    r = pbmap.register_planes(ref[:2], trg, 0, O.DEFAULT_6DoF)              # two planes: "Insuficient matching" (:312)
    assert r["status"] == 1 and len(r["match"]) == 2 and np.array_equal(r["pose"], np.eye(4, dtype=np.float32))
    r = pbmap.register_planes([], trg, 0, O.DEFAULT_6DoF)
    assert r["status"] == 1 and r["match"] == {} and r["area_matched"] == 0.0
    par = [p for p in ref if abs(p["normal"][0]) < 0.5]                      # four walls, no floor / ceiling
    r = pbmap.register_planes(par, trg, 0, O.DEFAULT_6DoF)
    assert r["status"] == 2 and len(r["match"]) == 4                         # matched, but x translation is unobservable
    with pytest.raises(ValueError):
        pbmap.register_planes(ref, trg, 0, 7)


def test_subgraph_selection_and_filters():
    """setReference / setTarget (:110-195): curved planes never match; max_match_planes keeps the largest areas; the
    Frame360.h:1034,1041 filters drop small and narrow planes."""
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ synth.default_motion(4, 0.2, 5.0)
    ref, trg = room_planes(T_wA), room_planes(T_wB)
    ref[0]["curvature"] = np.float32(0.01)
    r = pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF)
    assert r["status"] == 0 and 0 not in r["match"] and len(r["match"]) == 5
    ref[0]["curvature"] = np.float32(1e-5)
    r = pbmap.register_planes(ref, trg, 4, O.DEFAULT_6DoF)                  # the four largest areas survive
    areas = sorted(float(ref[i]["area"]) for i in r["match"])
    assert r["status"] == 0 and areas == sorted(float(p["area"]) for p in ref)[-4:]
    assert O.register_planes(ref, trg, 4, O.DEFAULT_6DoF)["match"] == r["match"]
    ref[2]["area"] = np.float32(0.05)
    ref[3]["elongation"] = np.float32(9.0)
    r = pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF)
    assert 2 not in r["match"] and 3 not in r["match"]


def test_planar_mode_rejects_a_tilted_interpretation():
    """PLANAR_3DoF: a motion that tilts the rig cannot be explained; the unconstrained mode recovers it."""
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    tilt = synth.make_pose(synth.rodrigues(np.array([0, 1.0, 0]), math.radians(25.0)), np.array([0, 0.1, 0.1]))
    ref, trg = room_planes(T_wA), room_planes(T_wA @ tilt)
    assert pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF)["status"] == 0
    r = pbmap.register_planes(ref, trg, 0, O.PLANAR_3DoF)
    assert len(r["match"]) < 6
    assert r["match"] == O.register_planes(ref, trg, 0, O.PLANAR_3DoF)["match"]


def test_reference_class_surface():
    """The RegisterRGBD360 mirror: lazy registration in the getters (:198-256), entropy formula (:229-238)."""
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ synth.default_motion(9, 0.06, 2.0)
    reg = pbmap.RegisterRGBD360(odometry_config=True)
    reg.setReference(room_planes(T_wA))
    reg.setTarget(room_planes(T_wB))
    pose = reg.getPose()                                                     # triggers RegisterPbMap
    assert synth.pose_error(pose, np.linalg.inv(T_wA) @ T_wB)[0] < 2e-6
    assert len(reg.getMatchedPlanes()) == 6 and reg.getAreaMatched() > 0
    cov = reg.getCovMat().astype(np.float64)
    assert np.allclose(cov @ reg.getInfoMat().astype(np.float64), np.eye(6), atol=1e-3)
    want = 0.5 * (6 * (1 + math.log(2 * math.pi)) + math.log(np.linalg.det(np.linalg.inv(reg.getInfoMat().astype(np.float64)))))
    assert abs(reg.calcEntropy() - want) < 1e-9
    total = sum(float(p["area"]) for p in room_planes(T_wA))
    assert abs(reg.areaSource - total) < 1e-3 and abs(reg.areaTarget - total) < 1e-3          # every wall entered the matching
    assert reg.RegisterPbMap(room_planes(T_wA), room_planes(T_wB), 2, pbmap.ODOMETRY_6DoF) is False      # two planes: insufficient
    assert reg.RegisterPbMap(room_planes(T_wA), room_planes(T_wB), 4, pbmap.ODOMETRY_6DoF)
    assert abs(reg.areaSource - sum(sorted(float(p["area"]) for p in room_planes(T_wA))[-4:])) < 1e-3


def test_golden_plane_lists():
    """tests/golden/pbmap_planes.json (planes the oracle extracted from two rendered frames + recorded registrations):
    the library and the numpy restatement both reproduce every recorded run."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pbmap_planes.json")))
    ref, trg = g["frames"]
    n_good = 0
    for key, want in g["runs"].items():
        mode, mmp = int(key[4]), int(key.split("max")[1])
        for r in (pbmap.register_planes(ref, trg, mmp, mode), O.register_planes(ref, trg, mmp, mode)):
            assert r["status"] == want["status"], key
            assert {str(k): v for k, v in r["match"].items()} == want["match"], key
            assert abs(r["area_matched"] - want["area_matched"]) < 1e-4
            if want["status"] == 0:
                rot, tr = synth.pose_error(r["pose"], np.array(want["pose"]))
                assert rot < 2e-6 and tr < 5e-6, (key, rot, tr)
                assert np.abs(np.asarray(r["info"]) - np.array(want["info"])).max() < 1e-5 * np.abs(want["info"]).max()
        if want["status"] == 0:
            n_good += 1
            rot, tr = synth.pose_error(np.array(want["pose"]), np.array(g["T_gt"]))
            assert rot < math.radians(0.1) and tr < 0.01, key        # planes alone land within 0.1 degree / 1 cm
    assert n_good >= 2


def test_hostile_inputs_and_sanitizers(tmp_path):
    """Records with non-finite or non-unit fields never enter the matching; the host matcher / pose fit (the very header the
    library compiles) runs seeded random + hostile plane lists clean under AddressSanitizer + UBSan (tools/pbmap_fuzz.cpp)."""
    import subprocess
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ synth.default_motion(4, 0.06, 2.0)
    ref, trg = room_planes(T_wA), room_planes(T_wB)
    ref[1]["d"] = np.float32(np.inf)
    ref[4]["normal"] = np.array([np.nan, 0, 1], np.float32)
    trg[2]["normal"] = np.array([0, 0, 3], np.float32)
    got = pbmap.register_planes(ref, trg, 0, O.ODOMETRY_6DoF)
    want = O.register_planes(ref, trg, 0, O.ODOMETRY_6DoF)
    assert got["status"] == want["status"] and got["match"] == want["match"]
    assert 1 not in got["match"] and 4 not in got["match"] and 2 not in got["match"].values()
    assert np.isfinite(got["pose"]).all()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pbmap_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "pbmap_fuzz.cpp"), "-o", exe])
    out = subprocess.run([exe, "1500"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok 1500 rounds"), out.stdout + out.stderr


def test_mirror_images_are_never_matched():
    """Three mutually orthogonal planes against their mirror image: every angle / distance constraint holds, but no rigid
    motion maps one onto the other -- the orientation test of the normal triple leaves one plane unmatched."""
    def corner(sign_z):
        planes = []
        for ax, (a, e) in enumerate(((4.0, 1.5), (6.0, 2.0), (9.0, 1.2))):
            n = np.zeros(3, np.float32)
            n[ax] = 1.0 if ax < 2 else sign_z
            c = np.array([0.3, 0.4, 0.5 * sign_z], np.float32) * 0 + np.float32(-2.0 - ax) * n
            planes.append(dict(centroid=c, normal=n, d=np.float32(-n @ c), curvature=np.float32(1e-5), count=500, root=ax,
                               area=np.float32(a), elongation=np.float32(e), ppal_dir=np.array([0, 0, 0], np.float32)))
        return planes
    same = pbmap.register_planes(corner(1.0), corner(1.0), 0, O.DEFAULT_6DoF)
    assert same["status"] == 0 and same["match"] == {0: 0, 1: 1, 2: 2}
    assert np.allclose(same["pose"], np.eye(4), atol=1e-6)
    mirrored = pbmap.register_planes(corner(1.0), corner(-1.0), 0, O.DEFAULT_6DoF)
    assert mirrored["status"] == 1 and len(mirrored["match"]) == 2
    assert O.register_planes(corner(1.0), corner(-1.0), 0, O.DEFAULT_6DoF)["match"] == mirrored["match"]


def _rect_plane(centre, normal, pdir, a, b, count, root):
    """Plane record of a uniformly sampled 2a x 2b rectangle."""
    n = np.asarray(normal, float) / np.linalg.norm(normal)
    p = np.asarray(pdir, float)
    p = p - (p @ n) * n
    p /= np.linalg.norm(p)
    c = np.asarray(centre, float)
    if n @ c > 0:
        n = -n
    return dict(centroid=c.astype(np.float32), normal=n.astype(np.float32), d=np.float32(-n @ c), curvature=np.float32(1e-6), count=count,
                root=root, area=np.float32(4 * a * b), elongation=np.float32(max(a, b) / min(a, b)), ppal_dir=(p if a >= b else np.cross(n, p)).astype(np.float32))


def test_merge_planes_pools_the_pieces_of_one_surface():
    """Frame360::mergePlanes (Frame360.h:655-733): two halves of a wall become the wall (exact pooled moments), a parallel wall
    0.5 m behind and a co-planar patch 1 m away stay separate, curved regions are never merged; library = numpy restatement."""
    n, pd = np.array([0.0, 0, -1.0]), np.array([1.0, 0, 0])
    left = _rect_plane([-1.0, 0.2, 2.5], n, pd, 1.0, 0.75, 2000, 10)
    right = _rect_plane([1.0, 0.2, 2.5], n, pd, 1.0, 0.75, 2000, 20)
    whole = _rect_plane([0.0, 0.2, 2.5], n, pd, 2.0, 0.75, 4000, 10)
    behind = _rect_plane([0.0, 0.2, 3.0], n, pd, 2.0, 0.75, 3000, 30)
    far = _rect_plane([4.5, 0.2, 2.5], n, pd, 0.5, 0.5, 500, 40)            # same plane, 1.5 m beyond the right half's edge
    curved = dict(_rect_plane([-1.0, 0.2, 2.5], n, pd, 1.0, 0.75, 2000, 50), curvature=np.float32(0.01))
    got = pbmap.merge_planes([left, right, behind, far, curved])
    want = O.merge_planes([left, right, behind, far, curved])
    assert len(got) == len(want) == 4
    m = got[0]
    assert m["count"] == 4000 and m["root"] == 10
    assert np.allclose(m["centroid"], whole["centroid"], atol=1e-5) and abs(m["area"] - whole["area"]) < 1e-3 * whole["area"]
    assert abs(m["elongation"] - whole["elongation"]) < 1e-3 and abs(abs(m["ppal_dir"] @ pd) - 1) < 1e-5
    assert float(m["normal"] @ whole["normal"]) > 1 - 1e-6 and abs(m["d"] - whole["d"]) < 1e-5
    for a, b in zip(got, want):
        assert (a["count"], a["root"]) == (b["count"], b["root"])
        assert np.allclose(a["centroid"], b["centroid"], atol=1e-5) and abs(a["area"] - b["area"]) <= 1e-4 * max(b["area"], 1.0)
        assert float(a["normal"] @ b["normal"]) > 1 - 1e-6
    # three pieces in a row merge transitively (plane j is re-evaluated after every merge, Frame360.h:727-731), whatever their order
    third = _rect_plane([3.0, 0.2, 2.5], n, pd, 1.0, 0.75, 2000, 60)
    for order in ([left, third, right], [third, left, right], [right, left, third]):
        out = pbmap.merge_planes(order)
        assert len(out) == 1 and out[0]["count"] == 6000 and abs(out[0]["area"] - 9.0) < 1e-2
        assert len(O.merge_planes(order)) == 1
    assert pbmap.merge_planes([]) == []
    sliver = _rect_plane([0.0, 0.2, 2.5], n, pd, 2.0, 0.1, 300, 70)          # elongation 20: never stored by Frame360.h:1041
    assert len(pbmap.merge_planes([left, sliver])) == 1 and len(O.merge_planes([left, sliver])) == 1
    assert len(pbmap.merge_planes([left, sliver], max_elongation=100.0)) == 1   # (kept, and merged into the wall it lies on)
    assert pbmap.merge_planes([left, sliver], max_elongation=100.0)[0]["count"] == 2300


def _poly_plane(poly_uv, origin, e1, e2, count, root, with_hull=True):
    """Plane record of a uniformly sampled CONVEX polygon (vertices counter-clockwise in the (e1, e2) frame of the plane through `origin`):
    exact-enough moments from a dense grid, the polygon itself as the record's hull."""
    e1, e2, origin = np.asarray(e1, float), np.asarray(e2, float), np.asarray(origin, float)
    n = np.cross(e1, e2)
    P = np.asarray(poly_uv, float)
    lo, hi = P.min(0), P.max(0)
    g = np.stack(np.meshgrid(np.linspace(lo[0], hi[0], 601), np.linspace(lo[1], hi[1], 601), indexing="ij"), -1).reshape(-1, 2)
    inside = np.ones(len(g), bool)
    for a, b in zip(P, np.roll(P, -1, 0)):
        inside &= (b[0] - a[0]) * (g[:, 1] - a[1]) - (b[1] - a[1]) * (g[:, 0] - a[0]) >= 0
    q = g[inside]
    c2 = q.mean(0)
    C = np.cov((q - c2).T, bias=True)
    w, V = np.linalg.eigh(C)
    l1, l2 = w
    c = origin + c2[0] * e1 + c2[1] * e2
    flip = n @ c > 0                                         # the normal points towards the origin
    nn = -n if flip else n
    pd = V[0, 1] * e1 + V[1, 1] * e2
    x, y = P[:, 0], P[:, 1]
    cr = x * np.roll(y, -1) - np.roll(x, -1) * y
    area = abs(cr.sum()) / 2
    cu, cv = ((x + np.roll(x, -1)) * cr).sum() / (3 * cr.sum()), ((y + np.roll(y, -1)) * cr).sum() / (3 * cr.sum())
    hull = np.array([origin + u * e1 + v * e2 for u, v in (P[::-1] if flip else P)], np.float32)
    rec = dict(centroid=c.astype(np.float32), normal=nn.astype(np.float32), d=np.float32(-nn @ c), curvature=np.float32(1e-6), count=count, root=root,
               area=np.float32(area), area_moment=np.float32(12 * np.sqrt(l1 * l2)), elongation=np.float32(np.sqrt(l2 / l1)), ppal_dir=pd.astype(np.float32),
               center_hull=(origin + cu * e1 + cv * e2).astype(np.float32), hull_points=len(P), hull=hull)
    if not with_hull:                                        # a caller-made record: moments only
        rec.update(area=rec["area_moment"], hull_points=0, center_hull=rec["centroid"])
        del rec["hull"]
    return rec


def test_merge_planes_tests_proximity_on_the_hull_polygons():
    """Frame360::mergePlanes compares the planes' hull POLYGONS vertex by vertex and edge by edge (Frame360.h:680-711; mergePlane2 then hulls
    the two contours again).  (1) The two bars of an L-shaped wall merge, and the merged record carries the convex hull of the L;
    (2) a patch that pokes into the MOMENT rectangle of a large triangular region but stays 0.33 m clear of the triangle itself is
    merged when the records carry moments only, and kept apart when they carry their polygons; (3) two patches 0.25 m apart along an
    edge merge through the edge-to-edge distance although no two of their vertices are within 0.3 m."""
    o, e1, e2 = np.array([-1.0, -1.5, 3.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    bar_a = _poly_plane([(0, 0), (2, 0), (2, 1), (0, 1)], o, e1, e2, 2000, 5)
    bar_b = _poly_plane([(0, 1), (1, 1), (1, 3), (0, 3)], o, e1, e2, 2000, 9)
    out = pbmap.merge_planes([bar_a, bar_b])
    assert len(out) == 1 and out[0]["count"] == 4000 and out[0]["root"] == 5
    hull_l = np.array([(0, 0), (2, 0), (2, 1), (1, 3), (0, 3)], float)          # the convex hull of the L
    x, y = hull_l[:, 0], hull_l[:, 1]
    area_l = abs((x * np.roll(y, -1) - np.roll(x, -1) * y).sum()) / 2
    assert abs(out[0]["area"] - area_l) < 1e-4 * area_l and len(out[0]["hull"]) == 5
    got = {tuple(np.round([(v - o) @ e1, (v - o) @ e2], 4)) for v in out[0]["hull"].astype(float)}
    assert got == {tuple(np.round(v, 4)) for v in hull_l}
    n_out = out[0]["normal"].astype(float)                                         # counter-clockwise seen from the normal's side
    hv = out[0]["hull"].astype(float)
    assert all(n_out @ np.cross(hv[(i + 1) % 5] - hv[i], hv[(i + 2) % 5] - hv[(i + 1) % 5]) > 0 for i in range(5))
    # (2) moments say "overlap", polygons say "0.33 m apart"
    tri = [(0, 0), (9, 0), (0, 9)]
    k = (4.5 + 0.33 / np.sqrt(2) + 0.01)
    patch = [(k, k), (k + 0.5, k), (k + 0.5, k + 0.5), (k, k + 0.5)]
    for with_hull, want in ((False, 1), (True, 2)):
        t = _poly_plane(tri, o, e1, e2, 8000, 1, with_hull)
        q = _poly_plane(patch, o, e1, e2, 300, 2, with_hull)
        assert len(pbmap.merge_planes([t, q])) == want, with_hull
    # (3) edge against edge: 4 m long neighbours, offset so that every vertex pair is > 0.3 m apart
    long_a = _poly_plane([(0, 0), (4, 0), (4, 1), (0, 1)], o, e1, e2, 3000, 3)
    long_b = _poly_plane([(1.5, 1.25), (2.5, 1.25), (2.5, 2.25), (1.5, 2.25)], o, e1, e2, 800, 4)
    assert len(pbmap.merge_planes([long_a, long_b])) == 1
    far_b = _poly_plane([(1.5, 1.35), (2.5, 1.35), (2.5, 2.35), (1.5, 2.35)], o, e1, e2, 800, 4)      # 0.35 m: stays apart
    assert len(pbmap.merge_planes([long_a, far_b])) == 2


def test_merge_planes_random_sets_match_the_numpy_restatement():
    rng = np.random.default_rng(8)
    for trial in range(20):
        planes = []
        for w in range(int(rng.integers(1, 4))):                               # a few walls, each cut into 1-4 pieces + noise
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            pdv = np.cross(nrm, rng.normal(size=3))
            dist = rng.uniform(1.5, 4.0)
            pieces = int(rng.integers(1, 5))
            for k in range(pieces):
                centre = -nrm * dist + pdv / np.linalg.norm(pdv) * (k - pieces / 2) * 1.9 + rng.normal(size=3) * 0.005
                planes.append(_rect_plane(centre, nrm + rng.normal(size=3) * 0.004, pdv, 1.0, rng.uniform(0.4, 1.0), int(rng.integers(200, 3000)),
                                          len(planes)))
        planes = [planes[i] for i in rng.permutation(len(planes))]
        got, want = pbmap.merge_planes(planes), O.merge_planes(planes)
        assert [(p["count"], p["root"]) for p in got] == [(p["count"], p["root"]) for p in want], trial
        for a, b in zip(got, want):
            assert np.allclose(a["centroid"], b["centroid"], atol=2e-5) and abs(a["area"] - b["area"]) <= 2e-4 * max(b["area"], 1.0)


# ---- colour descriptors and the radiometric unary constraint (Frame360.h:1045-1046; configLocaliser_spherical.ini:19-21) ----
WALL_RGB = [(200, 60, 50), (60, 180, 70), (50, 80, 200), (190, 180, 60), (170, 70, 180), (90, 190, 190)]


def colour_fields(rgb, brightness=1.0, rng=None, noise=0.0):
    """The colour record of a region of (nearly) one colour, as the device derives it: normalised colour, intensity, hue histogram."""
    c = np.clip(np.asarray(rgb, float) * brightness, 0, 255)
    if rng is not None and noise > 0:
        c = np.clip(c + rng.normal(size=3) * noise, 1, 255)
    s = c.sum()
    mx, mn = c.max(), c.min()
    hist = np.zeros(74, np.float32)
    if mx * 5 <= 255:
        hist[72] = 1
    elif (mx - mn) * 5 <= mx:
        hist[73] = 1
    else:
        k = int(np.argmax(c))
        x, y = c[(k + 1) % 3], c[(k + 2) % 3]
        h12 = (24 * k + 12 * (x - y) / (mx - mn)) % 72
        hist[int(h12) % 72] = 1
    return dict(color_count=1000, color_nrgb=(c / s).astype(np.float32), color_dev=np.full(3, 0.004, np.float32),
                intensity=np.float32(s), hist_h=hist)


def coloured(planes, colours, **kw):
    return [dict(p, **colour_fields(colours[k % len(colours)], **kw)) for k, p in enumerate(planes)]


def test_colour_constraint_rejects_same_shape_planes_of_another_colour():
    """Two frames of the room whose walls differ in nothing but their colour assignment: geometry alone matches all six walls;
    with the colour records, a wall whose colour changed is left unmatched, and the walls that kept theirs still fix the pose."""
    rng = np.random.default_rng(3)
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ motion(rng, 0.3, 12.0)
    ref = coloured(room_planes(T_wA), WALL_RGB)
    repainted = list(WALL_RGB)
    repainted[2] = (60, 200, 60)                      # wall 2 was blue, is green now
    trg = coloured(room_planes(T_wB), repainted)
    r = pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF)
    assert r["status"] == 0 and r["match"] == {0: 0, 1: 1, 3: 3, 4: 4, 5: 5}
    rot_err, tr_err = synth.pose_error(r["pose"], np.linalg.inv(T_wA) @ T_wB)
    assert rot_err < 2e-6 and tr_err < 5e-6
    p = pbmap.default_params(False)
    p.use_color = 0                                   # switched off: the repainted wall matches on its shape again
    assert pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF, p)["match"] == {k: k for k in range(6)}
    # planes WITHOUT colour (color_count 0) are never rejected for it
    bare = room_planes(T_wB)
    assert pbmap.register_planes(ref, bare, 0, O.DEFAULT_6DoF)["match"] == {k: k for k in range(6)}


def test_colour_constraint_survives_a_global_brightness_change():
    """The same scene 25 % darker / brighter: the normalised colour does not move, the intensity stays inside its loose bound
    (150 of 765) -- every wall still matches.  Only a change large enough to break the intensity bound (x 2) unmatches the bright walls."""
    rng = np.random.default_rng(4)
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ motion(rng, 0.2, 8.0)
    ref = coloured(room_planes(T_wA), WALL_RGB, brightness=0.6)
    for b in (0.45, 0.75):
        trg = coloured(room_planes(T_wB), WALL_RGB, brightness=b, rng=rng, noise=1.0)
        r = pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF)
        assert r["status"] == 0 and r["match"] == {k: k for k in range(6)}, b
    p = pbmap.default_params(False)
    trg = coloured(room_planes(T_wB), WALL_RGB, brightness=1.2)             # twice as bright: sums differ by 180 - 290
    assert len(pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF, p)["match"]) < 6
    p.intensity_threshold = 0.0                                            # bound off: normalised colour alone
    assert pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF, p)["match"] == {k: k for k in range(6)}


def test_hue_histogram_constraint_is_opt_in():
    rng = np.random.default_rng(5)
    T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    T_wB = T_wA @ motion(rng, 0.2, 8.0)
    ref = coloured(room_planes(T_wA), WALL_RGB)
    trg = coloured(room_planes(T_wB), WALL_RGB)
    for q in trg:                                     # same mean colour, but the histogram says "unsaturated"
        q["hist_h"] = np.zeros(74, np.float32)
        q["hist_h"][73] = 1
    p = pbmap.default_params(False)
    assert pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF, p)["match"] == {k: k for k in range(6)}
    p.hue_threshold = 0.45                            # configLocaliser_spherical.ini:21
    assert pbmap.register_planes(ref, trg, 0, O.DEFAULT_6DoF, p)["match"] == {}
    assert O.register_planes(ref, trg, 0, O.DEFAULT_6DoF, params_dict(p))["match"] == {}


@pytest.mark.parametrize("mode", [O.DEFAULT_6DoF, O.ODOMETRY_6DoF])
def test_colour_matches_numpy_restatement_on_random_scenes(mode):
    """Library vs the independent numpy restatement with colour records on every plane: colours drawn so that some true pairs are
    within the thresholds, some are not, and clutter planes carry random colours; hue test on in half of the trials."""
    rng = np.random.default_rng(300 + mode)
    odo = mode == O.ODOMETRY_6DoF
    fewer = 0
    for trial in range(20):
        T_wA = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
        T_wB = T_wA @ motion(rng, 0.08 if odo else rng.uniform(0.1, 0.8), 3.0 if odo else rng.uniform(0, 40))
        cols_a = [tuple(rng.integers(20, 236, size=3)) for _ in range(6)]
        cols_b = [tuple(np.clip(np.asarray(c) + rng.normal(size=3) * (4 if rng.random() < 0.7 else 60), 1, 255)) for c in cols_a]
        ref = coloured(room_planes(T_wA, rng, 0.003, 0.003), cols_a)
        trg = coloured(room_planes(T_wB, rng, 0.003, 0.003), cols_b, brightness=rng.uniform(0.8, 1.2))
        for k in range(int(rng.integers(0, 3))):
            ref.append(dict(clutter_plane(rng, 100 + k), **colour_fields(tuple(rng.integers(0, 256, size=3)))))
            trg.append(dict(clutter_plane(rng, 200 + k), **colour_fields(tuple(rng.integers(0, 256, size=3)))))
        trg = [trg[k] for k in rng.permutation(len(trg))]
        p = pbmap.default_params(odo)
        if trial % 2:
            p.hue_threshold = 0.45
        got = pbmap.register_planes(ref, trg, 0, mode, p)
        want = O.register_planes(ref, trg, 0, mode, params_dict(p))
        assert got["status"] == want["status"] and got["match"] == want["match"], trial
        if want["status"] == 0:
            rot_err, tr_err = synth.pose_error(got["pose"], want["pose"])
            assert rot_err < 2e-6 and tr_err < 5e-6
        p.use_color = 0
        fewer += len(got["match"]) < len(pbmap.register_planes(ref, trg, 0, mode, p)["match"])
    assert fewer >= 3          # the colour test really removed matches in some scenes


def test_merged_planes_pool_their_colour():
    """rgbd360_merge_planes (Frame360::mergePlanes -> mergePlane2 + calcMainColor on the pooled inliers): the merged record's colour
    is the count-weighted pool of its pieces'."""
    T = synth.make_pose(np.eye(3), np.asarray(synth.CAM_A, float))
    wall = room_planes(T)[2]
    a = dict(wall, count=3000, area=np.float32(4.0), **colour_fields((200, 60, 50)))
    b = dict(wall, count=1000, area=np.float32(2.0), root=7, **colour_fields((180, 80, 50)))
    b["centroid"] = (np.asarray(wall["centroid"]) + 0.1 * np.asarray(wall["ppal_dir"])).astype(np.float32)
    a["color_count"], b["color_count"] = 3000, 1000
    out = pbmap.merge_planes([a, b])
    assert len(out) == 1 and out[0]["color_count"] == 4000
    want = (3000 * a["color_nrgb"].astype(np.float64) + 1000 * b["color_nrgb"].astype(np.float64)) / 4000
    assert np.abs(out[0]["color_nrgb"] - want).max() < 1e-6
    assert abs(out[0]["intensity"] - (3000 * float(a["intensity"]) + 1000 * float(b["intensity"])) / 4000) < 1e-3
    assert abs(out[0]["hist_h"].sum() - 1.0) < 1e-5 and (out[0]["color_dev"] >= 0.004 - 1e-6).all()


def test_group_planes_pools_pieces_of_neighbouring_sensors():
    """Frame360::groupPlanes (Frame360.h:741-833), the step of getPlanes between the sensors' plane lists and mergePlanes: a plane of
    sensor s is pooled into a plane of sensor s - 1 (or one that absorbed a piece of it) when the two hull polygons come within 0.5 m on
    one surface; sensor 7 also sees sensor 0's planes (the ring closes); non-neighbours are left to mergePlanes; the size / curvature
    gates are the source's (area OR curvature for the new piece, area AND curvature for the absorbing plane); nothing is filtered."""
    o, e1, e2 = np.array([-1.0, -1.5, 3.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    sq = lambda x0, y0, s, count, root: _poly_plane([(x0, y0), (x0 + s, y0), (x0 + s, y0 + s), (x0, y0 + s)], o, e1, e2, count, root)
    a = sq(0.0, 0.0, 1.0, 2000, 1)           # sensor 0
    b = sq(1.4, 0.0, 1.0, 2000, 2)           # 0.4 m to the right of a: inside groupPlanes' 0.5 m, outside mergePlanes' 0.3 m
    assert len(pbmap.merge_planes([a, b])) == 2
    empty = []
    out = pbmap.group_planes([[a], [b]] + [empty] * 6)
    assert len(out) == 1 and out[0]["count"] == 4000 and abs(out[0]["area"] - 2.4) < 1e-3          # the hull of both squares
    # not neighbours (sensors 0 and 2): both stay
    assert len(pbmap.group_planes([[a], empty, [b]] + [empty] * 5)) == 2
    # a chain: the piece of sensor 2 joins what sensor 1's piece was pooled into
    c = sq(2.8, 0.0, 1.0, 2000, 3)
    out = pbmap.group_planes([[a], [b], [c]] + [empty] * 5)
    assert len(out) == 1 and out[0]["count"] == 6000
    # the ring closes: sensor 7 against sensor 0 ...
    assert len(pbmap.group_planes([[a]] + [empty] * 6 + [[b]])) == 1
    # ... but sensor 6 does not
    assert len(pbmap.group_planes([[a]] + [empty] * 5 + [[b], empty])) == 2
    # 0.6 m apart: too far even for groupPlanes; another surface (0.2 m behind): the offset along the normal forbids it
    assert len(pbmap.group_planes([[a], [sq(1.6, 0.0, 1.0, 2000, 2)]] + [empty] * 6)) == 2
    behind = _poly_plane([(1.4, 0), (2.4, 0), (2.4, 1), (1.4, 1)], o + np.array([0, 0, 0.2]), e1, e2, 2000, 2)
    assert len(pbmap.group_planes([[a], [behind]] + [empty] * 6)) == 2
    # gates: an absorbing plane under 0.5 m2 never absorbs; a small new piece still joins a large plane when it is flat (the source's OR)
    small_abs = sq(0.0, 0.0, 0.6, 700, 1)    # 0.36 m2
    assert len(pbmap.group_planes([[small_abs], [sq(0.9, 0.0, 1.0, 2000, 2)]] + [empty] * 6)) == 2
    small_new = sq(1.3, 0.0, 0.6, 700, 2)
    assert len(pbmap.group_planes([[a], [small_new]] + [empty] * 6)) == 1
    curved = dict(small_new)
    curved["curvature"] = np.float32(0.01)   # neither large nor flat: appended
    assert len(pbmap.group_planes([[a], [curved]] + [empty] * 6)) == 2
    # nothing is dropped: a sliver stays in the list (mergePlanes / the subgraph selection filter later)
    sliver = _poly_plane([(5, 5), (5.05, 5), (5.05, 7), (5, 7)], o, e1, e2, 100, 9)
    assert len(pbmap.group_planes([[a, sliver]] + [empty] * 7)) == 2
    # getPlanes = groupPlanes then mergePlanes: the frame's list
    frame = pbmap.merge_planes(pbmap.group_planes([[a], [b], [c]] + [empty] * 5))
    assert len(frame) == 1 and frame[0]["count"] == 6000


def test_pool_sensor_planes_is_the_tail_of_getPlanesSensor():
    """rgbd360_pool_sensor_planes = Frame360.h:1034-1068: a region under min_area_plane (0.12 m2) or over max_elongation_plane (6) never
    reaches local_planes_ (so groupPlanes' `area > 0.5 || curvature < max` gate, :764, never pools such a fragment into a neighbour's wall);
    flat regions of one surface -- normals within 0.99, centres within 5 cm along the normal, outlines within 0.2 m -- are pooled into
    the first kept plane in input order (isSamePlane(0.99, 0.05, 0.2) + mergePlane2); a parallel patch 8 cm behind, a co-planar patch
    0.5 m away and a curved region stay on their own."""
    e1, e2, org = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0.0, 0, 2.5])
    sq = lambda x0, y0, w, h: [(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)]
    wall_a = _poly_plane(sq(-1.0, -0.5, 1.0, 1.0), org, e1, e2, 2000, 10)
    wall_b = _poly_plane(sq(0.1, -0.5, 1.0, 1.0), org, e1, e2, 2000, 20)                  # same surface, outlines 0.1 m apart
    behind = _poly_plane(sq(0.1, -0.5, 1.0, 1.0), org + np.array([0, 0, 0.08]), e1, e2, 2000, 30)      # parallel, 8 cm behind
    away = _poly_plane(sq(1.6, -0.5, 1.0, 1.0), org, e1, e2, 2000, 40)                    # co-planar, 0.5 m beyond wall_b
    tiny = _poly_plane(sq(-1.0, 0.6, 0.3, 0.3), org, e1, e2, 90, 50)                      # 0.09 m2
    strip = _poly_plane(sq(-1.0, -0.9, 2.0, 0.2), org, e1, e2, 400, 60)                   # elongation 10
    curved = dict(_poly_plane(sq(-1.0, -0.5, 1.0, 1.0), org, e1, e2, 2000, 70), curvature=np.float32(0.01))
    got = pbmap.pool_sensor_planes([wall_a, tiny, wall_b, strip, behind, away, curved])
    assert [p["root"] for p in got] == [10, 30, 40, 70]
    m = got[0]
    assert m["count"] == 4000 and abs(m["area"] - 2.1) < 0.01                              # the hull of both squares
    assert float(m["normal"] @ wall_a["normal"]) > 1 - 1e-6 and abs(m["d"] - wall_a["d"]) < 1e-5
    assert all(p["area"] >= 0.12 and p["elongation"] <= 6.0 for p in got)
    # order matters as in the source: the incoming region is pooled into the FIRST kept plane that accepts it
    c = _poly_plane(sq(1.15, -0.5, 0.4, 1.0), org, e1, e2, 800, 80)                       # bridges wall_b and `away` (0.05 m / 0.05 m)
    got = pbmap.pool_sensor_planes([wall_b, away, c])
    assert [p["root"] for p in got] == [20, 40] and got[0]["count"] == 2800
    got = pbmap.pool_sensor_planes([away, wall_b, c])
    assert [p["root"] for p in got] == [40, 20] and got[0]["count"] == 2800
    # records without a polygon: the moment rectangle stands in for the outline
    bare = [_poly_plane(sq(-1.0, -0.5, 1.0, 1.0), org, e1, e2, 2000, 10, with_hull=False), _poly_plane(sq(0.1, -0.5, 1.0, 1.0), org, e1, e2, 2000, 20, with_hull=False)]
    assert len(pbmap.pool_sensor_planes(bare)) == 1
    assert pbmap.pool_sensor_planes([]) == []


def test_pool_sensor_planes_random_sets_match_the_numpy_restatement():
    """rgbd360_pool_sensor_planes against oracle/pbmap_ref.py's independent restatement (records without polygons: the moment rectangle is
    the outline) on random sets of wall pieces in random order: which regions survive, which are pooled into which, the pooled fit."""
    rng = np.random.default_rng(18)
    pooled_somewhere = dropped_somewhere = 0
    for trial in range(30):
        planes = []
        for w in range(int(rng.integers(1, 4))):
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            pdv = np.cross(nrm, rng.normal(size=3))
            pdv /= np.linalg.norm(pdv)
            dist = rng.uniform(1.5, 4.0)
            pieces = int(rng.integers(1, 5))
            gap = rng.choice([0.05, 0.15, 0.4])                                 # between neighbouring pieces' outlines: under / over the 0.2 m proximity
            for k in range(pieces):
                half = rng.uniform(0.25, 0.6)
                centre = -nrm * dist + pdv * k * (2 * 0.6 + gap) + nrm * rng.choice([0.0, 0.0, 0.08]) + rng.normal(size=3) * 0.002
                planes.append(_rect_plane(centre, nrm + rng.normal(size=3) * 0.004, pdv, 0.6, half if rng.random() < 0.8 else 0.05,
                                          int(rng.integers(200, 3000)), len(planes)))
                if rng.random() < 0.15:
                    planes[-1]["curvature"] = np.float32(0.01)
        planes = [planes[i] for i in rng.permutation(len(planes))]
        got, want = pbmap.pool_sensor_planes(planes), O.pool_sensor_planes(planes)
        assert [(p["count"], p["root"]) for p in got] == [(p["count"], p["root"]) for p in want], (trial, [(p["count"], p["root"]) for p in got], [(p["count"], p["root"]) for p in want])
        for a, b in zip(got, want):
            assert np.allclose(a["centroid"], b["centroid"], atol=2e-5) and abs(a["area"] - b["area"]) <= 2e-4 * max(b["area"], 1.0)
        pooled_somewhere += int(sum(p["count"] for p in got) == sum(p["count"] for p in planes if p["area"] >= 0.12 and p["elongation"] <= 6.0) and len(got) < sum(1 for p in planes if p["area"] >= 0.12 and p["elongation"] <= 6.0))
        dropped_somewhere += int(any(p["area"] < 0.12 or p["elongation"] > 6.0 for p in planes))
    assert pooled_somewhere >= 5 and dropped_somewhere >= 5, (pooled_somewhere, dropped_somewhere)


def test_merge_planes_has_no_containment_test():
    """Frame360::mergePlanes / groupPlanes test proximity vertex against vertex and edge against edge only (Frame360.h:680-711, 788-815): a
    panel inside a wall's hull, parallel to it and within normal_offset, but farther than `proximity` from the wall's OUTLINE, stays a
    plane of its own (round 5 merged it through a point-in-polygon test the reference does not have)."""
    e1, e2, org = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0.0, 0, 2.5])
    sq = lambda x0, y0, w, h: [(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)]
    wall = _poly_plane(sq(-2.0, -1.5, 4.0, 3.0), org, e1, e2, 8000, 10)
    panel = _poly_plane(sq(-0.4, -0.4, 0.8, 0.8), org + np.array([0, 0, -0.04]), e1, e2, 1500, 20)    # 4 cm in front, 1.1 m from every edge
    assert len(pbmap.merge_planes([wall, panel])) == 2
    assert len(pbmap.group_planes([[wall], [panel]])) == 2
    near = _poly_plane(sq(1.75, -0.4, 0.8, 0.8), org + np.array([0, 0, -0.04]), e1, e2, 1500, 30)      # the same panel at the wall's edge
    assert len(pbmap.merge_planes([wall, near])) == 1
