"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else): numpy restatement of RegisterRGBD360::RegisterPbMap
(reference include/RegisterRGBD360.h:110-338) -- subgraph selection, plane matching, pose of the matched planes.

The matcher (mrpt::pbmap::SubgraphMatcher::compareSubgraphs) and the pose fit (mrpt::pbmap::ConsistencyTest::
estimatePoseWithCovariance) are third-party MRPT code that is NOT in the reference tree and not pinned to a version
(top CMakeLists.txt:11-20: bare FIND_PACKAGE(MRPT ... pbmap)); the reference has no test or recorded result for them.
PARITY UNPINNED: this file restates the published method (Fernandez-Moral et al., "Fast place recognition with plane-based
maps", ICRA 2013) with the thresholds of the reference's config_files/configLocaliser_spherical*.ini, written
independently of rgbd360_amd/csrc/pbmap_register.h: the best interpretation is found by plain enumeration of every
assignment (no branch and bound), the rotation by numpy's SVD with the determinant fix (no Jacobi / cross products), the
translation by numpy's least squares.
"""
from __future__ import annotations

import math

import numpy as np

DEFAULT_6DoF, PLANAR_3DoF, ODOMETRY_6DoF, PLANAR_ODOMETRY_3DoF = 0, 1, 2, 3      # RegisterRGBD360.h:258-264


def default_params(odometry: bool = False) -> dict:
    """config_files/configLocaliser_spherical.ini / configLocaliser_sphericalOdometry.ini + Miscellaneous.h:54-60."""
    return dict(dist_d=0.5 if odometry else 0.4, angle_deg=50.0 if odometry else 40.0,
                elongation_threshold=2.5 if odometry else 3.8, area_threshold=3.0 if odometry else 4.0,
                dist_threshold=3.0 if odometry else 4.0, angle_threshold_deg=10.0 if odometry else 9.0,
                height_threshold=0.33, cos_normal_threshold=0.985 if odometry else 0.99, min_planes_recognition=3,
                max_curvature_plane=0.0013, min_area_plane=0.12, max_elongation_plane=6.0, up_axis=0, planar_normal_tol=0.08,
                max_conditioning=100.0, sigma_dist=0.02, sigma_normal=0.0398,
                # [unary] colour (configLocaliser_spherical*.ini:19-21); hue_threshold 0 = not tested (Frame360.h:673 has it commented out)
                use_color=1, color_threshold=0.07, intensity_threshold=100.0 if odometry else 150.0, hue_threshold=0.0)


def _f32(x):
    return float(np.float32(x))


def _ratio(a, b):
    lo, hi = min(a, b), max(a, b)
    if lo > 0:
        return hi / lo
    return math.inf if hi > 0 else 1.0


def select_subgraph(planes, max_match_planes, P):
    """setReference / setTarget (RegisterRGBD360.h:110-195) after the Frame360.h:1034,1041 size filters."""
    def well_formed(p):
        v = [*np.asarray(p["normal"], np.float32), *np.asarray(p["centroid"], np.float32), _f32(p["d"]), _f32(p["curvature"]),
             _f32(p["area"]), _f32(p["elongation"])]
        n = np.asarray(p["normal"], np.float32).astype(np.float64)
        return bool(np.all(np.isfinite(v))) and _f32(p["area"]) >= 0 and abs(float(n @ n) - 1.0) < 1e-3
    kept = [i for i, p in enumerate(planes) if well_formed(p)
            and not (_f32(p["area"]) < _f32(P["min_area_plane"])) and not (_f32(p["elongation"]) > _f32(P["max_elongation_plane"]))]
    flat = lambda i: _f32(planes[i]["curvature"]) < _f32(P["max_curvature_plane"])
    if max_match_planes > 0 and len(kept) > max_match_planes:
        areas = [_f32(planes[i]["area"]) if flat(i) else 0.0 for i in kept]
        thr = sorted(areas)[len(kept) - max_match_planes - 1]
        return [i for i, a in zip(kept, areas) if a > thr]
    return [i for i in kept if flat(i)]


def color_ok(a, b, P):
    """Radiometric unary constraint (mrpt::pbmap::SubgraphMatcher::evalUnaryConstraints, third-party): planes that both carry
    colour agree in every channel of the normalised colour, in mean intensity, and -- when hue_threshold > 0 -- in the
    Bhattacharyya distance of their hue histograms."""
    if not P.get("use_color", 1) or int(a.get("color_count", 0)) <= 0 or int(b.get("color_count", 0)) <= 0:
        return True
    ca, cb = np.asarray(a["color_nrgb"], np.float32).astype(np.float64), np.asarray(b["color_nrgb"], np.float32).astype(np.float64)
    if not np.all(np.abs(ca - cb) < _f32(P["color_threshold"])):
        return False
    if _f32(P.get("intensity_threshold", 0)) > 0 and not abs(_f32(a["intensity"]) - _f32(b["intensity"])) < _f32(P["intensity_threshold"]):
        return False
    if _f32(P.get("hue_threshold", 0)) > 0:
        ha, hb = np.asarray(a["hist_h"], np.float32).astype(np.float64), np.asarray(b["hist_h"], np.float32).astype(np.float64)
        if not math.sqrt(max(0.0, 1.0 - float(np.sqrt(ha * hb).sum()))) < _f32(P["hue_threshold"]):
            return False
    return True


def unary_ok(a, b, mode, P):
    if not _ratio(_f32(a["area"]), _f32(b["area"])) < _f32(P["area_threshold"]):
        return False
    if not _ratio(_f32(a["elongation"]), _f32(b["elongation"])) < _f32(P["elongation_threshold"]):
        return False
    if not color_ok(a, b, P):
        return False
    na, nb = np.asarray(a["normal"], np.float32).astype(np.float64), np.asarray(b["normal"], np.float32).astype(np.float64)
    da, db = _f32(a["d"]), _f32(b["d"])
    if mode in (ODOMETRY_6DoF, PLANAR_ODOMETRY_3DoF):
        if not float(na @ nb) > math.cos(_f32(P["angle_deg"]) * math.pi / 180):
            return False
        if not abs(da - db) < _f32(P["dist_d"]):
            return False
    if mode in (PLANAR_3DoF, PLANAR_ODOMETRY_3DoF):
        ua, ub = na[P["up_axis"]], nb[P["up_axis"]]
        if not abs(ua - ub) < _f32(P["planar_normal_tol"]):
            return False
        if abs(ua) > 0.98 and not abs(da - db) < _f32(P["dist_d"]):
            return False
    return True


def binary_ok(a1, a2, b1, b2, P):
    f = lambda v: np.asarray(v, np.float32).astype(np.float64)
    ang = lambda u, v: math.acos(min(1.0, max(-1.0, float(f(u) @ f(v)))))
    if not abs(ang(a1["normal"], a2["normal"]) - ang(b1["normal"], b2["normal"])) < _f32(P["angle_threshold_deg"]) * math.pi / 180:
        return False
    # the point that stands for a plane: the hull polygon's mass centre where the record carries one (what MRPT's
    # computeMassCenterAndArea leaves in v3center, Frame360.h:1028), else the inlier centroid -- as pbm::center_of in the library
    ctr = lambda p: f(p["center_hull"]) if int(p.get("hull_points", 0)) > 0 else f(p["centroid"])
    ca, cb = ctr(a2) - ctr(a1), ctr(b2) - ctr(b1)
    if not _ratio(math.sqrt(float(ca @ ca)), math.sqrt(float(cb @ cb))) < _f32(P["dist_threshold"]):
        return False
    h = _f32(P["height_threshold"])
    if not abs(float(f(a1["normal"]) @ ca) - float(f(b1["normal"]) @ cb)) < h:
        return False
    if not abs(float(f(a2["normal"]) @ ca) - float(f(b2["normal"]) @ cb)) < h:
        return False
    return True


def handedness_ok(ref, trg, pairs, i, j):
    """No mirror images: every triple of matched normals keeps the sign of n1 . (n2 x n3) (near-coplanar triples excepted)."""
    f = lambda v: np.asarray(v, np.float32).astype(np.float64)
    for a in range(len(pairs)):
        for b in range(a + 1, len(pairs)):
            ta = float(f(ref[pairs[a][0]]["normal"]) @ np.cross(f(ref[pairs[b][0]]["normal"]), f(ref[i]["normal"])))
            tb = float(f(trg[pairs[a][1]]["normal"]) @ np.cross(f(trg[pairs[b][1]]["normal"]), f(trg[j]["normal"])))
            if abs(ta) > 0.1 and abs(tb) > 0.1 and (ta > 0) != (tb > 0):
                return False
    return True


def best_interpretation(ref, trg, ri, ti, mode, P):
    """Every consistent assignment is visited; the winner has the most matches, then the largest matched reference area."""
    best = dict(n=-1, area=-1.0, pairs=[])

    def rec(k, pairs, used, area):
        if k == len(ri):
            if len(pairs) > best["n"] or (len(pairs) == best["n"] and area > best["area"]):
                best.update(n=len(pairs), area=area, pairs=list(pairs))
            return
        a = ref[ri[k]]
        for j in ti:
            if j in used or not unary_ok(a, trg[j], mode, P):
                continue
            if all(binary_ok(ref[i0], a, trg[j0], trg[j], P) for i0, j0 in pairs) and handedness_ok(ref, trg, pairs, ri[k], j):
                pairs.append((ri[k], j))
                used.add(j)
                rec(k + 1, pairs, used, area + float(np.float32(a["area"])))
                used.discard(j)
                pairs.pop()
        rec(k + 1, pairs, used, area)

    rec(0, [], set(), 0.0)
    return best["pairs"], max(best["area"], 0.0)


def fit_pose(ref, trg, pairs, P):
    """(status, T 4x4, info 6x6): status 0 ok, 2 not observable / inconsistent."""
    f = lambda v: np.asarray(v, np.float32).astype(np.float64)
    Nr = np.stack([f(ref[i]["normal"]) for i, _ in pairs])
    Nt = np.stack([f(trg[j]["normal"]) for _, j in pairs])
    w = np.array([_f32(trg[j]["area"]) for _, j in pairs])
    M = (Nr * w[:, None]).T @ Nt
    U, S, Vt = np.linalg.svd(M)
    if not (S[0] > 0 and S[1] > 1e-6 * S[0]):
        return 2, np.eye(4), np.zeros((6, 6))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt)) or 1.0])
    R = U @ D @ Vt
    e = np.array([_f32(trg[j]["d"]) - _f32(ref[i]["d"]) for i, j in pairs])
    H = (Nr * w[:, None]).T @ Nr
    ev = np.linalg.eigvalsh(H)
    if not (ev[0] > 0 and ev[-1] / ev[0] < _f32(P["max_conditioning"])):
        return 2, np.eye(4), np.zeros((6, 6))
    t = np.linalg.solve(H, (Nr * w[:, None]).T @ e)
    n = Nt @ R.T
    if not np.all(np.sum(n * Nr, axis=1) > _f32(P["cos_normal_threshold"])):
        return 2, np.eye(4), np.zeros((6, 6))
    info = np.zeros((6, 6))
    sd, sn = _f32(P["sigma_dist"]), _f32(P["sigma_normal"])
    for k in range(len(pairs)):
        nn = np.outer(n[k], n[k])
        info[:3, :3] += w[k] * nn / (sd * sd)
        info[3:, 3:] += w[k] * (np.eye(3) - nn) / (sn * sn)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return 0, T, info


def register_planes(ref, trg, max_match_planes=0, mode=DEFAULT_6DoF, P=None):
    P = P or default_params(mode in (ODOMETRY_6DoF, PLANAR_ODOMETRY_3DoF))
    ri, ti = select_subgraph(ref, max_match_planes, P), select_subgraph(trg, max_match_planes, P)
    pairs, area = best_interpretation(ref, trg, ri, ti, mode, P)
    out = dict(status=0, pose=np.eye(4), info=np.zeros((6, 6)), match=dict(pairs), area_matched=area)
    if len(pairs) < max(P["min_planes_recognition"], 3):
        out["status"] = 1
        return out
    st, T, info = fit_pose(ref, trg, pairs, P)
    out["status"] = st
    if st == 0:
        out["pose"], out["info"] = T, info
    return out


# ---- Frame360::mergePlanes (Frame360.h:655-733) on moment rectangles: independent restatement (numpy eigh, no Jacobi) ------------
def _axes(p):
    f = lambda v: np.asarray(v, np.float32).astype(np.float64)
    n, pp = f(p["normal"]), f(p.get("ppal_dir", (0, 0, 0)))
    if not np.linalg.norm(pp) > 0.5:
        pp = np.cross(n, np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0]))
    pp = pp - (pp @ n) * n
    pp /= np.linalg.norm(pp)
    return n, pp, np.cross(n, pp)


def _moments(p):
    n, pp, qq = _axes(p)
    el = _f32(p["elongation"]) if np.isfinite(_f32(p["elongation"])) and _f32(p["elongation"]) > 0 else 1.0
    area = _f32(p["area_moment"]) if _f32(p.get("area_moment", 0.0)) > 0 else _f32(p["area"])      # pbm::moment_area
    l1, l2 = area / (12.0 * el), area * el / 12.0
    cv = min(_f32(p["curvature"]), 0.5)
    l0 = cv * (l1 + l2) / (1.0 - cv)
    C = l0 * np.outer(n, n) + l1 * np.outer(qq, qq) + l2 * np.outer(pp, pp)
    return float(p["count"] if p["count"] > 0 else 1), np.asarray(p["centroid"], np.float32).astype(np.float64), C


def _outline(p):
    n, pp, qq = _axes(p)
    el = _f32(p["elongation"]) if np.isfinite(_f32(p["elongation"])) and _f32(p["elongation"]) > 0 else 1.0
    a, b = math.sqrt(3.0 * _f32(p["area"]) * el / 12.0), math.sqrt(3.0 * _f32(p["area"]) / (12.0 * el))
    c = np.asarray(p["centroid"], np.float32).astype(np.float64)
    return [c + su * a * pp + sv * b * qq for su in (-1, 0, 1) for sv in (-1, 0, 1)], (n, pp, qq, a, b, c)


def _same_surface(pj, pk, cos_normal, dist_d, proximity, normal_offset):
    nj = np.asarray(pj["normal"], np.float32).astype(np.float64)
    nk = np.asarray(pk["normal"], np.float32).astype(np.float64)
    if not nj @ nk > _f32(cos_normal) or not abs(_f32(pj["d"]) - _f32(pk["d"])) < _f32(dist_d):
        return False
    Pj, fj = _outline(pj)
    Pk, fk = _outline(pk)
    for a in Pj:
        for b in Pk:
            if np.linalg.norm(a - b) < _f32(proximity) and abs(nj @ (a - b)) < _f32(normal_offset):
                return True

    def inside(q, frame):
        n, pp, qq, a, b, c = frame
        d = q - c
        return abs(d @ pp) <= a and abs(d @ qq) <= b and abs(n @ d) < _f32(normal_offset)
    return any(inside(q, fj) for q in Pk) or any(inside(q, fk) for q in Pj)


def merge_planes(planes, max_curvature=0.0013, cos_normal=0.99, dist_d=0.45, proximity=0.3, normal_offset=0.06, min_area=0.12,
                 max_elongation=6.0):
    v = [dict(p) for p in planes if not _f32(p["area"]) < _f32(min_area) and not _f32(p["elongation"]) > _f32(max_elongation)]
    j = 0
    while j < len(v):
        if True:
            merged = True
            while merged and _f32(v[j]["curvature"]) < _f32(max_curvature):      # Frame360.h:663 re-tested after every merge (`j--`, :727-731)
                merged = False
                for k in range(j + 1, len(v)):
                    if not _f32(v[k]["curvature"]) < _f32(max_curvature) or not _same_surface(v[j], v[k], cos_normal, dist_d, proximity, normal_offset):
                        continue
                    (na, ca, Ca), (nb, cb, Cb) = _moments(v[j]), _moments(v[k])
                    n = na + nb
                    c = (na * ca + nb * cb) / n
                    Cm = (na * (Ca + np.outer(ca - c, ca - c)) + nb * (Cb + np.outer(cb - c, cb - c))) / n
                    w, V = np.linalg.eigh(Cm)
                    nn = V[:, 0]
                    d = -float(nn @ c)
                    if d < 0:
                        nn, d = -nn, -d
                    l0, l1, l2 = (max(float(x), 0.0) for x in w)
                    v[j] = dict(centroid=c.astype(np.float32), normal=nn.astype(np.float32), d=np.float32(d),
                                curvature=np.float32(l0 / (l0 + l1 + l2) if l0 + l1 + l2 > 0 else 0.0), count=int(n),
                                root=min(v[j]["root"], v[k]["root"]), area=np.float32(12.0 * math.sqrt(l1 * l2)),
                                elongation=np.float32(math.sqrt(l2 / l1) if l1 > 0 else math.inf), ppal_dir=V[:, 2].astype(np.float32))
                    del v[k]
                    merged = True
                    break
        j += 1
    return v


# ---- the tail of Frame360::getPlanesSensor (Frame360.h:1034-1068) on records without polygons: independent restatement of
#      rgbd360_pool_sensor_planes (mrpt's isSamePlane / isPlaneNearby on the moment rectangle's corners; the pooled fit as in merge_planes) ----
def _seg_seg_dist2(p0, p1, q0, q1):
    """Squared distance between two 3-D segments by dense sampling of the closed-form minimiser's candidates: the unconstrained optimum
    clamped, and the four endpoint-to-segment distances (exact for segments: the minimum is at one of them)."""
    def pt_seg(x, a, b):
        ab = b - a
        den = float(ab @ ab)
        t = 0.0 if den <= 0 else min(1.0, max(0.0, float((x - a) @ ab) / den))
        d = x - (a + t * ab)
        return float(d @ d)
    u, v, w = p1 - p0, q1 - q0, p0 - q0
    a, b, c, d, e = float(u @ u), float(u @ v), float(v @ v), float(u @ w), float(v @ w)
    D = a * c - b * b
    best = min(pt_seg(p0, q0, q1), pt_seg(p1, q0, q1), pt_seg(q0, p0, p1), pt_seg(q1, p0, p1))
    if D > 1e-12 * max(a * c, 1e-300):
        sc, tc = (b * e - c * d) / D, (a * e - b * d) / D
        if 0.0 <= sc <= 1.0 and 0.0 <= tc <= 1.0:
            dP = w + sc * u - tc * v
            best = min(best, float(dP @ dP))
    return best


def _corners(p):
    pts, _ = _outline(p)
    return [pts[0], pts[2], pts[8], pts[6]]          # (-,-) (-,+) (+,+) (+,-): the rectangle as a closed outline


def _is_same_plane(a, b, cos_normal, dist_normal, proximity):
    f = lambda v: np.asarray(v, np.float32).astype(np.float64)
    na = f(a["normal"])
    if na @ f(b["normal"]) < _f32(cos_normal):
        return False
    ca, cb = f(a["centroid"]), f(b["centroid"])
    if abs(na @ (cb - ca)) > _f32(dist_normal):
        return False
    p2 = float(_f32(proximity)) ** 2
    if (ca - cb) @ (ca - cb) < p2:
        return True
    A, B = _corners(a), _corners(b)
    if any((x - cb) @ (x - cb) < p2 for x in A) or any((ca - y) @ (ca - y) < p2 for y in B):
        return True
    if any((x - y) @ (x - y) < p2 for x in A for y in B):
        return True
    return any(_seg_seg_dist2(A[i], A[(i + 1) % 4], B[j], B[(j + 1) % 4]) < p2 for i in range(4) for j in range(4))


def pool_sensor_planes(planes, max_curvature=0.0013, min_area=0.12, max_elongation=6.0, cos_normal=0.99, dist_normal=0.05, proximity=0.2):
    v = []
    for p in planes:
        if _f32(p["area"]) < _f32(min_area) or _f32(p["elongation"]) > _f32(max_elongation):
            continue
        hit = None
        if _f32(p["curvature"]) < _f32(max_curvature):
            for j, q in enumerate(v):
                if _f32(q["curvature"]) < _f32(max_curvature) and _is_same_plane(q, p, cos_normal, dist_normal, proximity):
                    hit = j
                    break
        if hit is None:
            v.append(dict(p))
        else:
            v[hit] = merge_planes([v[hit], dict(p)], max_curvature=1e9, cos_normal=-2.0, dist_d=1e9, proximity=1e9, normal_offset=1e9, min_area=-1.0,
                                  max_elongation=1e9)[0]          # (the pooled fit of exactly these two: every test of merge_planes forced to pass)
    return v
