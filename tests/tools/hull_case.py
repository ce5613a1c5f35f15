"""One draw of tests/tools/hull_soak.py in detail: per checked plane the device hull's area against the exact hull's (Qhull, the CPU
checker), and what the deficit is made of.  python tests/tools/hull_case.py W seed which trans rot_deg angular min_inliers refine"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W, seed, which = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
trans, rot, ang, min_inl, refine = float(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), bool(int(sys.argv[8]))
H = W // 2
depth = synth.make_pair(W, H, seed=seed, trans=trans, rot_deg=rot)[which][1]
st = Frame360Stages(RegisterPhotoICP())
st.set_refinement(refine)
out = st.frame_planes(depth, convention=2, angular_threshold=ang, min_inliers=min_inl)
xyz = O.sphere_cloud(depth, 2)
lab = np.asarray(out["labels"]).reshape(-1)
for p in out["planes"]:
    if p["area"] <= 0.12 or p["elongation"] > 6.0 or p["hull_points"] < 3:
        continue
    exact, center, nv = O.f360_hull_stats(xyz, out["labels"], p)
    m = lab == int(p["root"])
    rr, cc = np.nonzero(m.reshape(H, W))
    dist = float(np.linalg.norm(p["centroid"]))
    px_m = dist * 2 * np.pi / W                         # width of a pixel at the plane's range
    print("plane root %8d: %7d inliers, rows %d-%d cols %d-%d, range %.2f m (pixel %.1f mm), area %.5f / exact %.5f = %.5f (deficit %.1f cm2 = %.2f px2), "
          "elongation %.2f, %d hull points / %d exact vertices, centre off %.1f mm" % (
              p["root"], p["count"], rr.min(), rr.max(), cc.min(), cc.max(), dist, px_m * 1e3, p["area"], exact, p["area"] / exact,
              (exact - p["area"]) * 1e4, (exact - p["area"]) / px_m ** 2, p["elongation"], p["hull_points"], nv,
              float(np.abs(p["center_hull"] - center).max()) * 1e3))
    if p["area"] / exact < 0.9985:
        # which exact hull vertices does the device polygon lack?
        from scipy.spatial import ConvexHull
        pts = np.asarray(xyz, np.float64).reshape(-1, 3)
        idx = np.nonzero(m)[0]
        n = np.asarray(p["normal"], np.float64); n /= np.linalg.norm(n)
        e1 = np.cross(n, [1.0, 0.0, 0.0] if abs(n[0]) < 0.9 else [0.0, 1.0, 0.0]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(n, e1)
        c = np.asarray(p["centroid"], np.float64)
        uv = np.stack([(pts[idx] - c) @ e1, (pts[idx] - c) @ e2], axis=1)
        hv = ConvexHull(uv).vertices
        dev = np.asarray(p["hull"], np.float64)
        duv = np.stack([(dev - c) @ e1, (dev - c) @ e2], axis=1)
        print("   device polygon (%d vertices, in-plane m):" % len(dev), np.round(duv, 3).tolist())
        for v in hv:
            d = np.sqrt(((duv - uv[v]) ** 2).sum(1)).min()
            i = int(idx[v])
            r, cpx = divmod(i, W)
            nb = [int(lab[j]) if 0 <= j < lab.size else None for j in (i - 1, i + 1, i - W, i + W)]
            print("   exact vertex pixel (%4d, %4d) uv (%.3f, %.3f): nearest device vertex %.1f mm away; neighbour labels l/r/u/d %s (own %d)" % (
                r, cpx, uv[v, 0], uv[v, 1], d * 1e3, nb, int(lab[i])))
