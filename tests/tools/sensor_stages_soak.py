"""Soak check of the per-sensor stages in front of the path against the CPU oracle, all bit-exact: pcl::FastBilateralFilter on a sensor
cloud (rgbd360_bilateral_filter: integer cell sums, the grid's float operations), CloudRGBD::getPointCloud + the median down-sampling
(rgbd360_sensor_cloud: random sizes, steps, depth ranges, strided images, blocks without depth), Frame360::stitchSphericalImage
(rgbd360_stitch_sphere: random sensor sizes and rig extrinsics).  python tests/tools/sensor_stages_soak.py [n_trials [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages, stitch_sphere
from oracle import oracle as O
O.set_num_threads(min(16, os.cpu_count() or 1))
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)      # [seed]: another draw of cases
reg = RegisterPhotoICP()
st = Frame360Stages(reg)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


bad = 0
for t in range(n_trials):
    # ---- bilateral filter: a slanted wall + boxes + holes + noise seen by a pinhole sensor
    rows, cols = int(rng.integers(40, 260)), int(rng.integers(50, 340))
    f, cx, cy = 131.25 * cols / 160, cols / 2 - 0.5, rows / 2 - 0.5
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    z = rng.uniform(1.0, 5.0) + rng.uniform(-0.6, 0.6) * (u - cx) / f + rng.uniform(-0.3, 0.3) * (v - cy) / f
    for _ in range(int(rng.integers(0, 4))):
        r0, c0 = int(rng.integers(0, rows - 8)), int(rng.integers(0, cols - 8))
        z[r0:r0 + int(rng.integers(4, rows // 2)), c0:c0 + int(rng.integers(4, cols // 2))] *= float(rng.choice([0.5, 0.8, 1.3]))
    z = z + rng.normal(size=z.shape) * float(rng.choice([0.0, 0.003, 0.01, 0.03]))
    z[rng.random(z.shape) < float(rng.choice([0.0, 0.02, 0.2]))] = np.nan
    xyz = np.stack([(u - cx) * z / f, (v - cy) * z / f, z], -1).astype(np.float32)
    xyz[~np.isfinite(xyz[..., 2])] = np.nan
    sigma_s, sigma_r = float(rng.choice([5.0, 7.0, 10.0, 15.0])), float(rng.choice([0.02, 0.05, 0.1]))
    ok_b = same(st.bilateral_filter(xyz, rows, cols, sigma_s, sigma_r), O.fast_bilateral(xyz, rows, cols, sigma_s, sigma_r))
    # ---- sensor cloud
    srows, scols, step = int(rng.integers(30, 250)), int(rng.integers(40, 330)), int(rng.choice([1, 2, 3, 4]))
    big = rng.uniform(100, 12000, (srows, scols + 7)).astype(np.uint16)
    big[rng.random(big.shape) < float(rng.choice([0.0, 0.25, 0.7]))] = 0
    off = int(rng.integers(0, 7))
    d = big[:, off:off + scols]
    lo, hi = float(rng.choice([0.3, 0.5, 1.0])), float(rng.choice([4.0, 10.0]))
    ok_c = same(st.sensor_cloud(d, step, lo, hi), O.sensor_cloud(np.ascontiguousarray(d), step, lo, hi))
    # ---- stitching
    r8, c8 = int(rng.choice([30, 60, 120])), int(rng.choice([40, 80, 160]))
    rgb8 = rng.integers(0, 256, size=(8, r8, c8, 3), dtype=np.uint8)
    d8 = rng.integers(300, 7000, size=(8, r8, c8)).astype(np.uint16)
    d8[rng.random(d8.shape) < float(rng.choice([0.0, 0.1, 0.5]))] = 0
    Rt = []
    for s in range(8):
        R = synth.rodrigues([1.0, 0.0, 0.0], np.radians(45.0 * s + rng.uniform(-2, 2)))
        Rt.append(np.linalg.inv(synth.make_pose(R, rng.normal(size=3) * 0.04)).astype(np.float32))
    K = (r8 * 262.5 / 240, r8 * 262.5 / 240, c8 / 2 - 0.5, r8 / 2 - 0.5)
    a, b = stitch_sphere(reg, rgb8, d8, np.stack(Rt), K)
    a_ref, b_ref = O.stitch_sphere(rgb8, d8, np.stack(Rt), K)
    ok_s = np.array_equal(a, a_ref) and np.array_equal(b, b_ref)
    good = ok_b and ok_c and ok_s
    bad += 0 if good else 1
    print("trial %2d: bilateral %3dx%-3d sigma %.0f / %.2f %s; sensor cloud %3dx%-3d step %d (%.1f, %.0f m) %s; stitch 8 x %3dx%-3d -> %s %s -> %s" % (
        t, cols, rows, sigma_s, sigma_r, "identical" if ok_b else "DIFFERENT", scols, srows, step, lo, hi, "identical" if ok_c else "DIFFERENT",
        c8, r8, "x".join(str(x) for x in np.asarray(b).shape), "identical" if ok_s else "DIFFERENT", "ok" if good else "FAIL"), flush=True)
print("sensor stages soak: %d / %d trials ok" % (n_trials - bad, n_trials))
sys.exit(1 if bad else 0)
