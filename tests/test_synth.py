"""The synthetic scene generator: determinism and geometric consistency of its ground truth."""
import math

import numpy as np

from rgbd360_amd import synth


def test_strip_size_does_not_change_the_image():
    T = synth.make_pose(synth.rodrigues([1, 2, 3], 0.1), synth.CAM_A)
    a = synth.render(T, 128, 64, seed=5, strip=32)
    b = synth.render(T, 128, 64, seed=5, strip=7)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_ground_truth_pose_is_consistent_with_the_rendered_depth():
    """Back-project source pixels, move them with T_gt, re-project into the target: the target range there must equal
    the moved point's range (planar walls, up to the nearest-pixel sampling)."""
    W, H = 512, 256
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(W, H, seed=3)
    res = 2 * math.pi / W
    rng = np.random.default_rng(0)
    r = rng.integers(20, H - 20, 400)
    c = rng.integers(0, W, 400)
    d = dB[r, c].astype(np.float64) * 1e-3
    phi = (H / 2 - 0.5 - r) * res
    theta = c * res
    p = np.stack([d * np.sin(phi), -d * np.cos(phi) * np.sin(theta), -d * np.cos(phi) * np.cos(theta)], 1)
    q = p @ T[:3, :3].T + T[:3, 3]
    dist = np.linalg.norm(q, axis=1)
    phi2 = np.arcsin(q[:, 0] / dist)
    th2 = np.arctan2(q[:, 1], q[:, 2]) + math.pi
    r2 = np.rint(H / 2 - 0.5 - phi2 / res).astype(int)
    c2 = np.rint(th2 / res).astype(int) % W
    ok = (r2 >= 0) & (r2 < H)
    dt = dA[r2[ok], c2[ok]].astype(np.float64) * 1e-3
    err = np.abs(dt - dist[ok])
    assert np.median(err) < 0.01 and np.mean(err < 0.05) > 0.9      # room corners / grazing walls are the outliers


def test_trajectory_steps_are_odometry_sized():
    for i in range(0, 70, 7):
        Ta, Tb = synth.trajectory_pose(i), synth.trajectory_pose(i + 1)
        rot, trans = synth.pose_error(np.linalg.inv(Ta) @ Tb, np.eye(4))
        assert 0.01 < trans < 0.12 and rot < math.radians(5)
        assert np.all(Ta[:3, 3] > synth.ROOM_LO + 0.3) and np.all(Ta[:3, 3] < synth.ROOM_HI - 0.3)
