"""Wall time of rgbd360_bilateral_filter (host cloud in, filtered cloud out) per sensor-cloud size.  python tools/bilateral_perf.py
(The bit-identity with the CPU oracle is the GPU test test_bilateral_filter_bit_exact; the oracle's own times in
profiles/r01_bilateral_perf.txt were taken with the test-side helper on the same box.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP


def noisy_cloud(rows, cols, seed=1, noise=0.01):
    rng = np.random.default_rng(seed)
    f, cx, cy = 525.0 * cols / 640.0, cols / 2 - 0.5, rows / 2 - 0.5
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    z = 2.5 + 0.4 * (u - cx) / f + rng.normal(size=u.shape) * noise
    z[rng.random(z.shape) < 0.02] = np.nan
    return np.stack([(u - cx) * z / f, (v - cy) * z / f, z], -1).astype(np.float32)


st = Frame360Stages(RegisterPhotoICP())
for rows, cols in ((120, 160), (240, 320), (480, 640), (1024, 2048)):
    xyz = noisy_cloud(rows, cols)
    got = st.bilateral_filter(xyz, rows, cols)
    t0 = time.perf_counter()
    for _ in range(20):
        got = st.bilateral_filter(xyz, rows, cols)
    t_dev = (time.perf_counter() - t0) / 20
    changed = np.nanmax(np.abs(got[:, 2] - xyz[..., 2].reshape(-1)))
    print("%4d x %4d: device call %.3f ms (host cloud in / out); largest change of a depth %.3f m" % (rows, cols, t_dev * 1e3, changed))
