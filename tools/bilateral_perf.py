"""Wall time of rgbd360_bilateral_filter (host cloud in, filtered cloud out) next to the CPU oracle, per sensor-cloud size.
python tools/bilateral_perf.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O
from rgbd360_amd.register import Frame360Stages, RegisterPhotoICP
from tests.test_oracle_cpu import _noisy_pinhole_cloud

O.build()
st = Frame360Stages(RegisterPhotoICP())
for rows, cols in ((120, 160), (240, 320), (480, 640), (1024, 2048)):
    xyz, _ = _noisy_pinhole_cloud(rows, cols, seed=1)
    got = st.bilateral_filter(xyz, rows, cols)
    t0 = time.perf_counter()
    for _ in range(20):
        got = st.bilateral_filter(xyz, rows, cols)
    t_dev = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    want = O.fast_bilateral(xyz, rows, cols)
    t_cpu = time.perf_counter() - t0
    same = np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    print("%4d x %4d: device call %.3f ms (host cloud in / out), CPU oracle %.2f ms, bit-identical %s" % (rows, cols, t_dev * 1e3, t_cpu * 1e3, same))
