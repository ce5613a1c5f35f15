"""Writes tests/golden/sample_pair.npz: the DATA of the one real RGB-D pair the reference ships, BASELINE.json configs[0].

Build container only.  Reads /root/reference/samples/sphere_images_{1,10}.bin (Frame360::serialize's Boost binary archive,
Frame360.h:231-266 + cvmat_serialization.h:23-55) and Calibration/Extrinsics/Rt_0[1-8].txt (Calib360.h:122-131) IN PLACE and stores
what they hold as arrays:

  rgb_1, rgb_10      [8, 240, 320, 3] uint8    the eight sensors' colour images of frames 1 and 10
  depth_1, depth_10  [8, 240, 320]    uint16   their depth images, millimetres (0 = no measurement)
  Rt                 [8, 4, 4]        float64  sensor -> rig transforms as the text files give them

Inputs only: sensor images and calibration numbers, no source text.  The expected outputs (CRCs of the stitched panoramas, the
oracle's poses and iteration counts, the plane chain's verdict) are tests/golden/config1_samples.json, written by
tests/tools/config1_samples.py --write from the same files.  -m gpu tests (tests/test_samples_gpu.py) put this pair through the
HIP path on the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
REF = "/root/reference"


def main():
    import config1_samples as c1          # the test-side parser of the archive layout
    out = {}
    for idx in (1, 10):
        frames = c1.load_frame(os.path.join(REF, "samples", "sphere_images_%d.bin" % idx))
        out["rgb_%d" % idx] = np.stack([f[0] for f in frames])
        out["depth_%d" % idx] = np.stack([f[1] for f in frames])
    out["Rt"] = np.stack([np.loadtxt(os.path.join(REF, "Calibration", "Extrinsics", "Rt_0%d.txt" % (s + 1))) for s in range(8)])
    path = os.path.join(HERE, "sample_pair.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
