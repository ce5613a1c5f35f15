"""Writes tests/golden/depth_model.json: probes of a seeded depth image corrected by each of the reference's eight intrinsic depth models
(Calibration/Intrinsics/distortion_model1..8, read in place; the files themselves are not copied), computed by the numpy restatement
oracle.depth_model_undistort.  python tests/golden/make_golden_depth_model.py   (build container only)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_depth_model import REF_MODELS, _probe_image
depth = _probe_image()
rng = np.random.default_rng(5)
pix = [[int(rng.integers(0, 240)), int(rng.integers(0, 320))] for _ in range(16)]
out = {"probe_pixels": pix, "how": "oracle.depth_model_undistort(oracle.depth_model_read(model, 2), seeded 320x240 image)"}
for k in range(1, 9):
    got = O.depth_model_undistort(O.depth_model_read(os.path.join(REF_MODELS, "distortion_model%d" % k), 2), depth)
    out["model%d" % k] = {"probes": [float(got[r, c]) for r, c in pix], "sum": float(got.astype(np.float64).sum())}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "depth_model.json"), "w"), indent=1)
print("written")
