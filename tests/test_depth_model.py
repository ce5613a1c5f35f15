"""Frame360::undistort's intrinsic depth model (clams::DiscreteDepthDistortionModel; Frame360.h:293-311, 1084-1097, Calib360.h:104-119):
`rgbd360_depth_model_*` (host code of the HIP library: runs without a GPU) against the independent numpy restatement in oracle/oracle.py
on model files written by the test, and -- in the build container, where the reference's own eight model files can be read in place --
against those files, pinned by a small committed fixture (tests/golden/depth_model.json: probes of the corrected image).  The model's
code is third-party (CLAMS, vendored by the reference): parity means "equals the restatement of that source, bit for bit"."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from rgbd360_amd.register import DepthModel, Rgbd360Error

REF_MODELS = "/root/reference/Calibration/Intrinsics"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_model.json")


def write_model(path, width, height, bin_w, bin_h, bin_depth, frustums):
    """frustums: rows of (counts, multipliers) per bin, row-major; the two middle vectors (numerators / denominators) are filler."""
    nby, nbx = len(frustums), len(frustums[0])
    with open(path, "wb") as f:
        f.write(b"DiscreteDepthDistortionModel v01\n")
        f.write(struct.pack("<iiii", width, height, bin_w, bin_h) + struct.pack("<d", bin_depth) + struct.pack("<ii", nbx, nby))
        for row in frustums:
            for counts, mults in row:
                nb = len(mults)
                f.write(struct.pack("<d", nb * bin_depth) + struct.pack("<i", nb) + struct.pack("<d", bin_depth))
                for vec in (counts, np.ones(nb), np.ones(nb), mults):
                    f.write(struct.pack("<iii", 4, nb, 1) + np.asarray(vec, np.float32).tobytes())


def test_depth_model_on_a_written_file(tmp_path):
    rng = np.random.default_rng(4)
    W, H, bw, bh, bd, nb = 64, 48, 8, 6, 2.0, 5
    frustums = [[(rng.choice([10.0, 200.0], size=nb, p=[0.3, 0.7]), 1.0 + 0.05 * rng.normal(size=nb)) for _ in range(W // bw)] for _ in range(H // bh)]
    path = tmp_path / "model"
    write_model(path, W, H, bw, bh, bd, frustums)
    m = DepthModel(path, downsample=1)
    assert (m.width, m.height, m.bin_width, m.bin_height, m.num_bins_x, m.num_bins_y, m.bin_depth) == (W, H, bw, bh, W // bw, H // bh, bd)
    ref = O.depth_model_read(path, 1)
    depth = rng.uniform(0.2, 11.0, size=(H, W)).astype(np.float32)       # beyond the last slice too (its multiplier stands)
    depth[rng.random((H, W)) < 0.1] = 0.0                                  # no measurement: untouched
    got, want = m.undistort(depth), O.depth_model_undistort(ref, depth)
    assert np.array_equal(got, want)
    assert np.array_equal(got == 0, depth == 0) and np.abs(got[depth > 0] / depth[depth > 0] - 1).max() < 0.3 and not np.array_equal(got, depth)
    # known answers: one frustum, slices of 2 m centred at 1, 3, 5 m; well-observed neighbours interpolate, a thin one falls back
    one = tmp_path / "one"
    write_model(one, 4, 4, 4, 4, 2.0, [[(np.array([100, 100, 10.0]), np.array([1.0, 1.2, 2.0]))]])
    mo = DepthModel(one, 1)
    z = np.zeros((4, 4), np.float32)
    z[0, :] = (0.5, 1.0, 2.0, 3.0)        # below the first centre: slice 0's own multiplier; at the centre: 1.0; halfway: 1.1; slice 1's centre: 1.2
    z[1, :] = (3.5, 4.5, 5.5, 9.0)        # towards the thin slice 2: slice 1's own 1.2 (3.5: idx1 = 2 is thin), 4.5 and 5.5: slice 2's own 2.0; 9.0: clamped to slice 2
    out = mo.undistort(z)
    assert np.allclose(out[0], [0.5, 1.0, 2.0 * 1.1, 3.0 * 1.2], rtol=1e-6) and np.allclose(out[1], [3.5 * 1.2, 4.5 * 2.0, 5.5 * 2.0, 9.0 * 2.0], rtol=1e-6)
    assert not out[2:].any()
    # what is not a measurement passes through: +Inf lands in the last slice without a float -> int conversion out of range and stays
    # +Inf; NaN, -Inf and negative values are left alone like zeros (`z > 0` is false for them)
    z[2, :] = (np.inf, np.nan, -np.inf, -2.0)
    z[3, :] = (1e30, 3.0e38, 0.0, 1.0)
    out = mo.undistort(z)
    assert out[2, 0] == np.inf and np.isnan(out[2, 1]) and out[2, 2] == -np.inf and out[2, 3] == -2.0
    assert out[3, 0] == np.float32(1e30) * np.float32(2.0) and out[3, 1] == np.inf and out[3, 2] == 0.0 and out[3, 3] == 1.0
    # the model of the full-resolution sensor on half-size images (Calib360.h:116: downsampleParams(2))
    half = DepthModel(path, downsample=2)
    assert (half.width, half.height, half.bin_width, half.bin_height) == (W // 2, H // 2, bw // 2, bh // 2)
    d2 = depth[::2, ::2].copy()
    assert np.array_equal(half.undistort(d2), O.depth_model_undistort(O.depth_model_read(path, 2), d2))
    # errors: another image size, bins that do not divide, a missing / foreign / truncated file
    with pytest.raises(Rgbd360Error):
        m.undistort(depth[:-1])
    for bad in (lambda: DepthModel(path, 4), lambda: DepthModel(tmp_path / "none", 1)):
        with pytest.raises(Rgbd360Error):
            bad()
    (tmp_path / "foreign").write_bytes(b"something else\n" + b"\x00" * 64)
    (tmp_path / "short").write_bytes(open(path, "rb").read()[:500])
    for name in ("foreign", "short"):
        with pytest.raises(Rgbd360Error):
            DepthModel(tmp_path / name, 1)


def _probe_image():
    rng = np.random.default_rng(77)
    depth = rng.uniform(0.4, 9.5, size=(240, 320)).astype(np.float32)
    depth[rng.random((240, 320)) < 0.05] = 0.0
    return depth


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference's model files are read in place in the build container only")
def test_reference_model_files_in_place():
    """The eight files of Calibration/Intrinsics load (no byte left over: the restatement's parser asserts it), have the geometry
    Calib360 expects after downsampleParams(2) -- 320 x 240 images, 4 x 3-pixel bins, 80 x 80 of them, 2 m slices -- and correct a seeded
    depth image exactly like the numpy restatement; the committed fixture pins probes of model 1's result."""
    depth = _probe_image()
    G = json.load(open(GOLDEN))
    for k in range(1, 9):
        path = os.path.join(REF_MODELS, "distortion_model%d" % k)
        m = DepthModel(path, 2)
        assert (m.width, m.height, m.bin_width, m.bin_height, m.num_bins_x, m.num_bins_y, m.bin_depth) == (320, 240, 4, 3, 80, 80, 2.0)
        got = m.undistort(depth)
        if k in (1, 5):
            assert np.array_equal(got, O.depth_model_undistort(O.depth_model_read(path, 2), depth))
        g = G["model%d" % k]
        assert [float(np.float32(x)) for x in g["probes"]] == [float(got[r, c]) for r, c in G["probe_pixels"]]
        assert abs(float(got.astype(np.float64).sum()) - g["sum"]) < 1e-6 * g["sum"]


def test_golden_fixture_is_self_consistent():
    G = json.load(open(GOLDEN))
    assert len(G["probe_pixels"]) == 16 and all(len(G["model%d" % k]["probes"]) == 16 for k in range(1, 9))
    depth = _probe_image()
    for k in range(1, 9):      # corrections stay moderate (a few per cent near, tens of per cent at 9 m); pixels without a measurement stay zero
        for (r, c), v in zip(G["probe_pixels"], G["model%d" % k]["probes"]):
            assert (depth[r, c] == 0) == (v == 0) and (v == 0 or abs(v / depth[r, c] - 1) < 0.5)
