"""BASELINE.json configs[0] (plumbing, CPU only): the reference's own sample frames through the restated .bin reader,
the restated spherical stitcher and the CPU oracle.  Runs only where /root/reference is mounted (the build container);
the GPU box never sees the reference.  Expected values: tests/golden/config1_samples.json (tools/config1_samples.py --write)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLES = "/root/reference/samples/sphere_images_1.bin"


@pytest.mark.skipif(not os.path.exists(SAMPLES), reason="reference samples are only present in the build container")
def test_sample_pair_stitches_and_aligns_like_the_committed_record():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config1_samples as c1
    from rgbd360_amd import synth
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config1_samples.json")))
    out, pano = c1.run()
    assert out["sensor_image_shape"] == [240, 320, 3] and out["panorama_shape"] == [320, 1920, 3]
    assert out["crc32"] == gold["crc32"]                       # byte-exact stitched panoramas (C++ oracle stitcher)
    assert max(out["numpy_vs_cpp_stitch_mismatching_pixels"]) <= 30      # the independent numpy stitcher agrees (of 614,400 px)
    # the product's host-side .bin reader returns the same 8 x (rgb, depth) as the test-side parser
    from rgbd360_amd.register import load_frame_bin
    rgb8, d8 = load_frame_bin(SAMPLES)
    fr = c1.load_frame(SAMPLES)
    assert all(np.array_equal(rgb8[s], fr[s][0]) and np.array_equal(d8[s], fr[s][1]) for s in range(8))
    assert all(f > 0.5 for f in out["valid_depth_fraction"])
    for name in ("PHOTO_CONSISTENCY", "PHOTO_DEPTH"):
        assert out[name]["status"] == gold[name]["status"] == 0
        assert out[name]["iters"] == gold[name]["iters"]
        rot, trans = synth.pose_error(np.array(out[name]["pose"]), np.array(gold[name]["pose"]))
        assert rot < 1e-6 and trans < 1e-6
    # sanity of the recovered motion between frames 1 and 10 of the sample sequence: a hand-held rig moved by a few
    # decimetres, rotated by about a degree
    T = np.array(out["PHOTO_DEPTH"]["pose"])
    rot, trans = synth.pose_error(T, np.eye(4))
    assert 0.05 < trans < 0.6 and rot < 0.1


@pytest.mark.skipif(not os.path.exists(SAMPLES), reason="reference samples are only present in the build container")
def test_sample_pair_plane_registration_reports_an_unobservable_translation():
    """RegisterPairRGBD360.cpp:94-110 runs RegisterPbMap on this very pair.  With the reference's PCL parameters
    (Frame360.h:949-977: depth-change factor 0.02, smoothing 8, 80 inliers, 0.0398 rad, 0.02 m) applied to the stitched panorama
    -- the reference segments per sensor, on clouds down-sampled by 2 and smoothed by pcl::FastBilateralFilter (Frame360.h:40-41,
    479-499; third-party, not built) -- only the floor and the ceiling of the raw sensor data survive the segmentation: the matcher pairs a dozen of their pieces, every matched normal is
    (anti)parallel to the up axis, and the pose fit correctly reports that the translation is not observable (status 2) instead
    of inventing one -- the dense alignment then has to start from the identity, as tools/config1_samples.py does."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config1_samples as c1
    from oracle import oracle as O
    from rgbd360_amd import pbmap
    _, pano = c1.run()
    H, W = pano[0][1].shape
    frames = []
    for _, d in pano:
        xyz = O.sphere_cloud(d, 2)
        nrm, _w = O.f360_normals(xyz, H, W, 0.02, 8.0, 1)
        frames.append(O.f360_plane_segment(xyz, nrm, H, W, 80, 0.0398, 0.02, 0.0013, 1, max_planes=1024)[1])
    assert min(len(f) for f in frames) >= 20
    for mode in (pbmap.ODOMETRY_6DoF, pbmap.PLANAR_ODOMETRY_3DoF):
        r = pbmap.register_planes(frames[0], frames[1], 25, mode, pbmap.default_params(True))
        assert r["status"] == 2 and len(r["match"]) >= 8
        up = [abs(float(frames[0][i]["normal"][0])) for i in r["match"]]
        assert min(up) > 0.98                                   # floor / ceiling pieces only
        assert np.array_equal(r["pose"], np.eye(4, dtype=np.float32))
