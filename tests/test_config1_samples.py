"""BASELINE.json configs[0] (plumbing, CPU only): the reference's own sample frames through the restated .bin reader,
the restated spherical stitcher and the CPU oracle.  Where /root/reference is mounted (the build container) the frames are parsed
from its files in place; elsewhere they come from the committed data fixture tests/golden/sample_pair.npz, which the first test
below proves equal to those files.  Expected values: tests/golden/config1_samples.json (tests/tools/config1_samples.py --write).
The same pair on the DEVICE: tests/test_samples_gpu.py."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLES = "/root/reference/samples/sphere_images_1.bin"


@pytest.mark.skipif(not os.path.exists(SAMPLES), reason="reference samples are only present in the build container")
def test_data_fixture_holds_the_reference_sample_frames_and_extrinsics():
    """tests/golden/sample_pair.npz (what the GPU box sees) = the sensor images of samples/sphere_images_{1,10}.bin and the numbers of
    Calibration/Extrinsics/Rt_0N.txt, value for value; written back in the archive layout it is the reference's file except for the
    15 version bytes of the Boost header no reader looks at."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import config1_samples as c1
    for idx in (1, 10):
        ref, fix = c1.frames(idx, "reference"), c1.frames(idx, "fixture")
        assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, fix))
    assert all(np.array_equal(a, b) for a, b in zip(c1.extrinsics("reference"), c1.extrinsics("fixture")))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.bin")
        c1.write_bin(path, c1.frames(1, "fixture"))
        a, b = open(path, "rb").read(), open(SAMPLES, "rb").read()
        assert len(a) == len(b) and a[:30] == b[:30] and a[45:] == b[45:]


def test_sample_pair_fixture_stitches_to_the_committed_checksums():
    """Runs everywhere (no reference needed): the oracle's stitcher on the data fixture reproduces the CRCs recorded from the
    reference's files, and the alignment record carries the iteration counts the device tests expect."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import config1_samples as c1
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config1_samples.json")))
    pano = c1.panoramas("fixture")
    got = {"rgb_1": c1.crc(pano[0][0]), "depth_1": c1.crc(pano[0][1]), "rgb_10": c1.crc(pano[1][0]), "depth_10": c1.crc(pano[1][1])}
    assert got == gold["crc32"]
    assert gold["alignments"]["m0_o0"]["libm"]["iters"] == gold["alignments"]["m2_o0"]["device"]["iters"] == [10, 10, 10, 7]
    assert set(gold["alignments"]) == {"m%d_o%d" % mo for mo in c1.ALIGNMENTS}


@pytest.mark.skipif(not os.path.exists(SAMPLES), reason="reference samples are only present in the build container")
def test_sample_pair_stitches_and_aligns_like_the_committed_record():
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
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
    of inventing one -- the dense alignment then has to start from the identity, as tests/tools/config1_samples.py does."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
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


def _pinhole_cloud(depth_mm, min_depth=0.3, max_depth=10.0):
    """CloudRGBD::getPointCloud (OpenNI2_Grabber/FrameRGBD/CloudRGBD.h:107-166): focal 525 * W / 640, centre (W/2 - 0.5, H/2 - 0.5)."""
    H, W = depth_mm.shape
    f = np.float32(525.0 * (W / 640.0))
    z = (np.float32(0.001) * depth_mm.astype(np.float32)).astype(np.float32)
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    xyz = np.stack([(u - np.float32(W / 2 - 0.5)) * z / f, (v - np.float32(H / 2 - 0.5)) * z / f, z], -1).astype(np.float32)
    xyz[~((z >= min_depth) & (z <= max_depth))] = np.nan
    return xyz


def _downsample_median(xyz, step=2, min_depth=0.3, max_depth=10.0):
    """DownsampleRGBD::downsamplePointCloud (OpenNI2_Grabber/FrameRGBD/DownsampleRGBD.h:209-300): per coordinate, element n/2 of the
    sorted valid values of every step x step block."""
    H, W, _ = xyz.shape
    b = xyz.reshape(H // step, step, W // step, step, 3).transpose(0, 2, 1, 3, 4).reshape(H // step, W // step, step * step, 3)
    ok = np.isfinite(b[..., 0]) & (b[..., 2] > min_depth) & (b[..., 2] < max_depth)
    n = ok.sum(-1)
    s = np.sort(np.where(ok[..., None], b, np.inf), axis=2)
    out = np.take_along_axis(s, np.minimum(n // 2, step * step - 1)[..., None, None].repeat(3, -1), axis=2)[:, :, 0, :]
    out[n == 0] = np.nan
    return out.astype(np.float32)


@pytest.mark.skipif(not os.path.exists(SAMPLES), reason="reference samples are only present in the build container")
def test_sample_pair_per_sensor_planes_with_the_bilateral_filter():
    """The reference's route for the rig's planes (Frame360.h:40-41, 479-499, 949-1075): per sensor, pinhole cloud down-sampled by 2,
    pcl::FastBilateralFilter (10 px, 0.05 m), normal map, segmentation, planes moved into the rig frame by Rt -- here with the
    oracle's restatements and the library's matcher.  With the filter the walls of the raw sensor data appear (without it only
    floor and ceiling do, see the test above); the matcher pairs floor, ceiling and the pieces of one wall direction of frames
    1 and 10, and the fit reports that the translation along that wall is not observable (two independent normal directions
    only: conditioning > 100, status 2), as ConsistencyTest's conditioning test would."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import config1_samples as c1
    from oracle import oracle as O
    from rgbd360_amd import pbmap
    Rt = [np.loadtxt("/root/reference/Calibration/Extrinsics/Rt_0%d.txt" % (s + 1)) for s in range(8)]
    frames = []
    walls_raw = walls_filtered = 0
    for idx in (1, 10):
        planes_rig = []
        for s, (_, depth) in enumerate(c1.load_frame("/root/reference/samples/sphere_images_%d.bin" % idx)):
            cloud = O.sensor_cloud(depth, 2, 0.3, 10.0)                       # (numpy twins above: same cloud to 1 ulp)
            assert np.nanmax(np.abs(cloud - _downsample_median(_pinhole_cloud(depth)))) < 1e-6
            H, W, _ = cloud.shape
            for filtered in (False, True):
                xyz = O.fast_bilateral(cloud, H, W, 10.0, 0.05) if filtered else np.ascontiguousarray(cloud).reshape(-1, 3)
                nrm, _w = O.f360_normals(xyz, H, W, 0.02, 8.0, 0)
                planes = O.f360_plane_segment(xyz, nrm, H, W, 40, 0.0398, 0.02, 0.0013, 0, max_planes=512)[1]
                n_walls = sum(p["count"] for p in planes if abs(p["normal"][2]) > 0.9)     # inliers of planes facing the sensor
                if filtered:
                    walls_filtered += n_walls
                else:
                    walls_raw += n_walls
            R, t = Rt[s][:3, :3], Rt[s][:3, 3]
            for p in planes:                                                  # (filtered planes) sensor frame -> rig frame
                n, c = R @ p["normal"].astype(np.float64), R @ p["centroid"].astype(np.float64) + t
                if n @ c > 0:
                    n = -n
                q = dict(p)
                q.update(normal=n.astype(np.float32), centroid=c.astype(np.float32), d=np.float32(-n @ c),
                         ppal_dir=(R @ p["ppal_dir"].astype(np.float64)).astype(np.float32))
                planes_rig.append(q)
        frames.append(planes_rig)
    print("wall inliers without / with the filter:", walls_raw, walls_filtered)
    assert walls_filtered >= 20000 and walls_filtered >= 1.5 * walls_raw, (walls_raw, walls_filtered)
    r = pbmap.register_planes(frames[0], frames[1], 25, pbmap.ODOMETRY_6DoF, pbmap.default_params(True))
    assert len(r["match"]) >= 12
    matched = [frames[0][i] for i in r["match"]]
    horizontal = [p for p in matched if abs(p["normal"][0]) > 0.95]           # x is up in the rig frame: floor / ceiling
    walls = [p for p in matched if abs(p["normal"][0]) < 0.3]
    assert len(horizontal) >= 6 and len(walls) >= 3
    for i, j in r["match"].items():                                           # same planes in both frames: normals within a few degrees,
        assert float(frames[0][i]["normal"] @ frames[1][j]["normal"]) > 0.99  # offsets within the motion between the frames
        assert abs(float(frames[0][i]["d"]) - float(frames[1][j]["d"])) < 0.1
    assert r["status"] == 2
    # Frame360::mergePlanes on the rig-frame pieces (the floor / ceiling / wall pieces of neighbouring sensors become single planes):
    # fewer planes, the same verdict
    merged = [pbmap.merge_planes(f) for f in frames]
    assert all(3 <= len(m) < len(f) for m, f in zip(merged, frames))
    rm = pbmap.register_planes(merged[0], merged[1], 25, pbmap.ODOMETRY_6DoF, pbmap.default_params(True))
    print("planes per frame", [len(f) for f in frames], "merged", [len(m) for m in merged], "matched", len(r["match"]), len(rm["match"]), "status", rm["status"])
    assert len(rm["match"]) >= 3 and rm["status"] in (0, 2)
