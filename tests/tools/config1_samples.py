"""BASELINE.json configs[0] plumbing: RegisterPairRGBD360-style run on the reference's own sample frames, CPU only.

Reads /root/reference/samples/sphere_images_{1,10}.bin and Calibration/Extrinsics/Rt_0N.txt IN PLACE where the reference is
mounted (the build container); elsewhere (the GPU box) the same sensor images and calibration numbers come from the committed data
fixture tests/golden/sample_pair.npz (tests/golden/make_golden_samples.py; round 6, so that the HIP path sees the pair).  Restates
  * the Boost binary archive layout of Frame360::loadFrame (Frame360.h:231-266 + cvmat_serialization.h:23-55):
    a 45-byte archive header, then 8 x {RGB 8UC3, depth 16UC1 mm} cv::Mat records + a timestamp Mat, each record
    int32 cols, int32 rows, uint64 elemSize, uint64 cvType, raw bytes;
  * Frame360::stitchSphericalImage / stitchImage (Frame360.h:386-405, 1099-1148) with Calib360's fixed QVGA camera
    matrix (Calib360.h:74-77) and extrinsics (Calib360.h:122-131),
and runs the CPU oracle's dense spherical alignment on the stitched 1920x320 pair.  The PbMap / GICP stages of the
original app (Registration/RegisterPairRGBD360.cpp:94-142) and the CLAMS undistortion (Frame360.h:293) are third-party
and skipped.  Used by tests/test_config1_samples.py (runs only where /root/reference exists) and to (re)generate
tests/golden/config1_samples.json.
"""
from __future__ import annotations

import json
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
PI = 3.14159265359


def load_frame(path):
    """-> list of 8 (rgb HxWx3 uint8, depth HxW uint16)"""
    buf = open(path, "rb").read()
    off = 45                                       # "serialization::archive" header of boost::archive::binary_oarchive
    mats = []
    while off < len(buf):
        cols, rows = struct.unpack_from("<ii", buf, off)
        elem, cvtype = struct.unpack_from("<QQ", buf, off + 8)
        off += 24
        n = cols * rows * elem
        mats.append((cols, rows, elem, cvtype, buf[off:off + n]))
        off += n
    assert len(mats) == 17, len(mats)              # 8 x (rgb, depth) + timestamp
    out = []
    for s in range(8):
        c, r, e, t, raw = mats[2 * s]
        assert (e, t) == (3, 16)                   # CV_8UC3
        rgb = np.frombuffer(raw, np.uint8).reshape(r, c, 3)
        c2, r2, e2, t2, raw2 = mats[2 * s + 1]
        assert (e2, t2) == (2, 2) and (c2, r2) == (c, r)   # CV_16UC1
        out.append((rgb, np.frombuffer(raw2, np.uint16).reshape(r2, c2)))
    return out


FIXTURE = os.path.join(ROOT, "tests", "golden", "sample_pair.npz")      # the same data as arrays (tests/golden/make_golden_samples.py)
_fixture = None


def have_reference():
    return os.path.exists(os.path.join(REF, "samples", "sphere_images_1.bin"))


def fixture():
    global _fixture
    if _fixture is None:
        _fixture = dict(np.load(FIXTURE))
    return _fixture


def frames(idx, source=None):
    """The eight (rgb, depth) sensor images of sample frame 1 or 10: parsed from the reference's .bin in place where the reference is
    mounted (build container), else from the committed data fixture.  source: "reference" | "fixture" | None (= whichever exists)."""
    if source == "reference" or (source is None and have_reference()):
        return load_frame(os.path.join(REF, "samples", "sphere_images_%d.bin" % idx))
    z = fixture()
    return [(z["rgb_%d" % idx][s], z["depth_%d" % idx][s]) for s in range(8)]


def extrinsics(source=None):
    """Rt [8] (sensor -> rig, float64 as the text files give them)."""
    if source == "reference" or (source is None and have_reference()):
        return [np.loadtxt(os.path.join(REF, "Calibration", "Extrinsics", "Rt_0%d.txt" % (s + 1))) for s in range(8)]
    return list(fixture()["Rt"])


def write_bin(path, fr):
    """Eight (rgb, depth) sensor images in the layout load_frame reads (Frame360::serialize): what the C++ examples load."""
    with open(path, "wb") as f:
        f.write(b"\x16\x00\x00\x00\x00\x00\x00\x00serialization::archive" + b"\x00" * 15)
        for rgb, dep in fr:
            f.write(struct.pack("<iiQQ", rgb.shape[1], rgb.shape[0], 3, 16) + np.ascontiguousarray(rgb, np.uint8).tobytes())
            f.write(struct.pack("<iiQQ", dep.shape[1], dep.shape[0], 2, 2) + np.ascontiguousarray(dep, np.uint16).tobytes())
        f.write(struct.pack("<iiQQ", 0, 0, 0, 0))


def load_extrinsics(source=None):
    Rt = []
    for M in extrinsics(source):
        M = M.astype(np.float32)
        Rt.append(np.linalg.inv(M.astype(np.float32)).astype(np.float32))      # Rt_inv (Calib360.h:129)
    return Rt


def stitch(frames, Rt_inv):
    """Frame360::stitchSphericalImage (float32 arithmetic like the reference)."""
    K = np.array([[262.5, 0, 159.5], [0, 262.5, 119.5], [0, 0, 1]], np.float32)    # Calib360.h:74-77
    size_w, size_h = frames[0][0].shape[1], frames[0][0].shape[0]
    W = size_h * 8                                  # Frame360.h:391
    H = int(W * 0.5 * 60.0 / 180)
    sphereRGB = np.zeros((H, W, 3), np.uint8)
    sphereDepth = np.zeros((H, W), np.uint16)
    F = np.float32
    offsetPhi = F(H // 2 - 0.5)                     # sphereRGB.rows/2 is integer division
    offsetTheta = F(-(size_h * 15 // 2) + 0.5)
    angle_pixel = F(2 * PI / W)
    for s in range(8):
        rows = np.arange(H)
        phi = ((offsetPhi - rows.astype(F)) * angle_pixel).astype(F)
        cols = np.arange((7 - s) * size_h, (8 - s) * size_h)
        theta = ((cols.astype(F) + offsetTheta) * angle_pixel).astype(F)
        v0 = np.sin(phi).astype(F)[:, None] * np.ones((1, len(cols)), F)
        cp = np.cos(phi).astype(F)[:, None]
        v1 = (cp * np.sin(theta).astype(F)[None, :]).astype(F)
        v2 = (cp * np.cos(theta).astype(F)[None, :]).astype(F)
        R, t = Rt_inv[s][:3, :3], Rt_inv[s][:3, 3]
        p = [(R[i, 0] * v0 + R[i, 1] * v1 + R[i, 2] * v2 + t[i]).astype(F) for i in range(3)]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = (K[0, 0] * p[0] / p[2] + K[0, 2]).astype(F)
            v = (K[1, 1] * p[1] / p[2] + K[1, 2]).astype(F)
        ok = (u >= 0) & (u < size_w) & (v >= 0) & (v < size_h)
        ui = np.where(ok, u, 0).astype(np.int64)    # cv::Mat::at<>(v,u) truncates the float indices
        vi = np.where(ok, v, 0).astype(np.int64)
        rgb, dep = frames[s]
        rr, cc = np.nonzero(ok)
        sphereRGB[rr, cols[cc]] = rgb[vi[ok], ui[ok]]
        d = dep[vi[ok], ui[ok]].astype(np.float64)
        fac = np.sqrt(1 + ((u[ok].astype(np.float64) - K[0, 2]) / K[0, 0]) ** 2 + ((v[ok].astype(np.float64) - K[1, 2]) / K[1, 1]) ** 2)
        sphereDepth[rr, cols[cc]] = (d * fac).astype(np.uint16)      # double product truncated to unsigned short
    return sphereRGB, sphereDepth


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


ALIGNMENTS = [(0, 0), (1, 0), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2)]      # (method, occlusion) run on the sample pair; Occ1 sums both modalities only


def panoramas(source=None):
    """The two stitched 1920 x 320 panoramas from the C++ oracle stitcher (oracle/frame360_ref.cpp)."""
    from oracle import oracle as O
    Rt_inv = np.stack(load_extrinsics(source))
    out = []
    for idx in (1, 10):
        fr = frames(idx, source)
        out.append(O.stitch_sphere(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), Rt_inv))
    return out


def run(source=None, all_modes=False):
    """Panoramas from the C++ oracle stitcher (oracle/frame360_ref.cpp); the numpy stitcher above is an independent
    second restatement whose disagreement (numpy float32 sin/cos vs libm: a handful of boundary pixels) is reported."""
    from oracle import oracle as O
    Rt_inv = load_extrinsics(source)
    pano, mismatch = [], []
    for idx in (1, 10):
        fr = frames(idx, source)
        rgb8 = np.stack([f[0] for f in fr])
        d8 = np.stack([f[1] for f in fr])
        a, b = O.stitch_sphere(rgb8, d8, np.stack(Rt_inv))
        a2, b2 = stitch(fr, Rt_inv)
        mismatch.append(int((a != a2).any(-1).sum() + (b != b2).sum()))
        pano.append((a, b))
    frames_ = fr
    out = {"numpy_vs_cpp_stitch_mismatching_pixels": mismatch,"sensor_image_shape": list(frames_[0][0].shape), "panorama_shape": list(pano[0][0].shape),
           "crc32": {"rgb_1": crc(pano[0][0]), "depth_1": crc(pano[0][1]), "rgb_10": crc(pano[1][0]), "depth_10": crc(pano[1][1])},
           "valid_depth_fraction": [float((p[1] > 0).mean()) for p in pano]}
    for method, name in ((0, "PHOTO_CONSISTENCY"), (2, "PHOTO_DEPTH")):
        ora = O.Oracle(n_pyr=4, math_mode=0, reduce_mode=1)
        ora.set_target(*pano[0])
        ora.set_source(*pano[1])
        st, pose = ora.align360(np.eye(4), method)
        out[name] = {"status": st, "iters": list(ora.result.iters)[:4], "pose": pose.astype(np.float64).tolist(),
                     "err_final": ora.result.err_final, "sso": float(ora.result.sso)}
    if all_modes:
        # every (method, occlusion) the device tests run on this pair (tests/test_samples_gpu.py), in the reference's arithmetic
        # ("libm": math_mode 0, float32 accumulators) and in the device's ("device": math_mode 1, float64 accumulation)
        out["alignments"] = {}
        for method, occ in ALIGNMENTS:
            rec = {}
            for tag, mm, rm in (("libm", 0, 0), ("device", 1, 1)):
                ora = O.Oracle(n_pyr=4, math_mode=mm, reduce_mode=rm)
                ora.set_target(*pano[0])
                ora.set_source(*pano[1])
                st, pose = ora.align360(np.eye(4), method, occ)
                rec[tag] = {"status": st, "iters": list(ora.result.iters)[:4], "pose": pose.astype(np.float64).tolist(),
                            "err_final": ora.result.err_final, "sso": float(ora.result.sso)}
            out["alignments"]["m%d_o%d" % (method, occ)] = rec
    return out, pano


if __name__ == "__main__":
    out, _ = run(all_modes=True)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1 and sys.argv[1] == "--write":
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config1_samples.json"), "w"), indent=1)
