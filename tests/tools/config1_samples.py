"""BASELINE.json configs[0] plumbing: RegisterPairRGBD360-style run on the reference's own sample frames, CPU only.

Reads /root/reference/samples/sphere_images_{1,10}.bin and Calibration/Extrinsics/Rt_0N.txt IN PLACE (the 6 MB of
reference data are never copied into this repository), restates
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


def load_extrinsics():
    Rt = []
    for s in range(8):
        M = np.loadtxt(os.path.join(REF, "Calibration", "Extrinsics", "Rt_0%d.txt" % (s + 1))).astype(np.float32)
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


def run():
    """Panoramas from the C++ oracle stitcher (oracle/frame360_ref.cpp); the numpy stitcher above is an independent
    second restatement whose disagreement (numpy float32 sin/cos vs libm: a handful of boundary pixels) is reported."""
    from oracle import oracle as O
    Rt_inv = load_extrinsics()
    pano, mismatch = [], []
    for idx in (1, 10):
        frames = load_frame(os.path.join(REF, "samples", "sphere_images_%d.bin" % idx))
        rgb8 = np.stack([f[0] for f in frames])
        d8 = np.stack([f[1] for f in frames])
        a, b = O.stitch_sphere(rgb8, d8, np.stack(Rt_inv))
        a2, b2 = stitch(frames, Rt_inv)
        mismatch.append(int((a != a2).any(-1).sum() + (b != b2).sum()))
        pano.append((a, b))
    out = {"numpy_vs_cpp_stitch_mismatching_pixels": mismatch,"sensor_image_shape": list(frames[0][0].shape), "panorama_shape": list(pano[0][0].shape),
           "crc32": {"rgb_1": crc(pano[0][0]), "depth_1": crc(pano[0][1]), "rgb_10": crc(pano[1][0]), "depth_10": crc(pano[1][1])},
           "valid_depth_fraction": [float((p[1] > 0).mean()) for p in pano]}
    for method, name in ((0, "PHOTO_CONSISTENCY"), (2, "PHOTO_DEPTH")):
        ora = O.Oracle(n_pyr=4, math_mode=0, reduce_mode=1)
        ora.set_target(*pano[0])
        ora.set_source(*pano[1])
        st, pose = ora.align360(np.eye(4), method)
        out[name] = {"status": st, "iters": list(ora.result.iters)[:4], "pose": pose.astype(np.float64).tolist(),
                     "err_final": ora.result.err_final, "sso": float(ora.result.sso)}
    return out, pano


if __name__ == "__main__":
    out, _ = run()
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1 and sys.argv[1] == "--write":
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config1_samples.json"), "w"), indent=1)
