"""Generates the committed golden fixtures (run from the repository root: python tests/golden/make_golden.py).

PARITY UNPINNED: the reference (EduFdez/rgbd360) ships no golden vectors, tests or recorded poses for this path and
cannot be built here (MRPT / OpenCV / PCL / Eigen / Boost are absent).  These fixtures therefore pin the CPU oracle
(oracle/photo_icp_ref.cpp, the line-by-line restatement of the reference) against regressions; they are data only:
  pair_256x128.npz    the seeded synthetic input pair (uint8 RGB, uint16 mm depth) and its ground-truth pose
  oracle_256x128.json oracle outputs on that pair: CRC32 + probe pixels of every pyramid plane and LUT, the
                      accept/reject trace, final pose, Hessian and iteration counts per method, in the
                      reference-faithful libm mode (math_mode 0) and in the device-arithmetic mode (math_mode 1)
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rgbd360_amd import synth          # noqa: E402
from oracle import oracle as O         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = [(0, 0), (5, 7), (31, 100), (64, 128), (100, 31), (127, 255)]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    (rgbA, dA), (rgbB, dB), T = synth.make_pair(256, 128, seed=1234)
    np.savez_compressed(os.path.join(HERE, "pair_256x128.npz"), rgbA=rgbA, dA=dA, rgbB=rgbB, dB=dB, T_gt=T)
    out = {"n_pyr": 3, "planes": {}, "lut": {}, "runs": {}}
    ora = O.Oracle(n_pyr=3, math_mode=0, reduce_mode=1)
    ora.set_target(rgbA, dA)
    ora.set_source(rgbB, dB)
    for level in range(3):
        ora.prepare_level(level)
        rows, cols = ora.level_dims(level)
        for name in O.PLANES:
            p = ora.plane(name, level)
            probes = [float(p[min(r >> level, rows - 1), min(c >> level, cols - 1)]) for r, c in PROBES]
            out["planes"]["%s/%d" % (name, level)] = {"crc32": crc(p), "probes": probes}
        lut = ora.lut(level)
        out["lut"][str(level)] = {"crc32_valid_xyz": crc(lut[lut[:, 0] != -10000]),
                                  "n_valid": int((lut[:, 0] != -10000).sum())}
    for math_mode in (0, 1):
        ora.set_modes(math_mode, 1)
        for method in (0, 1, 2):
            st, pose = ora.align360(np.eye(4), method)
            tr = ora.trace()
            out["runs"]["math%d/method%d" % (math_mode, method)] = {
                "status": st,
                "iters": list(ora.result.iters)[:3],
                "pose": pose.astype(np.float64).tolist(),
                "err_final": ora.result.err_final,
                "sso": float(ora.result.sso),
                "hessian": np.asarray(list(ora.result.hessian), dtype=np.float64).reshape(6, 6).T.tolist(),
                "trace": [{"level": t["level"], "it": t["it"], "accepted": t["accepted"], "error": t["error"],
                           "new_error": t["new_error"], "n_valid": t["n_valid"]} for t in tr],
                "pose_err_vs_ground_truth": list(synth.pose_error(pose, T)),
            }
            e = ora.error(1, T, method)
            H, g, Hd, gd, nvis = ora.hessgrad(1, T, method)
            out["runs"]["math%d/method%d" % (math_mode, method)]["at_gt_level1"] = {
                "rms": e[0], "err2": e[1], "n_valid": e[2], "n_visible": nvis, "H64": Hd.tolist(), "g64": gd.tolist()}
    # occlusion-aware variants (sequential semantics of RPI.h:3232-4249) on the same pair with a pasted occluder
    (orgbA, odA), (orgbB, odB), _ = synth.add_occluder(((rgbA, dA), (rgbB, dB), T))
    occ = O.Oracle(n_pyr=3, math_mode=0, reduce_mode=1)
    occ.set_target(orgbA, odA)
    occ.set_source(orgbB, odB)
    out["occlusion"] = {}
    probe_pose = synth.occlusion_test_poses(T)[2]
    for math_mode in (0, 1):
        occ.set_modes(math_mode, 1)
        for occlusion, method in ((1, 2), (2, 0), (2, 1), (2, 2)):
            st, pose = occ.align360(np.eye(4), method, occlusion)
            rec = {"status": st, "iters": list(occ.result.iters)[:3], "pose": pose.astype(np.float64).tolist(),
                   "err_final": occ.result.err_final, "sso": float(occ.result.sso)}
            e = occ.error_occ(1, probe_pose, method, occlusion)
            H, g, Hd, gd, nvis = occ.hessgrad_occ(1, probe_pose, method, occlusion)
            rec["at_probe_level1"] = {"error": e[0], "sum_photo": e[1], "sum_depth": e[2], "n_photo": e[3], "n_depth": e[4],
                                      "n_visible": nvis, "H64": Hd.tolist(), "g64": gd.tolist()}
            out["occlusion"]["math%d/occ%d/method%d" % (math_mode, occlusion, method)] = rec
    # pinhole single-sensor path (RPI.h:4254-4512) on a seeded 320x240 sensor pair (inputs regenerated from the seed in the
    # tests: synth.make_pinhole_pair(320, 240, seed=77); their CRC32 is recorded here)
    (prgbA, pdA), (prgbB, pdB), pT, pK = synth.make_pinhole_pair(320, 240, seed=77)
    out["pinhole"] = {"K": list(pK), "T_gt": pT.tolist(), "crc32_inputs": [crc(prgbA), crc(pdA), crc(prgbB), crc(pdB)], "runs": {}}
    pin = O.Oracle(n_pyr=3, math_mode=0, reduce_mode=1, mask_seams=0)
    pin.set_camera(*pK)
    pin.set_target(prgbA, pdA)
    pin.set_source(prgbB, pdB)
    for math_mode in (0, 1):
        pin.set_modes(math_mode, 1)
        for method in (0, 1, 2):
            st, pose = pin.align_pinhole(np.eye(4), method)
            rec = {"status": st, "iters": list(pin.result.iters)[:3], "pose": pose.astype(np.float64).tolist(),
                   "err_final": None if pin.result.err_final != pin.result.err_final else pin.result.err_final,
                   "trace": [{"level": t["level"], "it": t["it"], "accepted": t["accepted"], "n_valid": t["n_valid"]} for t in pin.trace()]}
            e = pin.error_pinhole(1, pT, method)
            H, g, Hd, gd, nrows = pin.hessgrad_pinhole(1, pT, method)
            rec["at_gt_level1"] = {"sum_photo": e[1], "sum_depth": e[2], "n_photo": e[3], "n_depth": e[4], "n_rows": nrows,
                                   "H64": Hd.tolist(), "g64": gd.tolist()}
            out["pinhole"]["runs"]["math%d/method%d" % (math_mode, method)] = rec
    with open(os.path.join(HERE, "oracle_256x128.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
