"""Generates tests/golden/pinhole_occ.json (run from the repository root: python tests/golden/make_golden_pinhole_occ.py).

PARITY UNPINNED, like make_golden.py's fixtures: the reference holds no vectors for these functions and cannot be built here.
The file pins the CPU oracle's restatement of the pinhole occlusion-aware passes (errorPhotoICP_Occ1/2, calcHessGrad_Occ1/2,
RPI.h:1107-2030) and of the salient-pixel-list mode (useSaliency(true), RPI.h:401-425, 590-690) on the seeded 320x240 sensor pair
synth.make_pinhole_pair(320, 240, seed=77) (inputs are regenerated from the seed in the tests; their CRC32 is recorded), in the
reference-faithful libm mode (math_mode 0) and the device-arithmetic mode (math_mode 1).  Data only.
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


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def probe_pose(T):
    """The rendered motion pushed 0.6 m along the optical axis: the warped image shrinks, target pixels collect up to four sources."""
    back = np.eye(4)
    back[2, 3] = 0.6
    return back @ T


def main():
    (rgbA, dA), (rgbB, dB), T, K = synth.make_pinhole_pair(320, 240, seed=77)
    out = {"K": list(K), "T_gt": T.tolist(), "crc32_inputs": [crc(rgbA), crc(dA), crc(rgbB), crc(dB)], "occ": {}, "salient": {}}
    P = probe_pose(T)
    for math_mode in (0, 1):
        ora = O.Oracle(n_pyr=3, math_mode=math_mode, reduce_mode=1, mask_seams=0)
        ora.set_camera(*K); ora.set_target(rgbA, dA); ora.set_source(rgbB, dB)
        for occ in (1, 2):
            for method in (0, 1, 2):
                st, pose = ora.align_pinhole(np.eye(4), method, occ)
                r = ora.result
                rec = {"status": st, "iters": list(r.iters)[:3], "pose": pose.astype(np.float64).tolist(),
                       "err_final": None if r.err_final != r.err_final else r.err_final, "sso": float(r.sso)}
                for name, pp in (("at_gt_level1", T), ("at_probe_level0", P)):
                    level = 1 if name == "at_gt_level1" else 0
                    e = ora.error_pinhole_occ(level, pp, method, occ)
                    H, g, Hd, gd, nvis = ora.hessgrad_pinhole_occ(level, pp, method, occ)
                    rec[name] = {"sum_photo": e[1], "sum_depth": e[2], "n_photo": e[3], "n_depth": e[4], "n_visible": nvis,
                                 "H64": Hd.tolist(), "g64": gd.tolist()}
                out["occ"]["math%d/occ%d/method%d" % (math_mode, occ, method)] = rec
        ora.use_saliency(True, 0.01)
        sal = {"list_len": [], "list_crc": []}
        for level in range(3):
            v = ora.salient_pixels(level)
            sal["list_len"].append(int(len(v))); sal["list_crc"].append(crc(v.astype(np.int32)))
        for method in (1, 2):
            e = ora.error_pinhole_salient(1, T, method)
            st, pose = ora.align_pinhole(np.eye(4), method, 0)
            sal["method%d" % method] = {"sum_photo": e[1], "sum_depth": e[2], "n_photo": e[3], "n_depth": e[4], "status": st,
                                        "iters": list(ora.result.iters)[:3], "pose": pose.astype(np.float64).tolist(),
                                        "err_final": ora.result.err_final}
        out["salient"]["math%d" % math_mode] = sal
    with open(os.path.join(HERE, "pinhole_occ.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote pinhole_occ.json", os.path.getsize(os.path.join(HERE, "pinhole_occ.json")), "bytes")


if __name__ == "__main__":
    main()
