"""Generates tests/golden/pbmap_planes.json (run from the repository root: python tests/golden/make_golden_pbmap.py).

PARITY UNPINNED (see oracle/pbmap_ref.py): the reference holds no recorded plane matches or PbMap poses.  The fixture is data
only: the planar regions the CPU oracle (oracle/frame360_ref.cpp) extracts from two seeded synthetic 512x256 frames
(0.3 m / 10 degree motion) as float32 plane records, and what the numpy restatement of RegisterPbMap returns on them per
registration mode -- it pins the oracle and the library against regressions."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rgbd360_amd import synth          # noqa: E402
from oracle import oracle as OR        # noqa: E402
from oracle import pbmap_ref as PB     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ("centroid", "normal", "d", "curvature", "count", "root", "area", "elongation", "ppal_dir")


def main():
    OR.build()
    W, H = 512, 256
    (_, dA), (_, dB), T = synth.make_pair(W, H, seed=5, trans=0.3, rot_deg=10.0)
    frames = []
    for d in (dA, dB):
        xyz = OR.sphere_cloud(d, 2)
        nrm, _ = OR.f360_normals(xyz, H, W, 0.05, 8.0, 1)
        planes = OR.f360_plane_segment(xyz, nrm, H, W, 40, 0.03, 0.05, 0.001, 1)[1]
        frames.append([{k: (np.asarray(p[k], np.float64).tolist() if k in ("centroid", "normal", "ppal_dir") else
                            (int(p[k]) if k in ("count", "root") else float(p[k]))) for k in FIELDS} for p in planes])
    out = {"width": W, "height": H, "T_gt": np.asarray(T).tolist(), "frames": frames, "runs": {}}
    for mode in (0, 1, 2, 3):
        for mmp in (0, 6):
            r = PB.register_planes(frames[0], frames[1], mmp, mode)
            out["runs"]["mode%d/max%d" % (mode, mmp)] = {
                "status": r["status"], "match": {str(k): v for k, v in r["match"].items()}, "area_matched": r["area_matched"],
                "pose": np.asarray(r["pose"]).tolist(), "info": np.asarray(r["info"]).tolist()}
    with open(os.path.join(HERE, "pbmap_planes.json"), "w") as f:
        json.dump(out, f, indent=0)
    print({k: (v["status"], len(v["match"])) for k, v in out["runs"].items()})


if __name__ == "__main__":
    main()
