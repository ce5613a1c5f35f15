"""How the boundary pixels of a label image fall on the waves of k_f360_hull_extremes (host-side count; run on the GPU box):
   python tools/hull_entries.py [width [angular_threshold [min_inliers]]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP, Frame360Stages
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ANG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
MIN_INLIERS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dA = synth.make_pair(W, W // 2, seed=5)[0][1]
st = Frame360Stages(RegisterPhotoICP())
out = st.frame_planes(dA, convention=2, angular_threshold=ANG, min_inliers=MIN_INLIERS, max_curvature=0.0013, max_planes=4096)
L = out["labels"]
rows, cols = L.shape
n = rows * cols
print("planes", len(out["planes"]), "labelled pixels", int((L >= 0).sum()), "of", n, "distinct labels", len(np.unique(L[L >= 0])))
d = np.zeros_like(L, bool)
d[:, 1:] |= L[:, 1:] != L[:, :-1]; d[:, :-1] |= L[:, :-1] != L[:, 1:]
d[1:, :] |= L[1:, :] != L[:-1, :]; d[:-1, :] |= L[:-1, :] != L[1:, :]
d[0, :] = d[-1, :] = True; d[:, 0] = d[:, -1] = True
# the kernel walks pixels whose label has a SLOT (a region above min_inliers that became a plane candidate): approximate by labels of returned planes
keep = np.zeros(int(L.max()) + 2, bool)
for p in out["planes"]:
    pass
lab, cnt = np.unique(L[L >= 0], return_counts=True)
keep[lab[cnt >= MIN_INLIERS]] = True
bnd = d & (L >= 0) & keep[np.maximum(L, 0)]
print("boundary pixels of slotted regions:", int(bnd.sum()), "= %.2f %% of the image" % (100.0 * bnd.sum() / n))
per_chunk = bnd.reshape(-1, 64).sum(1)                    # 64-pixel stretches in pixel order
nchunk = per_chunk.size
grid = min(256, (nchunk + 16 * 8 - 1) // (16 * 8))
wave_tot = np.zeros(grid * 16, np.int64)
distinct = []
for c in range(8):
    for b in range(grid):
        for w in range(16):
            k = (c * grid + b) * 16 + w
            if k < nchunk:
                wave_tot[b * 16 + w] += per_chunk[k]
print("grid", grid, "entries per wave: mean %.1f  max %d  p99 %d  waves with > 64: %d, > 128: %d" %
      (wave_tot.mean(), wave_tot.max(), np.percentile(wave_tot, 99), (wave_tot > 64).sum(), (wave_tot > 128).sum()))
blk = wave_tot.reshape(grid, 16)
print("per block: max of its waves, mean %.1f max %d; sum over its waves mean %.0f max %d" % (blk.max(1).mean(), blk.max(1).max(), blk.sum(1).mean(), blk.sum(1).max()))
print("full stretches (64 of 64):", int((per_chunk == 64).sum()), " stretches with any:", int((per_chunk > 0).sum()), "of", nchunk)
