"""Second, independent restatement of the reference path in vectorised numpy (TEST CODE, small images only).

Written separately from oracle/photo_icp_ref.cpp (different structure: whole-image array operations instead of
per-pixel loops) so that a transcription slip in either shows up as a disagreement.  Follows the same reference
lines ("RPI.h" = include/RegisterPhotoICP.h of EduFdez/rgbd360):
  gray / depth conversion RPI.h:485-486, 316-319 ; pyramids RPI.h:292-354 ; gradients RPI.h:365-398 ;
  seam mask RPI.h:4538-4549 ; LUT RPI.h:4554-4587 ; error pass RPI.h:2545-2739 ; H,g pass RPI.h:2745-3228.
All per-pixel arithmetic is float32 with the reference's operation order (numpy float32 ops round after every
operation, like the reference's SSE2 build).
"""
import numpy as np

F = np.float32
PI = 3.14159265359          # Miscellaneous.h:44
INVALID = F(-10000.0)


def rgb_to_gray(rgb):
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    v = (4899 * r + 9617 * g + 1868 * b + 8192) >> 14
    return v.astype(F) * F(1.0 / 255)


def depth_to_f32(d):
    return d.astype(F) * F(0.001) if d.dtype == np.uint16 else d.astype(F)


def pyr_down(img):
    rows, cols = img.shape
    dr, dc = rows // 2, cols // 2
    p = np.pad(img, 2, mode="reflect")                    # BORDER_REFLECT_101
    # horizontal 1-4-6-4-1 at even columns, for every (padded) row
    c = 2 * np.arange(dc) + 2
    h = p[:, c] * F(6) + (p[:, c - 1] + p[:, c + 1]) * F(4) + p[:, c - 2] + p[:, c + 2]
    r = 2 * np.arange(dr) + 2
    v = h[r] * F(6) + (h[r - 1] + h[r + 1]) * F(4) + h[r - 2] + h[r + 2]
    return (v * F(1.0 / 256)).astype(F)


def pyr_down_depth(img, dmin, dmax):
    rows, cols = img.shape
    dr, dc = rows // 2, cols // 2
    blk = img[:2 * dr, :2 * dc].reshape(dr, 2, dc, 2).transpose(0, 2, 1, 3).reshape(dr, dc, 4)   # (0,0),(0,1),(1,0),(1,1)
    valid = (blk > F(dmin)) & (blk < F(dmax))
    acc = np.zeros((dr, dc), F)
    for k in range(4):                                    # sequential float32 adds in the reference's order
        acc = np.where(valid[..., k], acc + blk[..., k], acc).astype(F)
    n = valid.sum(-1)
    out = np.zeros((dr, dc), F)
    nz = n > 0
    out[nz] = acc[nz] / n[nz].astype(F)
    return out


def gradient_xy(img):
    gx = np.zeros_like(img)
    gy = np.zeros_like(img)
    c = img[1:-1, 1:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        for g, plus, minus in ((gx, img[1:-1, 2:], img[1:-1, :-2]), (gy, img[2:, 1:-1], img[:-2, 1:-1])):
            mono = ((c > plus) & (c < minus)) | ((c < plus) & (c > minus))
            val = F(2.0) / (F(1) / (plus - c) + F(1) / (c - minus))
            g[1:-1, 1:-1] = np.where(mono, val, F(0)).astype(F)
    return gx, gy


def mask_seams(img):
    w = img.shape[1] // 8
    for s in range(1, 8):
        img[:, s * w - 1: s * w + 1] = 0
    return img


class Frames:
    def __init__(self, rgb_t, d_t, rgb_s, d_s, n_pyr=3, dmin=0.3, dmax=6.0, mask=True):
        self.n_pyr, self.dmin, self.dmax = n_pyr, F(dmin), F(dmax)
        self.gray_t = [rgb_to_gray(rgb_t)]
        self.gray_s = [rgb_to_gray(rgb_s)]
        self.dep_t = [depth_to_f32(d_t)]
        self.dep_s = [depth_to_f32(d_s)]
        for _ in range(1, n_pyr):
            self.gray_t.append(pyr_down(self.gray_t[-1]))
            self.gray_s.append(pyr_down(self.gray_s[-1]))
            self.dep_t.append(pyr_down_depth(self.dep_t[-1], dmin, dmax))
            self.dep_s.append(pyr_down_depth(self.dep_s[-1], dmin, dmax))
        self.gx, self.gy, self.dgx, self.dgy = [], [], [], []
        for l in range(n_pyr):
            a, b = gradient_xy(self.gray_t[l])
            c, d = gradient_xy(self.dep_t[l])
            if mask:
                a, b, c, d = (mask_seams(x) for x in (a, b, c, d))
            self.gx.append(a); self.gy.append(b); self.dgx.append(c); self.dgy.append(d)

    def lut(self, level):
        d = self.dep_s[level]
        rows, cols = d.shape
        angle_res = F(2 * PI / cols)
        theta = (np.arange(cols).astype(F) * angle_res).astype(F)
        half = F(0.5 * rows - 0.5)
        phi = ((half - np.arange(rows).astype(F)) * angle_res).astype(F)
        st, ct = np.sin(theta).astype(F), np.cos(theta).astype(F)     # float32 in -> float32 out (sinf / cosf)
        sp, cp = np.sin(phi).astype(F), np.cos(phi).astype(F)
        valid = (self.dmin < d) & (d < self.dmax)
        x = d * sp[:, None]
        y = (-d * cp[:, None]) * st[None, :]
        z = (-d * cp[:, None]) * ct[None, :]
        x = np.where(valid, x, INVALID).astype(F)
        return x.ravel(), y.astype(F).ravel(), z.astype(F).ravel(), valid.ravel()


def huber(e, k):
    ea = np.abs(e)
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.sqrt((F(2) * k * ea - k * k).astype(F)).astype(F) / ea
    return np.where(ea < k, F(1), w).astype(F)


def warp(fr, level, pose):
    """Front end shared by both passes.  pose: 4x4 float (p_trg = R p_src + t)."""
    x, y, z, valid = fr.lut(level)
    rows, cols = fr.dep_s[level].shape
    R = np.asarray(pose, F)[:3, :3]
    t = np.asarray(pose, F)[:3, 3]
    X = ((R[0, 0] * x + R[0, 1] * y) + R[0, 2] * z) + t[0]
    Y = ((R[1, 0] * x + R[1, 1] * y) + R[1, 2] * z) + t[1]
    Z = ((R[2, 0] * x + R[2, 1] * y) + R[2, 2] * z) + t[2]
    dist = np.sqrt(((X * X + Y * Y) + Z * Z).astype(F)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        dist_inv = (F(1) / dist).astype(F)
        angle_res = F(2 * PI / cols)
        angle_res_inv = F(1) / angle_res
        half = F(0.5 * rows - 0.5)
        phi = np.arcsin((X * dist_inv).astype(F)).astype(F)
        theta = (np.arctan2(Y, Z).astype(F).astype(np.float64) + PI).astype(F)
        fr_ = (half - phi * angle_res_inv).astype(F)
        fc_ = (theta * angle_res_inv).astype(F)
        rnd = lambda v: np.where(np.isfinite(v), np.sign(v) * np.floor(np.abs(v.astype(np.float64)) + 0.5), -1).astype(np.int64)
        r, c = rnd(fr_), rnd(fc_)
    vis = valid & (r >= 0) & (r < rows) & (c < cols) & (c >= 0)
    return dict(X=X, Y=Y, Z=Z, dist=dist, dist_inv=dist_inv, r=r, c=c, vis=vis, rows=rows, cols=cols,
                angle_res_inv=angle_res_inv)


def error_and_hg(fr, level, pose, method, sigma_p=F(6.0 / 255), sigma_d=F(0.2), thr_p=F(0.01), thr_d=F(0.01)):
    """Returns (err2, n_valid, H 6x6 float64-accumulated, g, n_visible) of the reference's two passes at `pose`."""
    w = warp(fr, level, pose)
    vis = w["vis"]
    idx = np.nonzero(vis)[0]
    r, c = w["r"][idx], w["c"][idx]
    X, Y, Z, dist, dinv = (w[k][idx] for k in ("X", "Y", "Z", "dist", "dist_inv"))
    k = w["angle_res_inv"]
    # jacobianProj23 * jacobianT36 (RPI.h:2994-3026)
    with np.errstate(invalid="ignore", divide="ignore"):
        z_inv = F(1) / Z
        z_inv2 = z_inv * z_inv
        D = F(1) / (F(1) + Y * Y * z_inv2) * k
        a1 = D * z_inv
        a2 = -Y * z_inv2 * D
        dinv2 = dinv * dinv
        xd = X * dinv2
        Da = F(1) / np.sqrt((F(1) - X * xd).astype(F)).astype(F) * k
        b0 = -Da * dinv * (F(1) - X * xd)
        b1 = Da * (xd * Y * dinv)
        b2 = Da * (xd * Z * dinv)
    zero = np.zeros_like(a1)
    Jw0 = np.stack([zero, a1, a2, a1 * (-Z) + a2 * Y, a2 * (-X), a1 * X], 1).astype(F)
    Jw1 = np.stack([b0, b1, b2, b1 * (-Z) + b2 * Y, b0 * Z + b2 * (-X), b0 * (-Y) + b1 * X], 1).astype(F)
    err2, nvalid = 0.0, 0
    H = np.zeros((6, 6))
    g = np.zeros(6)
    photo_skip = np.zeros(len(idx), bool)
    if method in (0, 2):
        gx, gy = fr.gx[level][r, c], fr.gy[level][r, c]
        nonsal = (np.abs(gx) < thr_p) & (np.abs(gy) < thr_p)
        photo_skip = nonsal
        ok = ~nonsal
        diff = (fr.gray_t[level][r, c] - fr.gray_s[level].ravel()[idx]).astype(F)
        wh = huber(diff, sigma_p)
        # error pass: double weight (RPI.h:2559-2562)
        werr = (wh.astype(np.float64) * (1.0 / np.float64(sigma_p)) * diff.astype(np.float64)).astype(F)
        err2 += float(np.sum((werr[ok] * werr[ok]).astype(F).astype(np.float64)))
        nvalid += int(ok.sum())
        wf = (wh * F(1.0 / np.float64(sigma_p))).astype(F)
        res = (wf * diff).astype(F)
        J = ((wf * gx)[:, None] * Jw0 + (wf * gy)[:, None] * Jw1).astype(F)
        Jo = J[ok].astype(np.float64)
        # float32 products, float64 accumulation
        H += np.einsum("ni,nj->ij", J[ok], J[ok], dtype=np.float64)
        g += Jo.T @ res[ok].astype(np.float64)
    if method in (1, 2):
        d2 = fr.dep_t[level][r, c]
        dgx, dgy = fr.dgx[level][r, c], fr.dgy[level][r, c]
        nonsal = (np.abs(dgx) < thr_d) & (np.abs(dgy) < thr_d)
        ok = np.isfinite(d2) & ~nonsal
        if method == 2:
            ok &= ~photo_skip
        with np.errstate(invalid="ignore", divide="ignore"):
            diff = (d2 - dist).astype(F)
            sd = (sigma_d * d2).astype(F)
            wd = (huber(diff, sd) / sd).astype(F)
            werr = (wd * diff).astype(F)
        err2 += float(np.sum((werr[ok] * werr[ok]).astype(F).astype(np.float64)))
        nvalid += int(ok.sum())
        n0, n1, n2 = X * dinv, Y * dinv, Z * dinv
        nJ = np.stack([n0, n1, n2, n1 * (-Z) + n2 * Y, n0 * Z + n2 * (-X), n0 * (-Y) + n1 * X], 1).astype(F)
        with np.errstate(invalid="ignore"):
            J = (wd[:, None] * ((dgx[:, None] * Jw0 + dgy[:, None] * Jw1) - nJ)).astype(F)
        H += np.einsum("ni,nj->ij", J[ok], J[ok], dtype=np.float64)
        g += J[ok].astype(np.float64).T @ werr[ok].astype(np.float64)
    return err2, nvalid, H, g, int(vis.sum())


# ---- occlusion-aware error passes (sequential semantics of RPI.h:3232-3367 and 3720-3856) -----------------------------
def _prefix_maxima(target, key, order_index):
    """For candidates landing on `target` pixels, visited in `order_index` order: flag those whose key is >= every earlier
    key on the same target (the ones that pass `if (buf > 0 && key < buf) continue; buf = key`), and flag the final owner of
    each target's buffer (the last flagged one)."""
    order = np.lexsort((order_index, target))
    t, k = target[order], key[order]
    pm = np.ones(len(order), bool)
    owner = np.zeros(len(order), bool)
    start = 0
    n = len(order)
    bounds = np.nonzero(np.diff(t))[0] + 1
    for end in list(bounds) + [n]:
        if end - start == 1:
            owner[start] = True
        else:
            run = np.maximum.accumulate(k[start:end])
            prev = np.concatenate(([F(0)], run[:-1]))
            pm[start:end] = k[start:end] >= prev
            owner[start + np.nonzero(pm[start:end])[0][-1]] = True
        start = end
    out_pm = np.zeros(n, bool); out_owner = np.zeros(n, bool)
    out_pm[order] = pm; out_owner[order] = owner
    return out_pm, out_owner


def occ_error(fr, level, pose, method, occ, sigma_p=F(6.0 / 255), sigma_d=F(0.2), thr_p=F(0.01), thr_d=F(0.01),
              thr_outlier=F(0.3)):
    """(sum photo, sum depth, n photo, n depth) of errorPhotoICP_sphereOcc1 (occ = 1) / ...Occ2 (occ = 2)."""
    w = warp(fr, level, pose)
    idx = np.nonzero(w["vis"])[0]
    r, c = w["r"][idx], w["c"][idx]
    dist, dinv = w["dist"][idx], w["dist_inv"][idx]
    d2 = fr.dep_t[level][r, c]
    if occ == 2:                                   # depth-outlier gate before the z-buffer (RPI.h:3788-3791)
        with np.errstate(invalid="ignore"):
            keep = ~(np.abs((d2 - dist).astype(F)) > thr_outlier)
        idx, r, c, dist, dinv, d2 = idx[keep], r[keep], c[keep], dist[keep], dinv[keep], d2[keep]
    target = r * w["cols"] + c
    pm, owner = _prefix_maxima(target, dinv, idx)
    gx, gy = fr.gx[level][r, c], fr.gy[level][r, c]
    sal_p = ~((np.abs(gx) < thr_p) & (np.abs(gy) < thr_p))
    dgx, dgy = fr.dgx[level][r, c], fr.dgy[level][r, c]
    sal_d = ~((np.abs(dgx) < thr_d) & (np.abs(dgy) < thr_d))
    diff_p = (fr.gray_t[level][r, c] - fr.gray_s[level].ravel()[idx]).astype(F)
    wp = (huber(diff_p, sigma_p).astype(np.float64) * (1.0 / np.float64(sigma_p)) * diff_p.astype(np.float64)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        diff_d = (d2 - dist).astype(F)
        sd = (sigma_d * d2).astype(F)
        wd = ((huber(diff_d, sd).astype(np.float64) / sd.astype(np.float64)) * diff_d.astype(np.float64)).astype(F)
    use_p = method in (0, 2)
    use_d = method in (1, 2)
    ok_p = sal_p if use_p else np.zeros(len(idx), bool)
    ok_d = np.isfinite(d2) & sal_d & (sal_p if method == 2 else True) if use_d else np.zeros(len(idx), bool)
    sq = lambda v: (v * v).astype(F).astype(np.float64)
    if occ == 1:        # residual per TARGET: the owner's; counters: every prefix maximum
        return (float(sq(wp[owner & ok_p]).sum()), float(sq(wd[owner & ok_d]).sum()), int((pm & ok_p).sum()), int((pm & ok_d).sum()))
    n = int(pm.sum())   # occ 2: residual per SOURCE pixel, never retracted; both averages over the accepted pixels
    return (float(sq(wp[pm & ok_p]).sum()), float(sq(wd[pm & ok_d]).sum()), n, n)


def occ2_visible(fr, level, pose, thr_outlier=F(0.3)):
    """numVisiblePixels of calcHessGrad_sphereOcc2: distinct target pixels hit by a gated source pixel (RPI.h:3981-3982)."""
    w = warp(fr, level, pose)
    idx = np.nonzero(w["vis"])[0]
    r, c = w["r"][idx], w["c"][idx]
    with np.errstate(invalid="ignore"):
        keep = ~(np.abs((fr.dep_t[level][r, c] - w["dist"][idx]).astype(F)) > thr_outlier)
    return len(np.unique((r * w["cols"] + c)[keep]))


# ---- pinhole single-sensor error pass (RPI.h:560-748, else-branch) --------------------------------------------------------
def pinhole_error(fr, level, pose, K, method, sigma_p=F(6.0 / 255), sigma_d=F(0.2)):
    """(sum photo, sum depth, n photo, n depth): every valid, visible source pixel, no saliency test."""
    d = fr.dep_s[level]
    rows, cols = d.shape
    s = F(1.0 / 2 ** level)
    fx, fy, ox, oy = (F(K[0]) * s, F(K[1]) * s, F(K[2]) * s, F(K[3]) * s)
    valid = ((fr.dmin < d) & (d < fr.dmax)).ravel()
    cc, rr = np.meshgrid(np.arange(cols).astype(F), np.arange(rows).astype(F))
    z = d.ravel()
    x = ((cc.ravel() - ox) * z * F(1.0 / np.float64(fx))).astype(F)
    y = ((rr.ravel() - oy) * z * F(1.0 / np.float64(fy))).astype(F)
    R = np.asarray(pose, F)[:3, :3]
    t = np.asarray(pose, F)[:3, 3]
    X = ((R[0, 0] * x + R[0, 1] * y) + R[0, 2] * z) + t[0]
    Y = ((R[1, 0] * x + R[1, 1] * y) + R[1, 2] * z) + t[1]
    Z = ((R[2, 0] * x + R[2, 1] * y) + R[2, 2] * z) + t[2]
    with np.errstate(invalid="ignore", divide="ignore"):
        iz = (1.0 / Z.astype(np.float64)).astype(F)
        tc = ((X * fx) * iz + ox).astype(F)
        tr = ((Y * fy) * iz + oy).astype(F)
        rnd = lambda v: np.where(np.isfinite(v), np.sign(v) * np.floor(np.abs(v.astype(np.float64)) + 0.5), -1).astype(np.int64)
        ri, ci = rnd(tr), rnd(tc)
    vis = valid & (ri >= 0) & (ri < rows) & (ci >= 0) & (ci < cols)
    idx = np.nonzero(vis)[0]
    r, c = ri[idx], ci[idx]
    sq = lambda v: (v * v).astype(F).astype(np.float64)
    sp = sd_ = 0.0
    n_p = n_d = 0
    if method in (0, 2):
        diff = (fr.gray_t[level][r, c] - fr.gray_s[level].ravel()[idx]).astype(F)
        we = ((huber(diff, sigma_p) * F(1.0 / np.float64(sigma_p))).astype(F) * diff).astype(F)
        sp, n_p = float(sq(we).sum()), len(idx)
    if method in (1, 2):
        d2 = fr.dep_t[level][r, c]
        fin = np.isfinite(d2)
        depth1 = Z[idx]
        with np.errstate(invalid="ignore", divide="ignore"):
            diff = (d2 - depth1).astype(F)
            sdev = (sigma_d * depth1).astype(F)
            we = ((huber(diff, sdev) / sdev).astype(F) * diff).astype(F)
        sd_, n_d = float(sq(we[fin]).sum()), int(fin.sum())
    return sp, sd_, n_p, n_d


# ---- pinhole occlusion-aware passes (sequential semantics of RPI.h:1107-2030) --------------------------------------------
def _pinhole_warp(fr, level, pose, K):
    d = fr.dep_s[level]
    rows, cols = d.shape
    s = F(1.0 / 2 ** level)
    fx, fy, ox, oy = (F(K[0]) * s, F(K[1]) * s, F(K[2]) * s, F(K[3]) * s)
    valid = ((fr.dmin < d) & (d < fr.dmax)).ravel()
    cc, rr = np.meshgrid(np.arange(cols).astype(F), np.arange(rows).astype(F))
    z = d.ravel()
    x = ((cc.ravel() - ox) * z * F(1.0 / np.float64(fx))).astype(F)
    y = ((rr.ravel() - oy) * z * F(1.0 / np.float64(fy))).astype(F)
    R = np.asarray(pose, F)[:3, :3]
    t = np.asarray(pose, F)[:3, 3]
    X = ((R[0, 0] * x + R[0, 1] * y) + R[0, 2] * z) + t[0]
    Y = ((R[1, 0] * x + R[1, 1] * y) + R[1, 2] * z) + t[1]
    Z = ((R[2, 0] * x + R[2, 1] * y) + R[2, 2] * z) + t[2]
    with np.errstate(invalid="ignore", divide="ignore"):
        iz = (1.0 / Z.astype(np.float64)).astype(F)
        tc = ((X * fx) * iz + ox).astype(F)
        tr = ((Y * fy) * iz + oy).astype(F)
        rnd = lambda v: np.where(np.isfinite(v), np.sign(v) * np.floor(np.abs(v.astype(np.float64)) + 0.5), -1).astype(np.int64)
        ri, ci = rnd(tr), rnd(tc)
    vis = valid & (ri >= 0) & (ri < rows) & (ci >= 0) & (ci < cols)
    return dict(X=X, Y=Y, Z=Z, iz=iz, r=ri, c=ci, vis=vis, rows=rows, cols=cols, fx=fx, fy=fy)


def pinhole_occ(fr, level, pose, K, method, occ, sigma_p=F(6.0 / 255), sigma_d=F(0.2), thr_p=F(0.01), thr_d=F(0.01), thr_outlier=F(1.0)):
    """errorPhotoICP_Occ1/2 and calcHessGrad_Occ1/2 at one pose, grouped by target pixel instead of swept in pixel order (valid while
    every inverse depth is positive: the z-buffer's accepted pixels are then the prefix maxima of 1/Z).
    Returns (sum photo, sum depth, n photo, n depth, H f64, g f64, numVisiblePixels)."""
    w = _pinhole_warp(fr, level, pose, K)
    idx0 = np.nonzero(w["vis"])[0]
    assert (w["iz"][idx0] > 0).all()
    out = []
    for which in ("error", "hess"):
        idx = idx0
        r, c = w["r"][idx], w["c"][idx]
        d2 = fr.dep_t[level][r, c]
        if occ == 2:      # the gates: depth against INVERSE depth in the error pass (RPI.h:1687-1690, sic), against depth in H, g (RPI.h:1857-1862)
            other = w["iz"][idx] if which == "error" else w["Z"][idx]
            with np.errstate(invalid="ignore"):
                keep = ~(np.abs((d2 - other).astype(F)) > thr_outlier)
            idx, r, c, d2 = idx[keep], r[keep], c[keep], d2[keep]
        X, Y, Z, iz = w["X"][idx], w["Y"][idx], w["Z"][idx], w["iz"][idx]
        target = r * w["cols"] + c
        pm, owner = _prefix_maxima(target, iz, idx) if len(idx) else (np.zeros(0, bool), np.zeros(0, bool))
        gx, gy = fr.gx[level][r, c], fr.gy[level][r, c]
        sal_p = ~((np.abs(gx) < thr_p) & (np.abs(gy) < thr_p))
        dgx, dgy = fr.dgx[level][r, c], fr.dgy[level][r, c]
        sal_d = ~((np.abs(dgx) < thr_d) & (np.abs(dgy) < thr_d))
        diff_p = (fr.gray_t[level][r, c] - fr.gray_s[level].ravel()[idx]).astype(F)
        wgt_p = (huber(diff_p, sigma_p) * F(1.0 / np.float64(sigma_p))).astype(F)
        res_p = (wgt_p * diff_p).astype(F)
        with np.errstate(invalid="ignore", divide="ignore"):
            diff_d = (d2 - Z).astype(F)
            sd = (sigma_d * Z).astype(F)
            wgt_d = (huber(diff_d, sd) / sd).astype(F)
            res_d = (wgt_d * diff_d).astype(F)
        fin = np.isfinite(d2)
        sq = lambda v: (v * v).astype(F).astype(np.float64)
        if which == "error":
            ok_p = sal_p if method in (0, 2) else np.zeros(len(idx), bool)
            ok_d = (fin & sal_d & (sal_p if method == 2 else True)) if method in (1, 2) else np.zeros(len(idx), bool)
            out += [float(sq(res_p[owner & ok_p]).sum()), float(sq(res_d[owner & ok_d]).sum()), int((pm & ok_p).sum()), int((pm & ok_d).sum())]
        else:
            n_vis = int(pm.sum()) + len(np.unique(target))      # a target's first arrival is counted twice (RPI.h:1421-1430)
            H = np.zeros((6, 6)); g = np.zeros(6)
            if method in (0, 2):
                fx, fy = w["fx"], w["fy"]
                iz2 = iz * iz
                Jw0 = np.stack([fx * iz, 0 * iz, -fx * X * iz2, -fx * Y * X * iz2, fx * (1 + X * X * iz2), -fx * Y * iz], 1).astype(F)
                Jw1 = np.stack([0 * iz, fy * iz, -fy * Y * iz2, -fy * (1 + Y * Y * iz2), fy * X * Y * iz2, fy * X * iz], 1).astype(F)
                rows_p = pm & sal_p & (res_p != 0)               # rows are kept per SOURCE pixel; both sums test the photo residual
                Jp = ((wgt_p * gx)[:, None] * Jw0 + (wgt_p * gy)[:, None] * Jw1).astype(F)
                H += np.einsum("ni,nj->ij", Jp[rows_p], Jp[rows_p], dtype=np.float64)
                g += Jp[rows_p].astype(np.float64).T @ res_p[rows_p].astype(np.float64)
                if method == 2:
                    rows_d = rows_p & sal_d & fin
                    Jz = np.stack([0 * X, 0 * X, 1 + 0 * X, Y, -X, 0 * X], 1).astype(F)
                    with np.errstate(invalid="ignore"):
                        Jd = (wgt_d[:, None] * ((dgx[:, None] * Jw0 + dgy[:, None] * Jw1) - Jz)).astype(F)
                    H += np.einsum("ni,nj->ij", Jd[rows_d], Jd[rows_d], dtype=np.float64)
                    g += Jd[rows_d].astype(np.float64).T @ res_d[rows_d].astype(np.float64)
            out += [H, g, n_vis]
    return tuple(out)


def pinhole_salient_list(fr, level, thr=F(0.01)):
    """vSalientPixels (RPI.h:420-424): interior pixels of the TARGET whose gray gradient exceeds thr in x or y, index order."""
    gx, gy = fr.gx[level], fr.gy[level]
    m = (np.abs(gx) > thr) | (np.abs(gy) > thr)
    m[0, :] = m[-1, :] = False
    m[:, 0] = m[:, -1] = False
    return np.nonzero(m.ravel())[0]
