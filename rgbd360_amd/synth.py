"""Deterministic synthetic spherical RGB-D frames.

An axis-aligned box room (x up in [-1.5, 1.5], y in [-3, 3], z in [-4, 4]) is rendered into
full-sphere equirectangular panoramas with the reference's pixel<->ray convention
(RegisterPhotoICP.h:4567-4582): row r <-> phi = (H/2 - 0.5 - r) * 2pi/W, column c <-> theta = c * 2pi/W,
ray = (sin phi, -cos phi sin theta, -cos phi cos theta).  Output matches what Frame360 hands to
RegisterPhotoICP (Frame360.h:104-111, 394): RGB uint8 HxWx3 and range uint16 millimetres (0 = invalid).

This is input generation only (no alignment arithmetic); both the HIP path and the CPU oracle consume
the same arrays.
"""
from __future__ import annotations

import math

import numpy as np

ROOM_LO = np.array([-1.5, -3.0, -4.0])
ROOM_HI = np.array([1.5, 3.0, 4.0])
CAM_A = np.array([0.1, -0.2, 0.3])
MIN_DEPTH, MAX_DEPTH = 0.3, 6.0

_WAVELENGTHS = np.array([0.15, 0.22, 0.33, 0.6, 1.1, 2.3])  # metres
_AMPS = np.array([0.10, 0.10, 0.09, 0.08, 0.07, 0.06])


def rodrigues(axis: np.ndarray, angle: float) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def make_pose(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _wall_params(seed: int):
    rng = np.random.default_rng(seed)
    K = len(_WAVELENGTHS)
    alpha = rng.uniform(0, math.pi, size=(6, K))
    phase = rng.uniform(0, 2 * math.pi, size=(6, K))
    jitter = rng.uniform(0.9, 1.1, size=(6, K))
    freq = 1.0 / (_WAVELENGTHS[None, :] * jitter)
    lattice = rng.uniform(-1.0, 1.0, size=(6, 48, 48))
    chroma = rng.uniform(-1.0, 1.0, size=(6, 2, K))
    return alpha, phase, freq, lattice, chroma


def render(T_wc: np.ndarray, width: int, height: int | None = None, seed: int = 1234, depth_f32: bool = False,
           strip: int = 32):
    """Render the room from camera-to-world pose T_wc.  Returns (rgb uint8 HxWx3, depth uint16 mm HxW),
    or float32 metres for the depth when depth_f32 (RegisterPhotoICP.h:318-319 accepts both).
    Rows are processed in strips so temporaries stay cache-sized; the result does not depend on `strip`."""
    W = int(width)
    H = int(height) if height is not None else W // 2
    params = _wall_params(seed)
    rgb = np.empty((H, W, 3), dtype=np.uint8)
    depth = np.empty((H, W), dtype=np.float32 if depth_f32 else np.uint16)
    for r0 in range(0, H, strip):
        r1 = min(H, r0 + strip)
        _render_rows(T_wc, W, H, r0, r1, params, depth_f32, rgb[r0:r1], depth[r0:r1])
    return rgb, depth


def _render_rows(T_wc, W, H, r0, r1, params, depth_f32, rgb_out, depth_out, pinhole=None):
    alpha, phase, freq, lattice, chroma = params
    n = r1 - r0
    if pinhole is not None:
        _shade(T_wc, _pinhole_rays(W, r0, r1, pinhole), n, W, params, depth_f32, rgb_out, depth_out)
        return
    res = 2 * math.pi / W
    phi = (H / 2 - 0.5 - np.arange(r0, r1, dtype=np.float64)) * res
    theta = np.arange(W, dtype=np.float64) * res
    sp, cp = np.sin(phi)[:, None], np.cos(phi)[:, None]
    st, ct = np.sin(theta)[None, :], np.cos(theta)[None, :]
    ray_c = np.stack([np.broadcast_to(sp, (n, W)), -cp * st, -cp * ct], axis=-1)  # n x W x 3
    _shade(T_wc, ray_c, n, W, params, depth_f32, rgb_out, depth_out)


def _pinhole_rays(W, r0, r1, K):
    """Camera-frame rays of a pinhole sensor (x right, y down, z forward), NOT normalised: z = 1, so the ray parameter of
    a hit is the z-depth the sensor stores (Frame360 sensor images, RegisterPhotoICP.h:4277-4300)."""
    fx, fy, ox, oy = K
    cc = (np.arange(W, dtype=np.float64) - ox) / fx
    rr = (np.arange(r0, r1, dtype=np.float64) - oy) / fy
    n = r1 - r0
    return np.stack([np.broadcast_to(cc[None, :], (n, W)), np.broadcast_to(rr[:, None], (n, W)), np.ones((n, W))], axis=-1)


def _shade(T_wc, ray_c, n, W, params, depth_f32, rgb_out, depth_out):
    alpha, phase, freq, lattice, chroma = params
    R, o = T_wc[:3, :3], T_wc[:3, 3]
    ray_w = ray_c @ R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        bound = np.where(ray_w > 0, ROOM_HI, ROOM_LO)
        tk = (bound - o) / ray_w
        tk = np.where(np.abs(ray_w) < 1e-12, np.inf, tk)
    axis = np.argmin(tk, axis=-1)
    t = np.take_along_axis(tk, axis[..., None], axis=-1)[..., 0]
    P = o + t[..., None] * ray_w
    side = (np.take_along_axis(ray_w, axis[..., None], axis=-1)[..., 0] > 0).astype(np.int64)
    wall = axis * 2 + side
    # wall-plane coordinates: the two axes other than `axis`
    ua = np.where(axis == 0, 1, 0)
    va = np.where(axis == 2, 1, 2)
    u = np.take_along_axis(P, ua[..., None], axis=-1)[..., 0]
    v = np.take_along_axis(P, va[..., None], axis=-1)[..., 0]

    tex = np.full((n, W), 0.5)
    c1 = np.zeros((n, W))
    c2 = np.zeros((n, W))
    for k in range(len(_WAVELENGTHS)):
        a = alpha[wall, k]
        s = np.sin(2 * math.pi * freq[wall, k] * (u * np.cos(a) + v * np.sin(a)) + phase[wall, k])
        tex += _AMPS[k] * s
        c1 += 0.02 * chroma[wall, 0, k] * s
        c2 += 0.02 * chroma[wall, 1, k] * s
    # seeded lattice (value) noise, 0.3 m cells, smoothstep-interpolated
    gu, gv = (u + 6.0) / 0.3, (v + 6.0) / 0.3
    iu, iv = np.floor(gu).astype(np.int64), np.floor(gv).astype(np.int64)
    fu, fv = gu - iu, gv - iv
    fu, fv = fu * fu * (3 - 2 * fu), fv * fv * (3 - 2 * fv)
    iu0, iv0 = np.clip(iu, 0, 46), np.clip(iv, 0, 46)
    n00 = lattice[wall, iu0, iv0]
    n10 = lattice[wall, iu0 + 1, iv0]
    n01 = lattice[wall, iu0, iv0 + 1]
    n11 = lattice[wall, iu0 + 1, iv0 + 1]
    tex += 0.08 * ((n00 * (1 - fu) + n10 * fu) * (1 - fv) + (n01 * (1 - fu) + n11 * fu) * fv)

    col = np.stack([tex + c1, tex, tex + c2], axis=-1)
    rgb_out[...] = np.clip(np.rint(col * 255.0), 0, 255).astype(np.uint8)
    valid = (t > MIN_DEPTH) & (t < MAX_DEPTH)
    if depth_f32:
        depth_out[...] = np.where(valid, t, 0.0).astype(np.float32)
    else:
        depth_out[...] = np.where(valid, np.rint(t * 1000.0), 0).astype(np.uint16)


def default_motion(seed: int = 1234, trans: float = 0.06, rot_deg: float = 2.0) -> np.ndarray:
    """Camera-B-in-camera-A motion: `trans` metres and `rot_deg` degrees about seeded random axes."""
    rng = np.random.default_rng(seed + 7919)
    ax = rng.normal(size=3)
    td = rng.normal(size=3)
    td = td / np.linalg.norm(td) * trans
    return make_pose(rodrigues(ax, math.radians(rot_deg)), td)


def make_pair(width: int, height: int | None = None, seed: int = 1234, trans: float = 0.06, rot_deg: float = 2.0,
              depth_f32: bool = False):
    """Target frame A, source frame B and the ground-truth relPose (source points -> target frame,
    p_trg = R p_src + t; RegisterPhotoICP.h:2663) as a 4x4 float64."""
    T_wA = make_pose(np.eye(3), CAM_A)
    M = default_motion(seed, trans, rot_deg)
    T_wB = T_wA @ M
    rgbA, dA = render(T_wA, width, height, seed, depth_f32)
    rgbB, dB = render(T_wB, width, height, seed, depth_f32)
    T_rel = np.linalg.inv(T_wA) @ T_wB
    return (rgbA, dA), (rgbB, dB), T_rel


def spoil_depth(depth_m: np.ndarray, seed: int = 0, ramps: bool = False, isolated: bool = True) -> np.ndarray:
    """A float32-metres depth image with what a caller's CV_32FC1 image may carry and the reference takes AS IS at level 0
    (RegisterPhotoICP.h:318-319: `pyramid[0] = img`): patches of NaN, +Inf, negative values, values beyond maxDepth and zeros, plus
    isolated pixels of each kind.  Patches are constant, so the monotone gradient inside them is zero; with `ramps` two patches hold
    strictly increasing values instead (negative, and from -1 to +1 through 0): their depth gradient is salient, and the reference's
    depth residual there is NaN (RPI.h:2721-2723: the Huber weight of a non-positive standard deviation) -- the sums turn NaN.
    Test input generation only."""
    d = np.array(depth_m, dtype=np.float32, copy=True)
    H, W = d.shape
    rng = np.random.default_rng(seed + 977)
    ph, pw = max(2, H // 16), max(2, W // 24)
    kinds = [np.float32(np.nan), np.float32(np.inf), np.float32(-1.5), np.float32(7.5), np.float32(0.0), np.float32(-np.inf), np.float32(25.0)]
    for k, v in enumerate(kinds):
        for _ in range(2):
            r0, c0 = int(rng.integers(1, H - ph - 1)), int(rng.integers(1, W - pw - 1))
            d[r0:r0 + ph, c0:c0 + pw] = v
        rr, cc = rng.integers(0, H, size=max(4, H * W // 2048)), rng.integers(0, W, size=max(4, H * W // 2048))
        if isolated:        # (at 2048 x 1024 some of the 7 x 1024 isolated pixels fall next to each other: -1.5 between -Inf and a valid depth is a ramp)
            d[rr, cc] = v
    if ramps:
        for lo, hi in ((-3.0, -0.5), (-1.0, 1.0)):
            r0, c0 = int(rng.integers(1, H - ph - 1)), int(rng.integers(1, W - pw - 1))
            d[r0:r0 + ph, c0:c0 + pw] = np.linspace(lo, hi, pw, dtype=np.float32)[None, :] + np.linspace(0, 0.2, ph, dtype=np.float32)[:, None]
    return d


def trajectory_pose(k: int, seed: int = 1234) -> np.ndarray:
    """Frame k of a smooth closed trajectory (config 4: pair i = frames i, i+1): a 0.6 m-radius loop in the
    y-z plane with ~6 cm steps, 2 degrees of yaw about the up axis per frame and a small seeded wobble."""
    rng = np.random.default_rng(seed + 104729)
    ph = rng.uniform(0, 2 * math.pi, size=3)
    n_loop = 63.0
    a = 2 * math.pi * k / n_loop
    pos = np.array([0.15 * math.sin(0.5 * a + ph[0]), 0.6 * math.cos(a), 0.6 * math.sin(a)])
    Rm = rodrigues(np.array([1.0, 0, 0]), math.radians(2.0) * k) @ rodrigues(
        np.array([0, math.cos(ph[1]), math.sin(ph[1])]), math.radians(1.5) * math.sin(0.7 * a + ph[2]))
    return make_pose(Rm, pos)


def make_sequence_pair(i: int, width: int, height: int | None = None, seed: int = 1234):
    """Pair i of the odometry-like sequence: target = frame i, source = frame i+1."""
    T_wA, T_wB = trajectory_pose(i, seed), trajectory_pose(i + 1, seed)
    rgbA, dA = render(T_wA, width, height, seed)
    rgbB, dB = render(T_wB, width, height, seed)
    return (rgbA, dA), (rgbB, dB), np.linalg.inv(T_wA) @ T_wB


def pose_error(Ta: np.ndarray, Tb: np.ndarray):
    """(rotation angle of Ra Rb^T in rad, ||ta - tb|| in m) -- SURVEY.md §8d metric (iii)."""
    Ta = np.asarray(Ta, dtype=np.float64).reshape(4, 4)
    Tb = np.asarray(Tb, dtype=np.float64).reshape(4, 4)
    Rd = Ta[:3, :3] @ Tb[:3, :3].T
    c = max(-1.0, min(1.0, (np.trace(Rd) - 1) / 2))
    # asin form is accurate for tiny angles
    s = 0.5 * math.sqrt((Rd[2, 1] - Rd[1, 2]) ** 2 + (Rd[0, 2] - Rd[2, 0]) ** 2 + (Rd[1, 0] - Rd[0, 1]) ** 2)
    ang = math.atan2(s, c)
    return ang, float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


def add_occluder(pair):
    """The box room is convex, so no surface hides another.  A near 'billboard' pasted into both frames (2 columns
    apart) makes source pixels of different depth land on the same target pixel once a pose moves them: real z-buffer
    conflicts for the occlusion-aware passes (RegisterPhotoICP.h:3232-4249)."""
    (rgbA, dA), (rgbB, dB), T = pair
    rgbA, dA, rgbB, dB = rgbA.copy(), dA.copy(), rgbB.copy(), dB.copy()
    H, W = dA.shape
    for (rgb, d, c0) in ((rgbA, dA, W // 3), (rgbB, dB, W // 3 + 2)):
        r0, r1, c1 = H // 3, 2 * H // 3, c0 + W // 8
        d[r0:r1, c0:c1] = 1200 if d.dtype == np.uint16 else 1.2
        rgb[r0:r1, c0:c1] = (np.indices((r1 - r0, c1 - c0)).sum(axis=0)[..., None] * 9 % 256).astype(np.uint8)
    return (rgbA, dA), (rgbB, dB), T


def occlusion_test_poses(T_gt):
    rng = np.random.default_rng(11)
    out = [np.eye(4), np.asarray(T_gt)]
    out.append(make_pose(rodrigues(rng.normal(size=3), 0.03), np.array([0.0, 0.25, 0.1])))    # large sideways step
    out.append(make_pose(np.eye(3), np.array([0.0, 0.0, -0.4])))                                # moving away: compression
    return out


# ---- pinhole single-sensor frames (RegisterPhotoICP::alignFrames) ----------------------------------
def pinhole_intrinsics(width: int, height: int):
    """The rig's sensor model as RegisterRGBD360.h:357-365 builds it: f = 525 * width / 640, principal point at the centre."""
    f = 525.0 * width / 640.0
    return (f, f, width / 2 - 0.5, height / 2 - 0.5)


def render_pinhole(T_wc: np.ndarray, width: int, height: int, seed: int = 1234, depth_f32: bool = False, K=None, strip: int = 32):
    """The room through a pinhole sensor at camera-to-world pose T_wc: (rgb uint8 HxWx3, z-depth uint16 mm | float32 m)."""
    K = pinhole_intrinsics(width, height) if K is None else K
    params = _wall_params(seed)
    rgb = np.empty((height, width, 3), dtype=np.uint8)
    depth = np.empty((height, width), dtype=np.float32 if depth_f32 else np.uint16)
    for r0 in range(0, height, strip):
        r1 = min(height, r0 + strip)
        _render_rows(T_wc, width, height, r0, r1, params, depth_f32, rgb[r0:r1], depth[r0:r1], pinhole=K)
    return rgb, depth


def make_pinhole_pair(width: int = 320, height: int = 240, seed: int = 1234, trans: float = 0.03, rot_deg: float = 1.0,
                      depth_f32: bool = False):
    """Target sensor frame A, source frame B, ground-truth relPose (p_trg = R p_src + t) and the intrinsics."""
    # the sensor looks towards a vertical corner of the room (two walls + floor/ceiling bands in view: well conditioned);
    # image "down" = room -x (x is up)
    R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])      # columns: camera x, y, z axes in the world
    T_wA = make_pose(R0 @ rodrigues(np.array([0.0, 1.0, 0.0]), 0.75), CAM_A + np.array([0.0, -1.0, -1.5]))
    M = default_motion(seed, trans, rot_deg)
    T_wB = T_wA @ M
    K = pinhole_intrinsics(width, height)
    rgbA, dA = render_pinhole(T_wA, width, height, seed, depth_f32, K)
    rgbB, dB = render_pinhole(T_wB, width, height, seed, depth_f32, K)
    return (rgbA, dA), (rgbB, dB), np.linalg.inv(T_wA) @ T_wB, K


def rig_extrinsics(n_sensors: int = 8):
    """Sensor -> rig poses of an n-sensor ring: sensor s looks outwards, turned by 360 / n degrees about the rig's up axis (x);
    image "down" = rig -x (the arrangement of the reference's 8-Asus rig, Calib360.h)."""
    R0 = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])
    return [make_pose(rodrigues(np.array([1.0, 0, 0]), math.radians(360.0 / n_sensors * s)) @ R0, np.zeros(3)) for s in range(n_sensors)]


def make_rig_pair(width: int = 320, height: int = 240, seed: int = 1234, trans: float = 0.05, rot_deg: float = 2.0, n_sensors: int = 8,
                  depth_f32: bool = False):
    """Two frames of an n-sensor pinhole rig in the room: (frame1, frame2, M, Rt, K); frame_k = list of (rgb, depth) per sensor,
    M = the rig's motion with p_rig1 = M p_rig2 (the pose RegisterDensePhotoICP estimates), Rt = sensor -> rig poses."""
    Rt = rig_extrinsics(n_sensors)
    T_w1 = make_pose(np.eye(3), np.asarray(CAM_A, float))
    M = default_motion(seed, trans, rot_deg)
    T_w2 = T_w1 @ M
    K = pinhole_intrinsics(width, height)
    f1 = [render_pinhole(T_w1 @ Rt[s], width, height, seed, depth_f32, K) for s in range(n_sensors)]
    f2 = [render_pinhole(T_w2 @ Rt[s], width, height, seed, depth_f32, K) for s in range(n_sensors)]
    return f1, f2, M, Rt, K
