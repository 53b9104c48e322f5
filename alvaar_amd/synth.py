"""Seeded synthetic inputs shared by tests and bench.py.

Identical bytes go to the HIP path, the oracle restatement and the compiled
reference.  Pure numpy; nothing here touches oracle/ or the GPU.
"""
from __future__ import annotations

import numpy as np


def texture_canvas(width: int, height: int, seed: int = 7, margin: int = 400) -> np.ndarray:
    """Random filled squares on black: (H+margin) x (W+margin) u8.

    W*H/100 squares, side 4 + r%12, gray r%256, uniform positions
    (recipe of SURVEY.md §8(d); our own RNG stream).
    """
    rng = np.random.RandomState(seed)
    ch, cw = height + margin, width + margin
    canvas = np.zeros((ch, cw), np.uint8)
    n = (width * height) // 100
    xs = rng.randint(0, cw, n)
    ys = rng.randint(0, ch, n)
    sides = 4 + rng.randint(0, 12, n)
    grays = rng.randint(0, 256, n)
    for x, y, s, g in zip(xs, ys, sides, grays):
        canvas[y:y + s, x:x + s] = g
    return canvas


def frame_gray(canvas: np.ndarray, k: int, width: int, height: int, noise_seed: int | None = None) -> np.ndarray:
    """Frame k = crop of the canvas at offset (2k, k); optional +-5 gray noise."""
    x0, y0 = 2 * k, k
    g = canvas[y0:y0 + height, x0:x0 + width].copy()
    if noise_seed is not None:
        rng = np.random.RandomState(noise_seed + k)
        g = np.clip(g.astype(np.int16) + rng.randint(-5, 6, g.shape), 0, 255).astype(np.uint8)
    return g


def gray_to_rgba(gray: np.ndarray, seed: int | None = None) -> np.ndarray:
    """RGBA = (g, g, g, 255); with a seed, R/G/B get independent +-3 jitter so the
    colour weights of the gray conversion are actually exercised."""
    h, w = gray.shape
    rgba = np.empty((h, w, 4), np.uint8)
    if seed is None:
        rgba[..., 0] = gray
        rgba[..., 1] = gray
        rgba[..., 2] = gray
    else:
        rng = np.random.RandomState(seed)
        for c in range(3):
            rgba[..., c] = np.clip(gray.astype(np.int16) + rng.randint(-3, 4, gray.shape), 0, 255)
    rgba[..., 3] = 255
    return rgba


def random_rgba(width: int, height: int, seed: int) -> np.ndarray:
    return np.random.RandomState(seed).randint(0, 256, (height, width, 4)).astype(np.uint8)


def stream_rgba(width: int, height: int, n_frames: int, seed: int = 7, noise: bool = False) -> np.ndarray:
    canvas = texture_canvas(width, height, seed)
    out = np.empty((n_frames, height, width, 4), np.uint8)
    for k in range(n_frames):
        out[k] = gray_to_rgba(frame_gray(canvas, k, width, height, 11 if noise else None))
    return out


# ----------------------------------------------------------------------------------------------
# SE3 helpers (Sophus conventions: tangent = (upsilon, omega), T <- Exp(d) * T)
def so3_exp(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)
    if th < 1e-10:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def se3_exp(xi: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    v, w = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)
    R = so3_exp(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return R, V @ v


def rot_to_quat_xyzw(R: np.ndarray) -> np.ndarray:
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_xyzw_to_rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose7(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """[tx,ty,tz,qx,qy,qz,qw] -- memory order of the reference's PoseParametersBlock
    (ceres_parametrization.hpp:64-71; Eigen coeffs() order x,y,z,w)."""
    return np.concatenate([np.asarray(t, float), rot_to_quat_xyzw(R)])


def make_pnp_problem(n: int, seed: int = 3, fx: float = 579.4, fy: float = 579.4, cx: float = 320.0,
                     cy: float = 240.0, width: int = 640, height: int = 480, noise_px: float = 0.5,
                     outlier_frac: float = 0.1, pose_noise: float = 0.01):
    """World points = back-projected pixels at depths U(3,9) under a ground-truth Twc;
    returns dict(uv, bv, wpt, pose_gt, pose_init)."""
    rng = np.random.RandomState(seed)
    R, t = se3_exp(np.array([0.3, -0.1, 0.2, 0.05, -0.03, 0.02]))
    uv = np.stack([rng.uniform(20, width - 20, n), rng.uniform(20, height - 20, n)], 1)
    z = rng.uniform(3, 9, n)
    pc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    wpt = pc @ R.T + t  # X_w = R_wc X_c + t_wc
    uvn = uv + rng.normal(0, noise_px, uv.shape)
    nout = int(outlier_frac * n)
    if nout:
        idx = rng.choice(n, nout, replace=False)
        uvn[idx] += rng.uniform(-60, 60, (nout, 2))
    bv = np.stack([(uvn[:, 0] - cx) / fx, (uvn[:, 1] - cy) / fy, np.ones(n)], 1)
    bv /= np.linalg.norm(bv, axis=1, keepdims=True)
    Rn, tn = se3_exp(rng.normal(0, pose_noise, 6))
    Ri, ti = Rn @ R, Rn @ t + tn
    return dict(uv=uvn, bv=bv, wpt=wpt, pose_gt=pose7(R, t), pose_init=pose7(Ri, ti),
                K=(fx, fy, cx, cy))


def make_ba_problem(n_kf: int = 20, n_pt: int = 3000, seed: int = 42, fx: float = 579.4, fy: float = 579.4,
                    cx: float = 320.0, cy: float = 240.0, width: int = 640, height: int = 480,
                    px_noise: float = 0.5, pose_noise: float = 0.005, invdepth_noise: float = 0.02,
                    n_fixed: int = 2):
    """Local-BA instance of SURVEY.md §8(d): poses exp([0.08k,0.01k,0,0.002k,0.01k,0]), points
    U([-3,5]x[-2.5,2.5]x[3,9]), pixel noise N(0,0.5^2), pose perturbation N(0,0.005^2) on KFs >= n_fixed,
    inverse-depth noise 2 %, first n_fixed KFs constant.  Anchored inverse depth: the first observer
    (lowest kf index) is the anchor and contributes no residual (optimizer.cpp:186-201)."""
    rng = np.random.RandomState(seed)
    K = np.array([fx, fy, cx, cy])
    Rs, ts = [], []
    for k in range(n_kf):
        R, t = se3_exp(np.array([0.08 * k, 0.01 * k, 0.0, 0.002 * k, 0.01 * k, 0.0]))
        Rs.append(R)
        ts.append(t)
    pts = np.stack([rng.uniform(-3, 5, n_pt), rng.uniform(-2.5, 2.5, n_pt), rng.uniform(3, 9, n_pt)], 1)
    poses_gt = np.stack([pose7(R, t) for R, t in zip(Rs, ts)])
    poses = poses_gt.copy()
    for k in range(n_fixed, n_kf):
        Rn, tn = se3_exp(rng.normal(0, pose_noise, 6))
        poses[k] = pose7(Rn @ Rs[k], Rn @ ts[k] + tn)
    anchor_kf = np.full(n_pt, -1, np.int32)
    anchor_uv = np.zeros((n_pt, 2))
    inv_depth = np.zeros(n_pt)
    obs_kf, obs_pt, obs_uv = [], [], []
    for p in range(n_pt):
        for k in range(n_kf):
            pc = Rs[k].T @ (pts[p] - ts[k])
            if pc[2] <= 0.5:
                continue
            u = fx * pc[0] / pc[2] + cx
            v = fy * pc[1] / pc[2] + cy
            if not (0 <= u < width and 0 <= v < height):
                continue
            un = u + rng.normal(0, px_noise)
            vn = v + rng.normal(0, px_noise)
            if anchor_kf[p] < 0:
                anchor_kf[p] = k
                anchor_uv[p] = (un, vn)
                inv_depth[p] = (1.0 / pc[2]) * (1.0 + rng.normal(0, invdepth_noise))
            else:
                obs_kf.append(k)
                obs_pt.append(p)
                obs_uv.append((un, vn))
    # keep only points with an anchor and >=1 residual; re-index densely
    obs_kf = np.asarray(obs_kf, np.int32)
    obs_pt = np.asarray(obs_pt, np.int32)
    obs_uv = np.asarray(obs_uv, float).reshape(-1, 2)
    used = np.zeros(n_pt, bool)
    used[obs_pt] = True
    remap = -np.ones(n_pt, np.int32)
    remap[used] = np.arange(int(used.sum()), dtype=np.int32)
    kf_const = np.zeros(n_kf, np.uint8)
    kf_const[:n_fixed] = 1
    akf, auv, rho = anchor_kf[used].copy(), anchor_uv[used].copy(), inv_depth[used].copy()
    # XYZ parameterisation of the same (noisy) initial points: X = T_w,anchor * (K^-1 [u,v,1] / rho)
    pts_xyz = np.empty((len(rho), 3))
    for i in range(len(rho)):
        pa = np.array([(auv[i, 0] - cx) / fx, (auv[i, 1] - cy) / fy, 1.0]) / rho[i]
        Ra = quat_xyzw_to_rot(poses[akf[i], 3:])
        pts_xyz[i] = Ra @ pa + poses[akf[i], :3]
    return dict(poses=poses, poses_gt=poses_gt, kf_const=kf_const, calib=K,
                anchor_kf=akf, anchor_uv=auv, inv_depth=rho, pts_xyz=pts_xyz,
                pts_gt=pts[used].copy(), obs_kf=obs_kf, obs_pt=remap[obs_pt].copy(), obs_uv=obs_uv)


def make_triangulation_problem(n: int, n_groups: int, seed: int, px_noise: float = 0.4, bad_frac: float = 0.15):
    """A new keyframe + n_groups earlier keyframes observing n points: Twc poses (pose7), unit bearings, pixel observations,
    group index per point.  A fraction of the points gets a gross pixel error or sits behind / very close to a camera, so that
    all three status values occur."""
    rng = np.random.RandomState(seed)
    fx = fy = 520.0
    cx, cy = 320.0, 240.0

    def pose(center, yaw):
        R = so3_exp(np.array([0.03 * rng.randn(), yaw, 0.02 * rng.randn()]))
        q = _quat_from_R(R)
        return np.concatenate([center, q]), R
    pose_new, Rn = pose(np.array([0.6, 0.05, 0.1]), 0.05)
    pose_kf, Rk = [], []
    for g in range(n_groups):
        p, R = pose(np.array([-0.3 - 0.25 * g, 0.02 * rng.randn(), 0.05 * rng.randn()]), -0.04 * g)
        pose_kf.append(p)
        Rk.append(R)
    pose_kf = np.array(pose_kf)
    group = rng.randint(0, n_groups, n).astype(np.int32)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2.5, 12, n)], 1)
    bad = rng.rand(n) < bad_frac
    bvl, bvr, ul, ur = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
    for i in range(n):
        g = group[i]
        pl = Rk[g].T @ (X[i] - pose_kf[g][:3])
        pr = Rn.T @ (X[i] - pose_new[:3])
        for p, bv, u in ((pl, bvl, ul), (pr, bvr, ur)):
            z = p[2] if abs(p[2]) > 1e-3 else 1e-3
            px = np.array([fx * p[0] / z + cx, fy * p[1] / z + cy]) + px_noise * rng.randn(2)
            if bad[i] and rng.rand() < 0.5:
                px += rng.uniform(-25, 25, 2)
            u[i] = px.astype(np.float32)
            b = np.array([(u[i, 0] - cx) / fx, (u[i, 1] - cy) / fy, 1.0])
            if bad[i] and rng.rand() < 0.15:
                b[2] = -1.0     # a bearing that points backwards: triangulates behind the camera
            bv[i] = b / np.linalg.norm(b)
    return dict(pose_kf=pose_kf, pose_new=pose_new, group=group, bvl=bvl, bvr=bvr, unpxl=ul, unpxr=ur, K=(fx, fy, cx, cy))


def _quat_from_R(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()
    return q if q[3] >= 0 else -q


def make_match_to_map_problem(n_points: int, seed: int, n_kf: int = 8, dist=(0.0, 0.0, 0.0, 0.0), dup_frac: float = 0.45, px_noise: float = 0.3,
                              max_flips: int = 10, twin_frac: float = 0.0):
    """A small consistent map for Mapper::matchToMap: n_kf keyframes (the last one is the frame being matched), map points with
    per-keyframe pixel observations and descriptors.  A fraction of the world points exists TWICE in the map -- an old map
    point seen in the first keyframes (local map) and a recent one tracked into the frame -- which is exactly what
    matchToMap is there to find and merge; the rest are only-old or only-tracked points."""
    rng = np.random.RandomState(seed)
    fx = fy = 520.0
    cx, cy, W, H = 320.0, 240.0, 640, 480
    calib = np.array([fx, fy, cx, cy, *dist, W, H], np.float64)
    poses, Rs, cs = [], [], []
    for k in range(n_kf):
        R = so3_exp(np.array([0.01 * rng.randn(), 0.02 * k + 0.01 * rng.randn(), 0.01 * rng.randn()]))
        c = np.array([0.12 * k, 0.01 * rng.randn(), 0.02 * rng.randn()])
        poses.append(np.concatenate([c, _quat_from_R(R)]))
        Rs.append(R)
        cs.append(c)
    poses = np.array(poses)

    def project(k, X):
        p = Rs[k].T @ (X - cs[k])
        x, y = p[0] / p[2], p[1] / p[2]
        r2 = x * x + y * y
        cd = 1 + dist[0] * r2 + dist[1] * r2 * r2
        xd = x * cd + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
        yd = y * cd + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
        return np.array([fx * xd + cx, fy * yd + cy]), p[2]
    last = n_kf - 1
    mp_id, mp_wpt, mp_is3d, obs_ptr, obs_kf, obs_px, obs_desc = [], [], [], [0], [], [], []
    kinds = []

    def add_mp(idv, X, kfs, base, is3d=True):
        rows = []
        for k in kfs:
            px, z = project(k, X)
            if z < 0.3 or not (10 <= px[0] < W - 10 and 10 <= px[1] < H - 10):   # keypoints live inside the image
                continue
            d = base.copy()
            for b in rng.randint(0, 256, rng.randint(2, max_flips)):
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            rows.append((k, np.clip(px + px_noise * rng.randn(2), 1.0, [W - 2.0, H - 2.0]).astype(np.float32), d))
        if not rows:
            return False
        mp_id.append(idv)
        mp_wpt.append(X)
        mp_is3d.append(1 if is3d else 0)
        for k, px, d in rows:
            obs_kf.append(k)
            obs_px.append(px)
            obs_desc.append(d)
        obs_ptr.append(len(obs_kf))
        return True
    half = n_kf // 2
    for j in range(n_points):
        X = np.array([rng.uniform(-2.2, 3.0), rng.uniform(-1.8, 1.8), rng.uniform(3.0, 9.0)])
        base = rng.randint(0, 256, 32).astype(np.uint8)
        if j > 0 and rng.rand() < twin_frac:   # a look-alike right next to the previous point: exercises the 0.9 ratio test
            X = prevX + np.array([0.004, 0.003, 0.0]) * rng.randn(3)
            base = prevBase.copy()
            for b in rng.randint(0, 256, 6):
                base[b >> 3] ^= np.uint8(1 << (b & 7))
        prevX, prevBase = X, base
        u = rng.rand()
        if u < dup_frac:          # the same physical point twice
            add_mp(5000 + j, X + 0.002 * rng.randn(3), range(0, half), base)
            add_mp(1000 + j, X, range(half + 1, n_kf), base, is3d=bool(rng.rand() < 0.7))
            kinds.append("dup")
        elif u < dup_frac + 0.3:  # only old (local map, nothing to match)
            add_mp(5000 + j, X, range(0, half + 1), base)
            kinds.append("old")
        else:                     # only tracked
            add_mp(1000 + j, X, range(rng.randint(half, last), n_kf), base)
            kinds.append("new")
    mp_id = np.array(mp_id, np.int32)
    obs_ptr = np.array(obs_ptr, np.int32)
    obs_kf = np.array(obs_kf, np.int32)
    frame_obs = np.flatnonzero(obs_kf == last).astype(np.int32)
    frame_kp_order = rng.permutation(frame_obs).astype(np.int32)
    in_frame = np.zeros(len(mp_id), bool)
    for m in range(len(mp_id)):
        in_frame[m] = (obs_kf[obs_ptr[m]:obs_ptr[m + 1]] == last).any()
    local = [int(i) for i in mp_id[~in_frame]] + [int(i) for i in mp_id[in_frame][:5]]   # a few observed ones: must be skipped
    local = list(rng.permutation(local))
    return dict(calib=calib, cell_size=35, kf_id=np.arange(10, 10 + n_kf, dtype=np.int32), kf_pose=poses, mp_id=mp_id,
                mp_wpt=np.array(mp_wpt), mp_is3d=np.array(mp_is3d, np.uint8), obs_ptr=obs_ptr, obs_kf=obs_kf,
                obs_px=np.array(obs_px, np.float32), obs_desc=np.array(obs_desc, np.uint8), frame_kp_order=frame_kp_order,
                local=np.array(local, np.int32), num_kp3d=int(in_frame.sum()))


def make_relpose_problem(n: int, seed: int, outlier_frac: float = 0.2, px_noise: float = 0.5, baseline: float = 0.5, fx: float = 579.4,
                         fy: float = 579.4, cx: float = 320.0, cy: float = 240.0):
    """Two views of n points for the map initialisation (previous keyframe = view 1, current frame = view 2): unit bearings
    bv1 / bv2 built from noisy pixel observations, a fraction of gross mismatches, and the true relative pose
    X1 = R12 X2 + t12 (|t12| = baseline)."""
    rng = np.random.RandomState(seed)
    R12 = so3_exp(np.array([0.04, -0.07, 0.02]) * (1 + 0.3 * rng.randn(3)))
    t12 = np.array([1.0, 0.15 * rng.randn(), 0.2 * rng.randn()])
    t12 = baseline * t12 / np.linalg.norm(t12)
    X1 = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2.2, 2.2, n), rng.uniform(3, 9, n)], 1)
    X2 = (X1 - t12) @ R12          # R12^T (X1 - t12)
    bad = rng.rand(n) < outlier_frac

    def bearings(X, shift):
        px = np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1) + px_noise * rng.randn(n, 2)
        if shift:
            px[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
        px = px.astype(np.float32).astype(np.float64)
        b = np.stack([(px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy, np.ones(n)], 1)
        return b / np.linalg.norm(b, axis=1, keepdims=True)
    return dict(bv1=np.ascontiguousarray(bearings(X1, False)), bv2=np.ascontiguousarray(bearings(X2, True)), R12=R12, t12=t12, bad=bad,
                K=(fx, fy, cx, cy))


# ----------------------------------------------------------------------------------------------
# A camera moving in front of a textured plane (rotation + translation): frames for System-level tests
def plane_camera_pose(k: int):
    """Twc of frame k: smooth translation (~2 px / frame at the start, so that the 40-px initialisation parallax is reached within ~20
    frames) with a slow rotation about all three axes.  Returns (R_wc 3x3, t_wc 3)."""
    t = np.array([0.9 * np.sin(0.016 * k), 0.45 * np.sin(0.011 * k), 0.3 * (1.0 - np.cos(0.01 * k))])
    w = np.array([0.03 * np.sin(0.02 * k), 0.05 * np.sin(0.013 * k), 0.04 * np.sin(0.009 * k)])
    return so3_exp(w), t


def render_plane(canvas: np.ndarray, width: int, height: int, f: float, R_wc: np.ndarray, t_wc: np.ndarray, plane_z: float = 4.0) -> np.ndarray:
    """The plane z = plane_z (world frame) textured with `canvas` (1 canvas pixel = plane_z / f metres, canvas centre on the optical
    axis of the identity pose), seen by a pinhole camera (f, principal point at the image centre) at pose Twc: inverse mapping of
    every output pixel, bilinear sampling, u8."""
    ch, cw = canvas.shape
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    d = np.stack([(xs - width * 0.5) / f, (ys - height * 0.5) / f, np.ones_like(xs)], -1) @ R_wc.T
    lam = (plane_z - t_wc[2]) / d[..., 2]
    X = t_wc[0] + lam * d[..., 0]
    Y = t_wc[1] + lam * d[..., 1]
    s = f / plane_z
    u = cw * 0.5 + s * X
    v = ch * 0.5 + s * Y
    u0 = np.clip(np.floor(u).astype(np.int64), 0, cw - 2)
    v0 = np.clip(np.floor(v).astype(np.int64), 0, ch - 2)
    a = np.clip(u - u0, 0.0, 1.0)
    b = np.clip(v - v0, 0.0, 1.0)
    c = canvas.astype(np.float64)
    out = (c[v0, u0] * (1 - a) * (1 - b) + c[v0, u0 + 1] * a * (1 - b) + c[v0 + 1, u0] * (1 - a) * b + c[v0 + 1, u0 + 1] * a * b)
    out[(lam <= 0) | (u < 0) | (v < 0) | (u > cw - 1) | (v > ch - 1)] = 0
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def plane_stream_frame(canvas: np.ndarray, k: int, width: int, height: int, f: float, noise_seed: int | None = None) -> np.ndarray:
    R, t = plane_camera_pose(k)
    g = render_plane(canvas, width, height, f, R, t)
    if noise_seed is not None:
        rng = np.random.RandomState(noise_seed + k)
        g = np.clip(g.astype(np.int16) + rng.randint(-4, 5, g.shape), 0, 255).astype(np.uint8)
    return gray_to_rgba(g)
