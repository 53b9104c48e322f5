"""TEST INFRASTRUCTURE: numpy-facing loaders for the two CPU oracles.

  orc  = oracle/libalva_oracle.so   (plain-C restatement)
  ref  = oracle/_ref/libalva_ref.so (the compiled reference itself)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORC_PATH = ROOT / "oracle" / "libalva_oracle.so"
REF_PATH = ROOT / "oracle" / "_ref" / "libalva_ref.so"

_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def ref_available() -> bool:
    return REF_PATH.exists()


_orc = None
_ref = None


def orc_lib():
    global _orc
    if _orc is None:
        newest = max(f.stat().st_mtime for f in (ROOT / "oracle").glob("alva_oracle*.[ch]"))
        if not ORC_PATH.exists() or ORC_PATH.stat().st_mtime < newest:
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
        _orc = C.CDLL(str(ORC_PATH))
    return _orc


def ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(str(REF_PATH))
        _ref.ref_build_info.restype = C.c_char_p
    return _ref


def pyr_dims(w, h, win, max_level):
    dims = np.zeros(2 * (max_level + 1), np.int32)
    n = orc_lib().orc_pyramid_dims(w, h, win, max_level, _p(dims))
    return [(int(dims[2 * l]), int(dims[2 * l + 1])) for l in range(n)]


def _pyr_call(fn, gray, win, max_level, with_dims):
    h, w = gray.shape
    dims = pyr_dims(w, h, win, max_level)
    gs = [np.zeros((lh + 2 * win, lw + 2 * win), np.uint8) for lw, lh in dims]
    ds = [np.zeros((lh + 2 * win, lw + 2 * win, 2), np.int16) for lw, lh in dims]
    nmax = max_level + 1
    gp = (_vp * nmax)(*[g.ctypes.data for g in gs] + [None] * (nmax - len(gs)))
    dp = (_vp * nmax)(*[d.ctypes.data for d in ds] + [None] * (nmax - len(ds)))
    gray = np.ascontiguousarray(gray)
    if with_dims:
        od = np.zeros(2 * nmax, np.int32)
        lv = fn(_p(gray), w, h, win, max_level, gp, dp, _p(od))
        assert lv + 1 == len(dims), (lv, dims)
    else:
        n = fn(_p(gray), w, h, win, max_level, gp, dp)
        assert n == len(dims)
    return gs, ds


class Orc:
    """Plain-C restatement."""
    _pfx = "orc_"
    _pnp = "orc_pnp_refine"

    @classmethod
    def fast(cls, gray, threshold=20, cap=200000):
        h, w = gray.shape
        xy = np.zeros((cap, 2), np.int32)
        sc = np.zeros(cap, np.int32)
        g = np.ascontiguousarray(gray)
        if cls._pfx == "ref_":
            n = ref_lib().ref_fast(_p(g), w, h, threshold, 1, _p(xy), _p(sc), cap)
        else:
            n = orc_lib().orc_fast(_p(g), w, h, threshold, _p(xy), _p(sc), cap)
        return xy[:n].copy(), sc[:n].copy()

    @classmethod
    def orb(cls, gray, nfeatures=2000, scale=1.2, nlevels=8, fast_thr=20, describe=True, cap=20000):
        h, w = gray.shape
        kp = np.zeros((cap, 6), np.float32)
        desc = np.zeros((cap, 32), np.uint8)
        fn = getattr(cls._lib(), cls._pfx + "orb_detect_and_compute")
        n = fn(_p(np.ascontiguousarray(gray)), w, h, nfeatures, _f(scale), nlevels, fast_thr, int(describe), _p(kp), _p(desc), cap)
        return kp[:n].copy(), desc[:n].copy()

    @classmethod
    def orb_pyramid(cls, gray, scale=1.2, nlevels=8):
        """cv::ORB's image pyramid (INTER_LINEAR_EXACT resize chain, orb.cpp:1041-1099) as a list of uint8 arrays"""
        import ctypes as C
        h, w = gray.shape
        fn = getattr(cls._lib(), cls._pfx + "orb_pyramid")
        fn.restype = C.c_long
        dims = np.zeros((nlevels, 2), np.int32)
        g = np.ascontiguousarray(gray)
        total = fn(_p(g), w, h, _f(scale), nlevels, None, _p(dims))
        out = np.zeros(total, np.uint8)
        fn(_p(g), w, h, _f(scale), nlevels, _p(out), _p(dims))
        levels, off = [], 0
        for lw, lh in dims:
            levels.append(out[off:off + lw * lh].reshape(lh, lw).copy())
            off += lw * lh
        return levels

    @classmethod
    def cell_mineig(cls, gray, x, y, cell):
        h, w = gray.shape
        blur = np.zeros((cell, cell), np.uint8)
        eig = np.zeros((cell, cell), np.float32)
        getattr(cls._lib(), cls._pfx + "cell_mineig")(_p(np.ascontiguousarray(gray)), w, h, x, y, cell, _p(blur), _p(eig))
        return blur, eig

    @classmethod
    def corner_subpix(cls, gray, pts):
        h, w = gray.shape
        p = np.ascontiguousarray(pts, np.float32).copy()
        getattr(cls._lib(), cls._pfx + "corner_subpix")(_p(np.ascontiguousarray(gray)), w, h, _p(p), len(p))
        return p

    @classmethod
    def detect_grid(cls, gray, cell, occupied=None, roi=None, max_quality=0.001, cap=20000):
        h, w = gray.shape
        occ = np.zeros((0, 2), np.float32) if occupied is None else np.ascontiguousarray(occupied, np.float32)
        if roi is None:
            roi = (20, 20, w - 40, h - 40)
        mq = C.c_double(max_quality)
        out = np.zeros((cap, 2), np.float32)
        n = getattr(cls._lib(), cls._pfx + "detect_grid")(_p(np.ascontiguousarray(gray)), w, h, cell, _p(occ), len(occ), roi[0], roi[1],
                                                          roi[2], roi[3], C.byref(mq), _p(out), cap)
        return out[:n].copy(), mq.value

    @classmethod
    def local_ba(cls, pb, max_iters=5, ftol=0.0, huber_chi2=5.9915, inv_depth=True):
        """pb: dict from synth.make_ba_problem (or make_ba_problem_xyz).  Returns dict(poses, pts, chi2, depth, info, ok)."""
        poses = np.ascontiguousarray(pb["poses"], np.float64).copy()
        kfc = np.ascontiguousarray(pb["kf_const"], np.uint8)
        calib = np.ascontiguousarray(pb["calib"], np.float64)
        akf = np.ascontiguousarray(pb["anchor_kf"], np.int32)
        auv = np.ascontiguousarray(pb["anchor_uv"], np.float64)
        pts = np.ascontiguousarray(pb["inv_depth"] if inv_depth else pb["pts_xyz"], np.float64).copy()
        okf = np.ascontiguousarray(pb["obs_kf"], np.int32)
        opt = np.ascontiguousarray(pb["obs_pt"], np.int32)
        ouv = np.ascontiguousarray(pb["obs_uv"], np.float64)
        nobs = len(okf)
        chi2 = np.zeros(nobs)
        depth = np.zeros(nobs, np.uint8)
        info = np.zeros(9)
        fn = getattr(cls._lib(), cls._pfx + "local_ba")
        ok = fn(len(poses), _p(poses), _p(kfc), _p(calib), int(inv_depth), len(akf), _p(akf), _p(auv), _p(pts), nobs, _p(okf), _p(opt),
                _p(ouv), max_iters, _d(ftol), _d(huber_chi2), _p(chi2), _p(depth), _p(info))
        return dict(ok=bool(ok), poses=poses, pts=pts, chi2=chi2, depth=depth, info=info)

    @classmethod
    def pnp_refine(cls, uv, wpt, pose7, K, max_iters=5, chi2th=5.9915, robust=True, l2=True):
        uv = np.ascontiguousarray(uv, np.float64)
        wpt = np.ascontiguousarray(wpt, np.float64)
        n = len(uv)
        pose = np.ascontiguousarray(pose7, np.float64).copy()
        out = np.zeros(max(n, 1), np.int32)
        nout = C.c_int(0)
        info = np.zeros(8)
        fn = getattr(cls._lib(), cls._pnp)
        ok = fn(_p(uv), _p(wpt), n, _p(pose), max_iters, _f(chi2th), int(robust), int(l2), _f(K[0]), _f(K[1]), _f(K[2]), _f(K[3]),
                _p(out), C.byref(nout), _p(info))
        return bool(ok), pose, out[:nout.value].copy(), info
    _lib = staticmethod(lambda: orc_lib())

    @classmethod
    def lk(cls, prev, curr, pts, init, num_levels=3, win=9, built=3, max_iters=30, eps=0.01):
        h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32)
        nxt = np.ascontiguousarray(init, np.float32).copy()
        n = len(pts)
        st = np.zeros(n, np.uint8)
        er = np.zeros(n, np.float32)
        fn = getattr(cls._lib(), cls._pfx + "lk")
        rc = fn(_p(np.ascontiguousarray(prev)), _p(np.ascontiguousarray(curr)), w, h, win, built, num_levels, max_iters,
                _f(eps), _p(pts), _p(nxt), _p(st), _p(er), n)
        assert rc == 0
        return nxt, st, er

    @classmethod
    def fbklt(cls, prev, curr, pts, prior, num_levels=3, win=9, built=3, err_thresh=30.0, fb_dist=0.5, max_iters=30, eps=0.01):
        h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32)
        pr = np.ascontiguousarray(prior, np.float32).copy()
        n = len(pts)
        st = np.zeros(n, np.uint8)
        fn = getattr(cls._lib(), cls._pfx + "fbklt")
        rc = fn(_p(np.ascontiguousarray(prev)), _p(np.ascontiguousarray(curr)), w, h, win, built, num_levels, _f(err_thresh),
                _f(fb_dist), max_iters, _f(eps), _p(pts), _p(pr), _p(st), n)
        assert rc == 0
        return pr, st

    @staticmethod
    def rgba2gray(rgba):
        h, w, _ = rgba.shape
        out = np.empty((h, w), np.uint8)
        orc_lib().orc_rgba2gray(_p(np.ascontiguousarray(rgba)), w, h, _p(out))
        return out

    @staticmethod
    def build_pyramid(gray, win=9, max_level=3):
        return _pyr_call(orc_lib().orc_build_pyramid, gray, win, max_level, False)

    @staticmethod
    def bf_match(q, t):
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        idx = np.empty(len(q), np.int32)
        dist = np.empty(len(q), np.int32)
        orc_lib().orc_bf_match_hamming(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
        return idx, dist


    @staticmethod
    def p3p_lmeds(bv, wpt, max_iters=100, err=3.0, fx=579.4, fy=579.4, seed=12345):
        bv = np.ascontiguousarray(bv, np.float64)
        wpt = np.ascontiguousarray(wpt, np.float64)
        n = len(bv)
        R = np.zeros((3, 3))
        t = np.zeros(3)
        out = np.zeros(n, np.int32)
        nout = C.c_int(0)
        ok = orc_lib().orc_p3p_lmeds(_p(bv), _p(wpt), n, max_iters, _f(err), C.c_uint32(seed), _f(fx), _f(fy), _p(R), _p(t), _p(out),
                                     C.byref(nout))
        return bool(ok), R, t, out[:nout.value].copy()

    @staticmethod
    def orb_blur(gray):
        h, w = gray.shape
        out = np.empty((h, w), np.uint8)
        orc_lib().orc_orb_blur(_p(np.ascontiguousarray(gray)), w, h, _p(out))
        return out

    @staticmethod
    def describe(gray, pts):
        h, w = gray.shape
        pts = np.ascontiguousarray(pts, np.float32)
        n = len(pts)
        desc = np.zeros((n, 32), np.uint8)
        valid = np.zeros(n, np.uint8)
        orc_lib().orc_describe(_p(np.ascontiguousarray(gray)), w, h, _p(pts), n, _p(desc), _p(valid))
        return desc, valid


def _dist_call(lib, name, arr, K, dist, shape):
    a = np.ascontiguousarray(arr)
    k = np.ascontiguousarray(dist, np.float64)
    out = np.zeros(shape, np.float32)
    getattr(lib, name)(_p(a), len(a), _d(float(K[0])), _d(float(K[1])), _d(float(K[2])), _d(float(K[3])), _p(k), _p(out))
    return out


def orc_undistort_points(px, K, dist):
    return _dist_call(orc_lib(), "orc_undistort_points", np.asarray(px, np.float32), K, dist, (len(px), 2))


def ref_undistort_points(px, K, dist):
    return _dist_call(ref_lib(), "ref_undistort_points", np.asarray(px, np.float32), K, dist, (len(px), 2))


def orc_project_dist(P, K, dist):
    return _dist_call(orc_lib(), "orc_project_dist", np.asarray(P, np.float64), K, dist, (len(P), 2))


def ref_project_dist(P, K, dist):
    return _dist_call(ref_lib(), "ref_project_dist", np.asarray(P, np.float64), K, dist, (len(P), 2))


def orc_clahe(gray, clip=3.0, tiles=(12, 9)):
    g = np.ascontiguousarray(gray)
    out = np.empty_like(g)
    orc_lib().orc_clahe(_p(g), g.shape[1], g.shape[0], _d(float(clip)), int(tiles[0]), int(tiles[1]), _p(out))
    return out


def ref_clahe(gray, clip=3.0, tiles=(12, 9)):
    g = np.ascontiguousarray(gray)
    out = np.empty_like(g)
    ref_lib().ref_clahe(_p(g), g.shape[1], g.shape[0], _d(float(clip)), int(tiles[0]), int(tiles[1]), _p(out))
    return out


def ref_match_to_map(pb, max_proj_err=2.0, dist_ratio=0.2, num_kp3d=None):
    """The reference's own Mapper::matchToMap on a map built with its own classes.  Returns (matches {kpId: mpId}, aux) where
    aux holds what the reference's containers determine: Tcw of the keyframes, the frame's keypoint grid, the iteration order of
    the local-map unordered_set."""
    nkf, nmp = len(pb["kf_id"]), len(pb["mp_id"])
    kfQ, kfT = np.zeros((nkf, 4)), np.zeros((nkf, 3))
    nfk = len(pb["frame_kp_order"])
    gc, ncw = C.c_int(0), C.c_int(0)
    cellPtr, cellKp = np.zeros(4096, np.int32), np.zeros(max(nfk, 1), np.int32)
    nloc = len(pb["local"])
    localOrder = np.zeros(max(nloc, 1), np.int32)
    mk, mm = np.zeros(max(nfk, 1), np.int32), np.zeros(max(nfk, 1), np.int32)
    a = {k: np.ascontiguousarray(v) for k, v in pb.items() if isinstance(v, np.ndarray)}
    n = ref_lib().ref_match_to_map(_p(a["calib"]), int(pb["cell_size"]), nkf, _p(a["kf_id"]), _p(a["kf_pose"]), nmp, _p(a["mp_id"]),
                                   _p(a["mp_wpt"]), _p(a["mp_is3d"]), _p(a["obs_ptr"]), _p(a["obs_kf"]), _p(a["obs_px"]), _p(a["obs_desc"]),
                                   nfk, _p(a["frame_kp_order"]), int(pb["num_kp3d"] if num_kp3d is None else num_kp3d), nloc, _p(a["local"]),
                                   _f(max_proj_err), _f(dist_ratio), _p(kfQ), _p(kfT), C.byref(gc), C.byref(ncw), _p(cellPtr), _p(cellKp),
                                   _p(localOrder), _p(mk), _p(mm))
    aux = dict(kf_q=kfQ, kf_t=kfT, grid_cells=gc.value, num_cells_w=ncw.value, cell_ptr=cellPtr[:gc.value + 1].copy(),
               cell_kp=cellKp[:cellPtr[gc.value]].copy(), local_order=localOrder[:nloc].copy())
    return {int(mk[i]): int(mm[i]) for i in range(n)}, aux


def py_match_to_map_aux(pb):
    """A stand-in for the reference-determined inputs when the reference is absent (GPU box): T_cw from the poses with numpy,
    the keypoint grid filled in frame_kp_order like Frame::addKeypointToGrid, the local list in its given order.  Any such
    choice is a valid input; oracle and HIP path must agree on it."""
    from scipy.spatial.transform import Rotation
    nkf = len(pb["kf_id"])
    kfq, kft = np.zeros((nkf, 4)), np.zeros((nkf, 3))
    for k in range(nkf):
        R = Rotation.from_quat(pb["kf_pose"][k, 3:]).as_matrix()
        q = Rotation.from_matrix(R.T).as_quat()
        kfq[k] = q if q[3] >= 0 else -q
        kft[k] = -R.T @ pb["kf_pose"][k, :3]
    W, H, cs = pb["calib"][8], pb["calib"][9], pb["cell_size"]
    ncw, nch = int(np.ceil(np.float32(W) / cs)), int(np.ceil(np.float32(H) / cs))
    cells = [[] for _ in range(ncw * nch)]
    owner = np.searchsorted(pb["obs_ptr"], pb["frame_kp_order"], side="right") - 1
    for o, m in zip(pb["frame_kp_order"], owner):
        px = pb["obs_px"][o]
        cells[int(np.floor(px[1] / np.float32(cs))) * ncw + int(np.floor(px[0] / np.float32(cs)))].append(int(pb["mp_id"][m]))
    cell_ptr = np.zeros(len(cells) + 1, np.int32)
    cell_ptr[1:] = np.cumsum([len(c) for c in cells])
    return dict(kf_q=kfq, kf_t=kft, grid_cells=len(cells), num_cells_w=ncw, cell_ptr=cell_ptr,
                cell_kp=np.array([i for c in cells for i in c], np.int32), local_order=np.asarray(pb["local"], np.int32))


def flatten_match_to_map(pb, aux):
    """ids -> indices: what a host integration hands to the device path"""
    idx_of = {int(v): i for i, v in enumerate(pb["mp_id"])}
    cell_mp = np.array([idx_of[int(i)] for i in aux["cell_kp"]], np.int32)
    local = np.array([idx_of[int(i)] for i in aux["local_order"]], np.int32)
    return cell_mp, local


def orc_match_to_map(pb, aux, max_proj_err=2.0, dist_ratio=0.2, num_kp3d=None):
    cell_mp, local = flatten_match_to_map(pb, aux)
    nmp = len(pb["mp_id"])
    out = np.full(nmp, -1, np.int32)
    a = {k: np.ascontiguousarray(v) for k, v in pb.items() if isinstance(v, np.ndarray)}
    kfq, kft, cp = np.ascontiguousarray(aux["kf_q"]), np.ascontiguousarray(aux["kf_t"]), np.ascontiguousarray(aux["cell_ptr"], np.int32)
    orc_lib().orc_match_to_map(_p(a["calib"]), int(pb["cell_size"]), int(aux["num_cells_w"]), int(aux["grid_cells"]), _p(cp), _p(cell_mp),
                               len(pb["kf_id"]), _p(kfq), _p(kft), nmp, _p(a["mp_wpt"]), _p(a["mp_is3d"]), _p(a["obs_ptr"]), _p(a["obs_kf"]),
                               _p(a["obs_px"]), _p(a["obs_desc"]), len(pb["kf_id"]) - 1, int(pb["num_kp3d"] if num_kp3d is None else num_kp3d),
                               len(local), _p(local), _f(max_proj_err), _f(dist_ratio), _p(out))
    return {int(pb["mp_id"][m]): int(pb["mp_id"][out[m]]) for m in range(nmp) if out[m] >= 0}


def _tri_out(n):
    return dict(lpt=np.zeros((n, 3)), wpt=np.zeros((n, 3)), inv_depth=np.zeros(n), status=np.zeros(n, np.uint8), parallax=np.zeros(n))


def orc_triangulate(T, group, bvl, bvr, unpxl, unpxr, K, max_err=3.0):
    """oracle restatement of the per-keypoint part of Mapper::triangulateTemporal"""
    n = len(bvl)
    o = _tri_out(n)
    T, group = np.ascontiguousarray(T, np.float64), np.ascontiguousarray(group, np.int32)
    bvl, bvr = np.ascontiguousarray(bvl, np.float64), np.ascontiguousarray(bvr, np.float64)
    unpxl, unpxr = np.ascontiguousarray(unpxl, np.float32), np.ascontiguousarray(unpxr, np.float32)
    orc_lib().orc_triangulate(n, _p(T), _p(group), _p(bvl), _p(bvr), _p(unpxl), _p(unpxr), _d(K[0]), _d(K[1]), _d(K[2]), _d(K[3]),
                              _f(max_err), _p(o["lpt"]), _p(o["wpt"]), _p(o["inv_depth"]), _p(o["status"]), _p(o["parallax"]))
    return o


def ref_triangulate(pose_kf, pose_new, group, bvl, bvr, unpxl, unpxr, K, max_err=3.0):
    """the reference's own pieces (Sophus, MultiViewGeometry::triangulate, CameraCalibration); also returns the transform
    blocks T [nGroups, 36] it used"""
    n, ng = len(bvl), len(pose_kf)
    o = _tri_out(n)
    T = np.zeros((ng, 36))
    pose_kf, pose_new = np.ascontiguousarray(pose_kf, np.float64), np.ascontiguousarray(pose_new, np.float64)
    group = np.ascontiguousarray(group, np.int32)
    bvl, bvr = np.ascontiguousarray(bvl, np.float64), np.ascontiguousarray(bvr, np.float64)
    unpxl, unpxr = np.ascontiguousarray(unpxl, np.float32), np.ascontiguousarray(unpxr, np.float32)
    ref_lib().ref_triangulate(n, ng, _p(pose_kf), _p(pose_new), _p(group), _p(bvl), _p(bvr), _p(unpxl), _p(unpxr), _d(K[0]), _d(K[1]),
                              _d(K[2]), _d(K[3]), _f(max_err), _p(T), _p(o["lpt"]), _p(o["wpt"]), _p(o["inv_depth"]), _p(o["status"]),
                              _p(o["parallax"]))
    return o, T


class Ref:
    """The compiled reference (OpenCV 4.5.5 / Ceres 2.0.0 / OpenGV / AlvaAR slam sources)."""
    _pfx = "ref_"
    _pnp = "ref_ceres_pnp_nocap"

    @classmethod
    def fast(cls, gray, threshold=20, cap=200000):
        h, w = gray.shape
        xy = np.zeros((cap, 2), np.int32)
        sc = np.zeros(cap, np.int32)
        g = np.ascontiguousarray(gray)
        if cls._pfx == "ref_":
            n = ref_lib().ref_fast(_p(g), w, h, threshold, 1, _p(xy), _p(sc), cap)
        else:
            n = orc_lib().orc_fast(_p(g), w, h, threshold, _p(xy), _p(sc), cap)
        return xy[:n].copy(), sc[:n].copy()

    @classmethod
    def orb(cls, gray, nfeatures=2000, scale=1.2, nlevels=8, fast_thr=20, describe=True, cap=20000):
        h, w = gray.shape
        kp = np.zeros((cap, 6), np.float32)
        desc = np.zeros((cap, 32), np.uint8)
        fn = getattr(cls._lib(), cls._pfx + "orb_detect_and_compute")
        n = fn(_p(np.ascontiguousarray(gray)), w, h, nfeatures, _f(scale), nlevels, fast_thr, int(describe), _p(kp), _p(desc), cap)
        return kp[:n].copy(), desc[:n].copy()

    @classmethod
    def cell_mineig(cls, gray, x, y, cell):
        h, w = gray.shape
        blur = np.zeros((cell, cell), np.uint8)
        eig = np.zeros((cell, cell), np.float32)
        getattr(cls._lib(), cls._pfx + "cell_mineig")(_p(np.ascontiguousarray(gray)), w, h, x, y, cell, _p(blur), _p(eig))
        return blur, eig

    @classmethod
    def corner_subpix(cls, gray, pts):
        h, w = gray.shape
        p = np.ascontiguousarray(pts, np.float32).copy()
        getattr(cls._lib(), cls._pfx + "corner_subpix")(_p(np.ascontiguousarray(gray)), w, h, _p(p), len(p))
        return p

    @classmethod
    def detect_grid(cls, gray, cell, occupied=None, roi=None, max_quality=0.001, cap=20000):
        h, w = gray.shape
        occ = np.zeros((0, 2), np.float32) if occupied is None else np.ascontiguousarray(occupied, np.float32)
        if roi is None:
            roi = (20, 20, w - 40, h - 40)
        mq = C.c_double(max_quality)
        out = np.zeros((cap, 2), np.float32)
        n = getattr(cls._lib(), cls._pfx + "detect_grid")(_p(np.ascontiguousarray(gray)), w, h, cell, _p(occ), len(occ), roi[0], roi[1],
                                                          roi[2], roi[3], C.byref(mq), _p(out), cap)
        return out[:n].copy(), mq.value

    @classmethod
    def local_ba(cls, pb, max_iters=5, ftol=0.0, huber_chi2=5.9915, inv_depth=True):
        """pb: dict from synth.make_ba_problem (or make_ba_problem_xyz).  Returns dict(poses, pts, chi2, depth, info, ok)."""
        poses = np.ascontiguousarray(pb["poses"], np.float64).copy()
        kfc = np.ascontiguousarray(pb["kf_const"], np.uint8)
        calib = np.ascontiguousarray(pb["calib"], np.float64)
        akf = np.ascontiguousarray(pb["anchor_kf"], np.int32)
        auv = np.ascontiguousarray(pb["anchor_uv"], np.float64)
        pts = np.ascontiguousarray(pb["inv_depth"] if inv_depth else pb["pts_xyz"], np.float64).copy()
        okf = np.ascontiguousarray(pb["obs_kf"], np.int32)
        opt = np.ascontiguousarray(pb["obs_pt"], np.int32)
        ouv = np.ascontiguousarray(pb["obs_uv"], np.float64)
        nobs = len(okf)
        chi2 = np.zeros(nobs)
        depth = np.zeros(nobs, np.uint8)
        info = np.zeros(9)
        fn = getattr(cls._lib(), cls._pfx + "local_ba")
        ok = fn(len(poses), _p(poses), _p(kfc), _p(calib), int(inv_depth), len(akf), _p(akf), _p(auv), _p(pts), nobs, _p(okf), _p(opt),
                _p(ouv), max_iters, _d(ftol), _d(huber_chi2), _p(chi2), _p(depth), _p(info))
        return dict(ok=bool(ok), poses=poses, pts=pts, chi2=chi2, depth=depth, info=info)

    @classmethod
    def pnp_refine(cls, uv, wpt, pose7, K, max_iters=5, chi2th=5.9915, robust=True, l2=True):
        uv = np.ascontiguousarray(uv, np.float64)
        wpt = np.ascontiguousarray(wpt, np.float64)
        n = len(uv)
        pose = np.ascontiguousarray(pose7, np.float64).copy()
        out = np.zeros(max(n, 1), np.int32)
        nout = C.c_int(0)
        info = np.zeros(8)
        fn = getattr(cls._lib(), cls._pnp)
        ok = fn(_p(uv), _p(wpt), n, _p(pose), max_iters, _f(chi2th), int(robust), int(l2), _f(K[0]), _f(K[1]), _f(K[2]), _f(K[3]),
                _p(out), C.byref(nout), _p(info))
        return bool(ok), pose, out[:nout.value].copy(), info
    _lib = staticmethod(lambda: ref_lib())

    @classmethod
    def lk(cls, prev, curr, pts, init, num_levels=3, win=9, built=3, max_iters=30, eps=0.01):
        h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32)
        nxt = np.ascontiguousarray(init, np.float32).copy()
        n = len(pts)
        st = np.zeros(n, np.uint8)
        er = np.zeros(n, np.float32)
        fn = getattr(cls._lib(), cls._pfx + "lk")
        rc = fn(_p(np.ascontiguousarray(prev)), _p(np.ascontiguousarray(curr)), w, h, win, built, num_levels, max_iters,
                _f(eps), _p(pts), _p(nxt), _p(st), _p(er), n)
        assert rc == 0
        return nxt, st, er

    @classmethod
    def fbklt(cls, prev, curr, pts, prior, num_levels=3, win=9, built=3, err_thresh=30.0, fb_dist=0.5, max_iters=30, eps=0.01):
        h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32)
        pr = np.ascontiguousarray(prior, np.float32).copy()
        n = len(pts)
        st = np.zeros(n, np.uint8)
        fn = getattr(cls._lib(), cls._pfx + "fbklt")
        rc = fn(_p(np.ascontiguousarray(prev)), _p(np.ascontiguousarray(curr)), w, h, win, built, num_levels, _f(err_thresh),
                _f(fb_dist), max_iters, _f(eps), _p(pts), _p(pr), _p(st), n)
        assert rc == 0
        return pr, st

    @staticmethod
    def rgba2gray(rgba):
        h, w, _ = rgba.shape
        out = np.empty((h, w), np.uint8)
        rc = ref_lib().ref_rgba2gray(_p(np.ascontiguousarray(rgba)), w, h, _p(out))
        assert rc == 0
        return out

    @staticmethod
    def build_pyramid(gray, win=9, max_level=3):
        return _pyr_call(ref_lib().ref_build_pyramid, gray, win, max_level, True)

    @staticmethod
    def bf_match(q, t):
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        idx = np.empty(len(q), np.int32)
        dist = np.empty(len(q), np.int32)
        ref_lib().ref_bf_match_hamming(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
        return idx, dist

    @staticmethod
    def orb_blur(gray):
        h, w = gray.shape
        b = 32
        out = np.empty((h + 2 * b, w + 2 * b), np.uint8)
        ref_lib().ref_orb_blur(_p(np.ascontiguousarray(gray)), w, h, b, _p(out))
        return out[b:b + h, b:b + w].copy()

    @staticmethod
    def describe(gray, pts):
        h, w = gray.shape
        pts = np.ascontiguousarray(pts, np.float32)
        n = len(pts)
        desc = np.zeros((n, 32), np.uint8)
        valid = np.zeros(n, np.uint8)
        ref_lib().ref_describe(_p(np.ascontiguousarray(gray)), w, h, _p(pts), n, _p(desc), _p(valid))
        return desc, valid

    @staticmethod
    def p3p_lmeds(bv, wpt, max_iters=100, err=3.0, fx=579.4, fy=579.4, do_random=False):
        bv = np.ascontiguousarray(bv, np.float64)
        wpt = np.ascontiguousarray(wpt, np.float64)
        n = len(bv)
        R = np.zeros((3, 3))
        t = np.zeros(3)
        out = np.zeros(n, np.int32)
        nout = C.c_int(0)
        ok = ref_lib().ref_p3p_lmeds(_p(bv), _p(wpt), n, max_iters, _f(err), int(do_random), _f(fx), _f(fy), _p(R), _p(t), _p(out),
                                     C.byref(nout))
        return bool(ok), R, t, out[:nout.value].copy()


# ---- f2b: two-view initialisation (compute5ptEssentialMatrix).  Models are (R 3x3, t 3): X1 = R X2 + t -------------------------------
Ref.orb_pyramid = classmethod(Orc.orb_pyramid.__func__)   # same marshalling, ref_orb_pyramid (cv::resize INTER_LINEAR_EXACT chain)


def _bv(a):
    return np.ascontiguousarray(a, np.float64)


def _split(m12):
    return m12[:9].reshape(3, 3).copy(), m12[9:].copy()


def _join(R, t):
    return np.concatenate([np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64).ravel()])


def relpose_sturm_roots(coeffs, which="orc"):
    c = np.ascontiguousarray(coeffs, np.float64)
    r = np.zeros(len(c))
    fn = orc_lib().orc_sturm_roots if which == "orc" else ref_lib().ref_sturm_roots
    n = fn(_p(c), len(c), _p(r))
    return r[:n].copy()


def relpose_nullspace(bv1, bv2, which="orc"):
    EE = np.zeros((9, 4))
    (orc_lib().orc_nister_nullspace if which == "orc" else ref_lib().ref_nister_nullspace)(_p(_bv(bv1)), _p(_bv(bv2)), _p(EE))
    return EE


def relpose_compose_a(EE, which="orc"):
    A = np.zeros((10, 20))
    (orc_lib().orc_nister_compose_a if which == "orc" else ref_lib().ref_nister_compose_a)(_p(_bv(EE)), _p(A))
    return A


def relpose_fivept(bv1, bv2, which="orc"):
    E = np.zeros((10, 9))
    n = (orc_lib().orc_fivept_nister if which == "orc" else ref_lib().ref_fivept_nister)(_p(_bv(bv1)), _p(_bv(bv2)), _p(E))
    return E[:n].reshape(n, 3, 3).copy()


def relpose_model(bv1, bv2, idx8, which="orc"):
    idx = np.ascontiguousarray(idx8, np.int32)
    m = np.zeros(12)
    fn = orc_lib().orc_relpose_model if which == "orc" else ref_lib().ref_relpose_model
    ok = fn(_p(_bv(bv1)), _p(_bv(bv2)), len(bv1), _p(idx), _p(m))
    return (bool(ok),) + _split(m)


def relpose_scores(bv1, bv2, R, t, which="orc"):
    n = len(bv1)
    s = np.zeros(n)
    (orc_lib().orc_relpose_scores if which == "orc" else ref_lib().ref_relpose_scores)(_p(_bv(bv1)), _p(_bv(bv2)), n, _p(_join(R, t)), _p(s))
    return s


def relpose_optimize(bv1, bv2, inliers, R, t, which="orc"):
    inl = np.ascontiguousarray(inliers, np.int32)
    o = np.zeros(12)
    info = np.zeros(3, np.int32)
    if which == "orc":
        orc_lib().orc_relpose_optimize(_p(_bv(bv1)), _p(_bv(bv2)), len(bv1), _p(inl), len(inl), _p(_join(R, t)), _p(o), _p(info))
    else:
        ref_lib().ref_relpose_optimize(_p(_bv(bv1)), _p(_bv(bv2)), len(bv1), _p(inl), len(inl), _p(_join(R, t)), _p(o))
    return _split(o) + (info,)


def relpose_draw_samples(n, count, seed=12345):
    s = np.zeros((count, 8), np.int32)
    orc_lib().orc_relpose_draw_samples(n, count, C.c_uint32(seed), _p(s))
    return s


def relpose_ransac(bv1, bv2, max_iters=100, err=3.0, fx=579.4, fy=579.4, seed=12345, which="orc"):
    """-> ok, R, t (RANSAC model before refinement), inlier mask, iterations"""
    n = len(bv1)
    m = np.zeros(12)
    mask = np.zeros(n, np.uint8)
    info = np.zeros(2, np.int32)
    if which == "orc":
        ok = orc_lib().orc_relpose_ransac(_p(_bv(bv1)), _p(_bv(bv2)), n, max_iters, _f(err), C.c_uint32(seed), _f(fx), _f(fy), _p(m), _p(mask),
                                          _p(info))
    else:
        ok = ref_lib().ref_relpose_ransac(_p(_bv(bv1)), _p(_bv(bv2)), n, max_iters, _f(err), _f(fx), _f(fy), _p(m), _p(mask), _p(info))
    return (bool(ok),) + _split(m) + (mask.astype(bool), int(info[0]))


def compute_5pt(bv1, bv2, max_iters=100, err=3.0, optimize=True, fx=579.4, fy=579.4, seed=12345, which="orc"):
    """MultiViewGeometry::compute5ptEssentialMatrix -> ok, Rwc, twc, outlier indices"""
    n = len(bv1)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    out = np.zeros(max(n, 1), np.int32)
    nout = C.c_int(0)
    if which == "orc":
        ok = orc_lib().orc_compute_5pt(_p(_bv(bv1)), _p(_bv(bv2)), n, max_iters, _f(err), int(optimize), C.c_uint32(seed), _f(fx), _f(fy), _p(R),
                                       _p(t), _p(out), C.byref(nout))
    else:
        ok = ref_lib().ref_compute_5pt(_p(_bv(bv1)), _p(_bv(bv2)), n, max_iters, _f(err), int(optimize), _f(fx), _f(fy), _p(R), _p(t), _p(out),
                                       C.byref(nout))
    return bool(ok), R, t, out[:nout.value].copy()
