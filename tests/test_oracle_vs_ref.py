"""CPU: the plain-C restatement (oracle/alva_oracle.c) against the compiled reference
(oracle/_ref/libalva_ref.so) on seeded inputs.  This is what pins the restatement."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref

pytestmark = pytest.mark.ref


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (64, 48, 2), (1280, 720, 3), (36, 20, 4)])
def test_gray_bit_exact(w, h, seed):
    rgba = synth.random_rgba(w, h, seed)
    assert np.array_equal(Orc.rgba2gray(rgba), Ref.rgba2gray(rgba))


@pytest.mark.parametrize("w,h,levels", [(640, 480, 3), (1280, 720, 3), (100, 76, 3), (52, 44, 3), (333, 201, 2)])
def test_pyramid_bit_exact(w, h, levels):
    canvas = synth.texture_canvas(w, h, seed=w + h)
    g = synth.frame_gray(canvas, 3, w, h, noise_seed=11)
    og, od = Orc.build_pyramid(g, 9, levels)
    rg, rd = Ref.build_pyramid(g, 9, levels)
    assert len(og) == len(rg)
    for l in range(len(og)):
        assert np.array_equal(og[l], rg[l]), f"gray level {l}"
        assert np.array_equal(od[l], rd[l]), f"deriv level {l}"


@pytest.mark.parametrize("nq,nt,seed", [(1, 1, 0), (17, 33, 1), (300, 257, 2)])
def test_bf_match_bit_exact(nq, nt, seed):
    rng = np.random.RandomState(seed)
    q = rng.randint(0, 256, (nq, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (nt, 32)).astype(np.uint8)
    t[nt // 2] = t[0]  # force exact ties: lowest index must win
    if nq > 2:
        q[2] = t[0]
    oi, od = Orc.bf_match(q, t)
    ri, rd = Ref.bf_match(q, t)
    assert np.array_equal(oi, ri) and np.array_equal(od, rd)


def _test_points(w, h, n, seed):
    rng = np.random.RandomState(seed)
    pts = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1).astype(np.float32)
    # exercise the border rule (31 / size-31 after cvRound, incl. .5 ties) and integer positions
    pts[:8] = [[30.5, 100], [31.5, 100], [30.49, 100], [w - 31.5, 50], [w - 31.49, 50], [100, 30.5], [100, h - 31.5], [31, 31]]
    pts[8:16] = np.round(pts[8:16])
    return pts


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (200, 120, 2), (1280, 720, 3)])
def test_orb_blur_bit_exact(w, h, seed):
    g = synth.frame_gray(synth.texture_canvas(w, h, seed), 2, w, h, noise_seed=seed)
    assert np.array_equal(Orc.orb_blur(g), Ref.orb_blur(g))


@pytest.mark.parametrize("w,h,n,seed", [(640, 480, 500, 1), (200, 120, 100, 2)])
def test_describe_bit_exact(w, h, n, seed):
    g = synth.frame_gray(synth.texture_canvas(w, h, seed), 2, w, h, noise_seed=seed)
    pts = _test_points(w, h, n, seed)
    od, ov = Orc.describe(g, pts)
    rd, rv = Ref.describe(g, pts)
    assert np.array_equal(ov, rv)
    assert np.array_equal(od, rd)
    assert 0 < ov.sum() < n


def klt_case(w, h, n, seed, shift=(2, 1), noise=True):
    canvas = synth.texture_canvas(w, h, seed)
    prev = synth.frame_gray(canvas, 3, w, h, noise_seed=11 if noise else None)
    curr = synth.frame_gray(canvas, 3 + 1, w, h, noise_seed=12 if noise else None)  # moves by (2,1)
    rng = np.random.RandomState(seed)
    pts = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)
    pts[: n // 4] = np.round(pts[: n // 4])
    pts[0] = (0.0, 0.0)
    pts[1] = (w - 1.0, h - 1.0)
    pts[2] = (-20.0, 5.0)   # far outside: window check
    init = pts + rng.uniform(-1.5, 1.5, pts.shape).astype(np.float32)
    return prev, curr, pts, init.astype(np.float32)


@pytest.mark.parametrize("w,h,n,levels,seed", [(640, 480, 600, 3, 1), (640, 480, 300, 1, 2), (640, 480, 300, 0, 3), (200, 150, 200, 3, 4)])
def test_lk_bit_exact(w, h, n, levels, seed):
    prev, curr, pts, init = klt_case(w, h, n, seed)
    on, os_, oe = Orc.lk(prev, curr, pts, init, levels)
    rn, rs, re_ = Ref.lk(prev, curr, pts, init, levels)
    assert np.array_equal(os_, rs)
    assert np.array_equal(on.view(np.uint32), rn.view(np.uint32))
    ok = rs.astype(bool)
    assert np.array_equal(oe[ok].view(np.uint32), re_[ok].view(np.uint32))
    assert 0.2 * n < ok.sum() < n


@pytest.mark.parametrize("w,h,n,levels,seed", [(640, 480, 800, 3, 5), (640, 480, 400, 1, 6), (1280, 720, 500, 3, 7)])
def test_fbklt_bit_exact(w, h, n, levels, seed):
    prev, curr, pts, init = klt_case(w, h, n, seed)
    op, os_ = Orc.fbklt(prev, curr, pts, init, levels)
    rp, rs = Ref.fbklt(prev, curr, pts, init, levels)
    assert np.array_equal(os_, rs)
    assert np.array_equal(op.view(np.uint32), rp.view(np.uint32))
    assert 0.2 * n < rs.sum() < n


@pytest.mark.parametrize("n,seed,outl", [(2000, 3, 0.1), (192, 4, 0.3), (12, 5, 0.0), (501, 6, 0.45)])
def test_p3p_lmeds(n, seed, outl):
    pb = synth.make_pnp_problem(n, seed, outlier_frac=outl)
    ok1, R1, t1, o1 = Orc.p3p_lmeds(pb["bv"], pb["wpt"])
    ok2, R2, t2, o2 = Ref.p3p_lmeds(pb["bv"], pb["wpt"])
    assert ok1 == ok2 and ok2
    # FP64 restatement vs OpenGV/Eigen: same hypothesis wins, pose equal to rounding noise
    assert np.abs(t1 - t2).max() < 1e-8 and np.abs(R1 - R2).max() < 1e-8
    assert np.array_equal(o1, o2)
    from alvaar_amd.synth import quat_xyzw_to_rot
    Rgt = quat_xyzw_to_rot(pb["pose_gt"][3:])
    assert np.abs(R2 - Rgt).max() < 0.05


@pytest.mark.parametrize("n,seed,outl,noise", [(2000, 3, 0.1, 0.01), (192, 4, 0.3, 0.02), (30, 5, 0.0, 0.005), (500, 6, 0.2, 0.05)])
def test_pnp_refine(n, seed, outl, noise):
    pb = synth.make_pnp_problem(n, seed, outlier_frac=outl, pose_noise=noise)
    ok1, p1, o1, i1 = Orc.pnp_refine(pb["uv"], pb["wpt"], pb["pose_init"], pb["K"])
    ok2, p2, o2, i2 = Ref.pnp_refine(pb["uv"], pb["wpt"], pb["pose_init"], pb["K"])
    assert ok1 == ok2
    assert np.array_equal(o1, o2)
    assert i1[0] == i2[0] and i1[4] == i2[4], (i1, i2)          # same number of LM iterations in both solves
    assert np.allclose(i1[[1, 2, 5, 6]], i2[[1, 2, 5, 6]], rtol=1e-9)  # initial / final costs
    assert np.abs(p1 - p2).max() < 1e-9
    assert np.abs(p2[:3] - pb["pose_gt"][:3]).max() < 0.05


def ba_compare(a, b, pose_tol=1e-8, pt_tol=1e-7):
    assert a["ok"] == b["ok"]
    assert a["info"][0] == b["info"][0] and a["info"][3] == b["info"][3], (a["info"], b["info"])
    assert np.allclose(a["info"][1:3], b["info"][1:3], rtol=1e-8), (a["info"], b["info"])
    assert np.abs(a["poses"] - b["poses"]).max() < pose_tol
    assert np.abs(a["pts"] - b["pts"]).max() < pt_tol
    assert np.array_equal(a["depth"], b["depth"])
    assert np.allclose(a["chi2"], b["chi2"], rtol=1e-6, atol=1e-8)
    assert np.array_equal(a["chi2"] > 5.9915, b["chi2"] > 5.9915)


@pytest.mark.parametrize("nkf,npt,seed,iters,ftol", [(6, 200, 1, 5, 0.0), (20, 600, 42, 5, 0.0), (8, 300, 2, 5, 1e-3), (5, 80, 3, 2, 0.0)])
def test_local_ba_invdepth(nkf, npt, seed, iters, ftol):
    pb = synth.make_ba_problem(nkf, npt, seed)
    a = Orc.local_ba(pb, iters, ftol)
    b = Ref.local_ba(pb, iters, ftol)
    ba_compare(a, b)
    assert b["info"][2] < 0.2 * b["info"][1]


def xyz_problem(nkf, npt, seed):
    """XYZ mode has a residual for EVERY observation (no anchor): add the anchor observations back."""
    pb = synth.make_ba_problem(nkf, npt, seed)
    n = len(pb["anchor_kf"])
    pb = dict(pb)
    pb["obs_kf"] = np.concatenate([pb["anchor_kf"], pb["obs_kf"]]).astype(np.int32)
    pb["obs_pt"] = np.concatenate([np.arange(n, dtype=np.int32), pb["obs_pt"]]).astype(np.int32)
    pb["obs_uv"] = np.concatenate([pb["anchor_uv"], pb["obs_uv"]])
    return pb


@pytest.mark.parametrize("nkf,npt,seed", [(6, 150, 5), (12, 400, 6)])
def test_local_ba_xyz(nkf, npt, seed):
    pb = xyz_problem(nkf, npt, seed)
    a = Orc.local_ba(pb, 5, 0.0, inv_depth=False)
    b = Ref.local_ba(pb, 5, 0.0, inv_depth=False)
    ba_compare(a, b, pt_tol=1e-6)


def _img(w, h, seed, noise=True, k=2):
    return synth.frame_gray(synth.texture_canvas(w, h, seed), k, w, h, noise_seed=seed if noise else None)


@pytest.mark.parametrize("cell,x,y,seed", [(12, 24, 36, 1), (40, 0, 0, 2), (15, 600, 450, 3), (35, 70, 35, 4)])
def test_cell_mineig_bit_exact(cell, x, y, seed):
    g = _img(640, 480, seed)
    ob, oe = Orc.cell_mineig(g, x, y, cell)
    rb, re_ = Ref.cell_mineig(g, x, y, cell)
    assert np.array_equal(ob, rb)
    assert np.array_equal(oe.view(np.uint32), re_.view(np.uint32))


def test_corner_subpix_bit_exact():
    g = _img(640, 480, 5)
    rng = np.random.RandomState(0)
    pts = np.stack([rng.uniform(0, 640, 600), rng.uniform(0, 480, 600)], 1).astype(np.float32)
    pts[:100] = np.round(pts[:100])
    pts[100:110] = [[2, 2], [637, 477], [0, 0], [639, 479], [4, 100], [5, 100], [634, 100], [300, 3], [300, 475], [6.5, 6.5]]
    assert np.array_equal(Orc.corner_subpix(g, pts).view(np.uint32), Ref.corner_subpix(g, pts).view(np.uint32))


@pytest.mark.parametrize("w,h,cell,seed,nocc", [(640, 480, 12, 1, 0), (640, 480, 40, 2, 30), (1280, 720, 15, 3, 200), (640, 480, 35, 4, 0),
                                               (200, 160, 12, 5, 10)])
def test_detect_grid_bit_exact(w, h, cell, seed, nocc):
    g = _img(w, h, seed)
    rng = np.random.RandomState(seed)
    occ = np.stack([rng.uniform(0, w - 1, nocc), rng.uniform(0, h - 1, nocc)], 1).astype(np.float32)
    mq = 0.001
    for rep in range(3):  # the adaptive threshold carries over between calls
        op, omq = Orc.detect_grid(g, cell, occ, max_quality=mq)
        rp, rmq = Ref.detect_grid(g, cell, occ, max_quality=mq)
        assert omq == rmq
        assert op.shape == rp.shape and len(rp) > 0
        assert np.array_equal(op.view(np.uint32), rp.view(np.uint32))
        mq = rmq


@pytest.mark.parametrize("w,h,seed,thr", [(640, 480, 1, 20), (200, 120, 2, 10), (1280, 720, 3, 20), (64, 48, 4, 5)])
def test_fast_bit_exact(w, h, seed, thr):
    g = _img(w, h, seed)
    oxy, osc = Orc.fast(g, thr)
    rxy, rsc = Ref.fast(g, thr)
    assert len(rxy) > 0
    assert np.array_equal(oxy, rxy) and np.array_equal(osc, rsc)


@pytest.mark.parametrize("w,h,seed,scale,nlevels", [(640, 480, 1, 1.2, 8), (1280, 720, 3, 1.2, 8), (333, 201, 4, 1.2, 8), (640, 480, 7, 1.5, 4),
                                                   (320, 240, 9, 2.0, 4), (400, 300, 8, 1.1, 12)])
def test_orb_pyramid_bit_exact(w, h, seed, scale, nlevels):
    """cv::ORB's pyramid (sizes orb.cpp:1041-1058, resize(prev, cur, INTER_LINEAR_EXACT) :1086-1099): every byte of every level"""
    g = _img(w, h, seed, noise=True)
    ol, rl = Orc.orb_pyramid(g, scale, nlevels), Ref.orb_pyramid(g, scale, nlevels)
    assert len(ol) == len(rl) == nlevels
    for a, b in zip(ol, rl):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(ol[0], g)


def orb_key(kp):
    """canonical order: (octave, y, x) -- cv::ORB's own order within a level comes from std::nth_element"""
    return np.lexsort((kp[:, 0], kp[:, 1], kp[:, 5]))


@pytest.mark.parametrize("w,h,seed,nf", [(640, 480, 1, 2000), (1280, 720, 3, 4000), (320, 240, 5, 300)])
def test_orb_detect_and_compute_set_exact(w, h, seed, nf):
    g = _img(w, h, seed, noise=False)
    okp, od = Orc.orb(g, nf)
    rkp, rd = Ref.orb(g, nf)
    assert len(okp) == len(rkp) and len(rkp) > 0.3 * nf
    oi, ri = orb_key(okp), orb_key(rkp)
    assert np.array_equal(okp[oi].view(np.uint32), rkp[ri].view(np.uint32))   # x, y, size, angle, response, octave: bitwise
    assert np.array_equal(od[oi], rd[ri])


@pytest.mark.parametrize("w,h,seed,nf,scale,nlevels,thr", [(640, 480, 7, 500, 1.5, 4, 20), (400, 300, 8, 1000, 1.2, 3, 7), (320, 240, 9, 200, 2.0, 2, 40),
                                                           (256, 256, 10, 300, 1.2, 1, 20)])
def test_orb_other_pyramid_parameters(w, h, seed, nf, scale, nlevels, thr):
    g = _img(w, h, seed, noise=True)
    okp, od = Orc.orb(g, nf, scale=scale, nlevels=nlevels, fast_thr=thr)
    rkp, rd = Ref.orb(g, nf, scale=scale, nlevels=nlevels, fast_thr=thr)
    assert len(okp) == len(rkp) and len(rkp) > 10
    oi, ri = orb_key(okp), orb_key(rkp)
    assert np.array_equal(okp[oi].view(np.uint32), rkp[ri].view(np.uint32))
    assert (od[oi] != rd[ri]).any(axis=1).sum() <= max(1, len(okp) // 1000)


@pytest.mark.parametrize("thr", [1, 60, 120])
def test_fast_threshold_extremes(thr):
    g = _img(320, 240, 12)
    oxy, osc = Orc.fast(g, thr)
    rxy, rsc = Ref.fast(g, thr)
    assert np.array_equal(oxy, rxy) and np.array_equal(osc, rsc)


@pytest.mark.parametrize("maxq", [0.0, -1.0, 1e-9, 10.0])
def test_detect_grid_unusual_quality_thresholds(maxq):
    from alvaar_amd import synth
    w, h, cell = 320, 240, 12
    g = synth.frame_gray(synth.texture_canvas(w, h, 4), 2, w, h, noise_seed=4)
    g[60:120, 80:200] = 90
    occ = np.random.RandomState(5).uniform(20, [w - 20, h - 20], (40, 2)).astype(np.float32)
    op, oq = Orc.detect_grid(g, cell, occupied=occ, max_quality=maxq)
    rp, rq = Ref.detect_grid(g, cell, occupied=occ, max_quality=maxq)
    assert oq == rq and np.array_equal(op.view(np.uint32), rp.view(np.uint32))
