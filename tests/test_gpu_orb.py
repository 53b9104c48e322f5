"""GPU parity: a5' -- FAST-9/16 + NMS (bit-exact, row-major order) and the full ORB detector.
ORB: the keypoint SET {x, y, size, angle, response, octave} must equal the reference's bitwise (cv::ORB's order within a
level is std::nth_element's; ours is row-major, so both sides are put in (octave, y, x) order first).  Descriptors:
identical bytes; the only non-reproducible step is cos/sin of the angle in double on the device vs glibc, so up to
0.1 % of descriptors may differ in a bit -- the test states and checks that bound (observed: 0)."""
import numpy as np
import pytest

from oracles import Orc, Ref, ref_available
from test_oracle_vs_ref import _img, orb_key

pytestmark = pytest.mark.gpu


def _ambiguous():
    """keypoints whose (float) cos / sin rotation was not provably the host library's (alva_orb_ambiguous_rotations), reset on read"""
    import ctypes as C
    from alvaar_amd.capi import lib, check
    n = C.c_int(0)
    check(lib.alva_orb_ambiguous_rotations(C.byref(n), 1))
    return n.value


@pytest.mark.parametrize("w,h,seed,thr", [(640, 480, 1, 20), (200, 120, 2, 10), (1280, 720, 3, 20), (64, 48, 4, 5)])
def test_fast_bit_exact(ctx, w, h, seed, thr):
    import torch
    g = _img(w, h, seed)
    xy, sc = ctx.fast(torch.from_numpy(g).cuda(), thr)
    O = Ref if ref_available() else Orc
    rxy, rsc = O.fast(g, thr)
    assert len(rxy) > 0
    assert np.array_equal(xy.cpu().numpy(), rxy) and np.array_equal(sc.cpu().numpy(), rsc)


@pytest.mark.parametrize("w,h,seed,nf", [(640, 480, 1, 2000), (1280, 720, 3, 4000), (320, 240, 5, 300)])
def test_orb_detect_and_compute(ctx, w, h, seed, nf):
    import torch
    import alvaar_amd
    g = _img(w, h, seed, noise=False)
    orb = alvaar_amd.Orb(ctx, w, h, nf)
    for rep in range(2):  # second call reuses every buffer
        kp, desc = orb.detect_and_compute(torch.from_numpy(g).cuda())
    kp, desc = kp.cpu().numpy(), desc.cpu().numpy()
    O = Ref if ref_available() else Orc
    rkp, rd = O.orb(g, nf)
    assert len(kp) == len(rkp) and len(rkp) > 0.3 * nf
    ri = orb_key(rkp)
    assert np.array_equal(kp.view(np.uint32), rkp[ri].view(np.uint32))   # ours is already in (octave, y, x) order
    bad = (desc != rd[ri]).any(axis=1).sum()
    assert bad == 0, bad                                     # descriptors bit-exact (north_star: Hamming scores bit-exact)
    assert _ambiguous() == 0                                 # ... and provably so: no rotation near a float rounding boundary
    orb.close()


@pytest.mark.parametrize("w,h,seed,nf,scale,nlevels,thr", [(640, 480, 7, 500, 1.5, 4, 20), (400, 300, 8, 1000, 1.2, 3, 7), (320, 240, 9, 200, 2.0, 2, 40),
                                                           (256, 256, 10, 300, 1.2, 1, 20), (640, 480, 11, 500, 2.0, 5, 20)])
def test_orb_other_pyramid_parameters(ctx, w, h, seed, nf, scale, nlevels, thr):
    """cv::ORB::create with other scale factors / level counts / FAST thresholds than the reference's defaults"""
    import torch
    import alvaar_amd
    g = _img(w, h, seed, noise=True)
    orb = alvaar_amd.Orb(ctx, w, h, nf, scale=scale, nlevels=nlevels, fast_threshold=thr)
    kp, desc = orb.detect_and_compute(torch.from_numpy(g).cuda())
    kp, desc = kp.cpu().numpy(), desc.cpu().numpy()
    rkp, rd = Orc.orb(g, nf, scale=scale, nlevels=nlevels, fast_thr=thr)
    assert len(kp) == len(rkp) and len(rkp) > 10
    ri = orb_key(rkp)
    assert np.array_equal(kp.view(np.uint32), rkp[ri].view(np.uint32))
    assert (desc != rd[ri]).any(axis=1).sum() == 0
    assert _ambiguous() == 0
    orb.close()


@pytest.mark.parametrize("w,h,seed,scale,nlevels", [(640, 480, 1, 1.2, 8), (1280, 720, 3, 1.2, 8), (333, 201, 4, 1.2, 8), (640, 480, 7, 1.5, 4),
                                                   (320, 240, 9, 2.0, 4), (400, 300, 8, 1.1, 12), (96, 64, 5, 1.2, 8), (640, 480, 11, 2.0, 5), (800, 600, 12, 1.7, 6)])
def test_orb_pyramid_and_blur_levels(ctx, w, h, seed, scale, nlevels, monkeypatch):
    """Every byte of every pyramid level (cv::ORB's INTER_LINEAR_EXACT chain, orb.cpp:1086-1099) and of its 7x7 blur (:1188) against the
    oracle (pinned to cv::resize / cv::GaussianBlur in test_oracle_vs_ref.py) -- from the fused launch, in which every tile recomputes
    its ancestors from level 0 (steep pyramids leave their deep levels to the per-level launch: scale 2.0 x 5 levels and 1.7 x 6 do), and
    from ALVA_ORB_PYRAMID=chain."""
    import torch
    import alvaar_amd
    g = _img(w, h, seed, noise=True)
    want = Orc.orb_pyramid(g, scale, nlevels)
    for mode in ("fused", "chain"):
        if mode == "chain":
            monkeypatch.setenv("ALVA_ORB_PYRAMID", "chain")
        orb = alvaar_amd.Orb(ctx, w, h, 300, scale=scale, nlevels=nlevels)
        for rep in range(2):
            orb.detect_and_compute(torch.from_numpy(g).cuda())
        for l, lv in enumerate(want):
            got = orb.level(l).cpu().numpy()
            assert got.shape == lv.shape and np.array_equal(got, lv), (mode, l, int((got != lv).sum()))
            assert np.array_equal(orb.level(l, blurred=True).cpu().numpy(), Orc.orb_blur(lv)), (mode, "blur", l)
        orb.close()


def test_orb_many_features_takes_the_radix_cull(ctx):
    """more than 4096 candidates at a level after the FAST cull: the Harris cull's selection passes instead of its LDS ranking"""
    import torch
    import alvaar_amd
    w, h, nf = 1280, 720, 14000
    g = _img(w, h, 3, noise=True)   # 14 k FAST corners at level 0: 6 080 + ties enter its Harris cull
    orb = alvaar_amd.Orb(ctx, w, h, nf)
    kp, desc = orb.detect_and_compute(torch.from_numpy(g).cuda())
    kp, desc = kp.cpu().numpy(), desc.cpu().numpy()
    rkp, rd = Orc.orb(g, nf, cap=4 * nf)
    assert len(kp) == len(rkp) and (rkp[:, 5] == 0).sum() > 2500
    ri = orb_key(rkp)
    assert np.array_equal(kp.view(np.uint32), rkp[ri].view(np.uint32))
    assert (desc != rd[ri]).any(axis=1).sum() == 0
    orb.close()


@pytest.mark.parametrize("thr", [1, 60, 120])
def test_fast_threshold_extremes(ctx, thr):
    import torch
    g = _img(320, 240, 12)
    xy, sc = ctx.fast(torch.from_numpy(g).cuda(), thr)
    rxy, rsc = Orc.fast(g, thr)
    assert np.array_equal(xy.cpu().numpy(), rxy) and np.array_equal(sc.cpu().numpy(), rsc)
