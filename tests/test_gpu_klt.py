"""GPU parity: a4 -- pyramidal LK and forward-backward KLT.  The HIP kernel replays the reference
build's float accumulation order, so positions are compared BITWISE and status flags exactly."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref, ref_available
from test_oracle_vs_ref import klt_case

pytestmark = pytest.mark.gpu


def _pyrs(ctx, prev, curr, built=3):
    import torch
    import alvaar_amd
    h, w = prev.shape
    pp = alvaar_amd.Pyramid(ctx, w, h, 9, built)
    cp = alvaar_amd.Pyramid(ctx, w, h, 9, built)
    pp.build_from_gray(torch.from_numpy(prev).cuda())
    cp.build_from_gray(torch.from_numpy(curr).cuda())
    return pp, cp


@pytest.mark.parametrize("w,h,n,levels,seed", [(640, 480, 600, 3, 1), (640, 480, 300, 1, 2), (640, 480, 300, 0, 3), (200, 152, 200, 3, 4)])
def test_lk_bitwise(ctx, w, h, n, levels, seed):
    import torch
    prev, curr, pts, init = klt_case(w, h, n, seed)
    pp, cp = _pyrs(ctx, prev, curr)
    nx, st, er = ctx.lk_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    nx, st, er = nx.cpu().numpy(), st.cpu().numpy(), er.cpu().numpy()
    for name, O in [("orc", Orc)] + ([("ref", Ref)] if ref_available() else []):
        on, os_, oe = O.lk(prev, curr, pts, init, levels)
        assert np.array_equal(st, os_), name
        assert np.array_equal(nx.view(np.uint32), on.view(np.uint32)), name
        ok = os_.astype(bool)
        assert np.array_equal(er[ok].view(np.uint32), oe[ok].view(np.uint32)), name


@pytest.mark.parametrize("w,h,n,levels,seed", [(640, 480, 2120, 3, 5), (640, 480, 400, 1, 6), (1280, 720, 4080, 3, 7)])
def test_fbklt_bitwise(ctx, w, h, n, levels, seed):
    import torch
    prev, curr, pts, init = klt_case(w, h, n, seed)
    pp, cp = _pyrs(ctx, prev, curr)
    pr, st = ctx.fbklt_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    pr, st = pr.cpu().numpy(), st.cpu().numpy()
    for name, O in [("orc", Orc)] + ([("ref", Ref)] if ref_available() else []):
        op, os_ = O.fbklt(prev, curr, pts, init, levels)
        assert np.array_equal(st, os_), name
        assert np.array_equal(pr.view(np.uint32), op.view(np.uint32)), name
    assert 0.2 * n < st.sum() < n


def test_fbklt_identity_property(ctx):
    """Size-independent property at the full BASELINE size: tracking a frame against itself keeps every
    interior textured point within the FB gate and moves it by < 0.01 px."""
    import torch
    w, h, n = 640, 480, 2120
    canvas = synth.texture_canvas(w, h, 7)
    g = synth.frame_gray(canvas, 0, w, h)
    pp, cp = _pyrs(ctx, g, g)
    rng = np.random.RandomState(0)
    pts = np.stack([rng.uniform(20, w - 20, n), rng.uniform(20, h - 20, n)], 1).astype(np.float32)
    pr, st = ctx.fbklt_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(pts).cuda(), 3)
    pr, st = pr.cpu().numpy(), st.cpu().numpy().astype(bool)
    assert st.sum() > 0.5 * n
    assert np.abs(pr[st] - pts[st]).max() < 0.01


@pytest.mark.parametrize("dx,dy,levels,seed", [(14, 9, 3, 21), (9, -6, 0, 22), (30, 18, 3, 23)])
def test_klt_large_motion_restages_the_search_tile(ctx, dx, dy, levels, seed):
    """Displacements far beyond the 3 px of travel the LDS tile of the searched image allows for: the window leaves the tile
    (several times at level 0 without pyramid help) and the tile is re-staged; results stay bitwise equal to the oracle, also
    with a poor initial guess that makes the iteration wander."""
    import torch
    w, h, n = 320, 240, 500
    canvas = synth.texture_canvas(w + 64, h + 64, seed)
    prev = canvas[32:32 + h, 32:32 + w].copy()
    curr = canvas[32 - dy:32 - dy + h, 32 - dx:32 - dx + w].copy()     # content moves by (+dx, +dy)
    rng = np.random.RandomState(seed)
    pts = rng.uniform(12, [w - 12, h - 12], (n, 2)).astype(np.float32)
    init = (pts + rng.uniform(-6, 6, pts.shape)).astype(np.float32)
    pp, cp = _pyrs(ctx, prev, curr)
    nx, st, er = ctx.lk_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    on, os_, oe = Orc.lk(prev, curr, pts, init, levels)
    assert np.array_equal(st.cpu().numpy(), os_)
    assert np.array_equal(nx.cpu().numpy().view(np.uint32), on.view(np.uint32))
    pr, fs = ctx.fbklt_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    op, ofs = Orc.fbklt(prev, curr, pts, init, levels)
    assert np.array_equal(fs.cpu().numpy(), ofs) and np.array_equal(pr.cpu().numpy().view(np.uint32), op.view(np.uint32))
    moved = np.abs(on[os_.astype(bool)] - pts[os_.astype(bool)]).max()
    assert moved > 4.0        # the windows really travelled


@pytest.mark.parametrize("lanes", [5, 8, 16, 32, 64])
def test_fbklt_batch_bitwise(ctx, lanes):
    """alva_fbklt_track_batch: three image pairs of two sizes in ONE launch, every wave layout, each pair bit-identical to the oracle
    (and the compiled reference where present).  The 2120-point case contains taps whose fourth bilinear weight rounds to -1."""
    import torch
    from alvaar_amd import capi
    cases = [klt_case(640, 480, 2120, 5), klt_case(640, 480, 401, 6), klt_case(1280, 720, 4080, 7)]
    pyrs = [_pyrs(ctx, c[0], c[1]) for c in cases]
    outs, sts = capi.fbklt_track_batch(ctx, [p[0] for p in pyrs], [p[1] for p in pyrs], [torch.from_numpy(c[2]).cuda() for c in cases],
                                       [torch.from_numpy(c[3]).cuda() for c in cases], 3, lanes)
    ctx.sync()
    for (prev, curr, pts, init), pr, st in zip(cases, outs, sts):
        pr, st = pr.cpu().numpy(), st.cpu().numpy()
        for name, O in [("orc", Orc)] + ([("ref", Ref)] if ref_available() else []):
            op, os_ = O.fbklt(prev, curr, pts, init, 3)
            assert np.array_equal(st, os_), name
            assert np.array_equal(pr.view(np.uint32), op.view(np.uint32)), name


@pytest.mark.parametrize("w,h,levels,seed", [(320, 240, 3, 31), (320, 240, 0, 32), (212, 158, 2, 33)])
def test_klt_border_windows_with_wandering_priors(ctx, w, h, levels, seed):
    """Everything the row layout's search-tile paths can meet at once: points on, near and beyond the image border (their 16 x 16 tile
    is read through the clamped path), priors up to 12 px off (the window leaves the tile and the level, re-stages near the border, runs
    out of bounds on some level) on content that moved by (7, -4).  Positions bitwise, status and error exact, for plain LK and fb-KLT."""
    import torch
    n = 5000
    canvas = synth.texture_canvas(w + 64, h + 64, seed)
    prev = canvas[32:32 + h, 32:32 + w].copy()
    curr = canvas[32 + 4:32 + 4 + h, 32 - 7:32 - 7 + w].copy()
    rng = np.random.RandomState(seed)
    pts = np.stack([rng.uniform(-6, w + 6, n), rng.uniform(-6, h + 6, n)], 1).astype(np.float32)
    pts[: n // 5] = np.round(pts[: n // 5])
    init = (pts + rng.uniform(-12, 12, pts.shape)).astype(np.float32)
    pp, cp = _pyrs(ctx, prev, curr)
    nx, st, er = ctx.lk_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    nx, st, er = nx.cpu().numpy(), st.cpu().numpy(), er.cpu().numpy()
    on, os_, oe = Orc.lk(prev, curr, pts, init, levels)
    assert np.array_equal(st, os_)
    assert np.array_equal(nx.view(np.uint32), on.view(np.uint32))
    ok = os_.astype(bool)
    assert np.array_equal(er[ok].view(np.uint32), oe[ok].view(np.uint32))
    pr, fs = ctx.fbklt_track(pp, cp, torch.from_numpy(pts).cuda(), torch.from_numpy(init).cuda(), levels)
    op, ofs = Orc.fbklt(prev, curr, pts, init, levels)
    assert np.array_equal(fs.cpu().numpy(), ofs) and np.array_equal(pr.cpu().numpy().view(np.uint32), op.view(np.uint32))
    assert 0.05 * n < ok.sum() < 0.98 * n and 0 < ofs.sum() < ok.sum()      # some of everything: tracked, lost, out of bounds
