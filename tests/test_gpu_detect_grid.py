"""GPU parity: a5 -- the reference's grid Shi-Tomasi detector (FeatureExtractor::detectFeaturePoints), bit-exact:
the detected point list (order included), the sub-pixel positions (float bits) and the adaptive threshold."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref, ref_available
from test_oracle_vs_ref import _img

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,cell,seed,nocc", [(640, 480, 12, 1, 0), (640, 480, 40, 2, 30), (1280, 720, 15, 3, 200), (640, 480, 35, 4, 0),
                                               (200, 160, 12, 5, 10), (640, 480, 13, 6, 5)])
def test_detect_grid_bit_exact(ctx, w, h, cell, seed, nocc):
    import torch
    g = _img(w, h, seed)
    rng = np.random.RandomState(seed)
    occ = np.stack([rng.uniform(0, w - 1, nocc), rng.uniform(0, h - 1, nocc)], 1).astype(np.float32)
    gd = torch.from_numpy(g).cuda()
    od = torch.from_numpy(occ).cuda() if nocc else None
    mq = 0.001
    O = Ref if ref_available() else Orc
    for rep in range(3):
        pts, nmq = ctx.detect_grid(gd, cell, od, max_quality=mq)
        rp, rmq = O.detect_grid(g, cell, occ, max_quality=mq)
        pts = pts.cpu().numpy()
        assert nmq == rmq
        assert pts.shape == rp.shape and len(rp) > 0
        assert np.array_equal(pts.view(np.uint32), rp.view(np.uint32))
        mq = rmq


def test_detect_grid_properties_full_size(ctx):
    """BASELINE sizes: 2120 / 4080 cells; at most one primary + one secondary per cell, all inside the roi before
    refinement (|refined - integer| <= 3 px), and the call is idempotent for a fixed threshold."""
    import torch
    for (w, h, cell) in [(640, 480, 12), (1280, 720, 15)]:
        g = torch.from_numpy(_img(w, h, 7, noise=False, k=0)).cuda()
        a, q1 = ctx.detect_grid(g, cell, max_quality=0.001)
        b, q2 = ctx.detect_grid(g, cell, max_quality=0.001)
        assert torch.equal(a, b) and q1 == q2
        n = a.shape[0]
        assert 0.5 * (w // cell) * (h // cell) < n <= (w // cell) * (h // cell)
        a = a.cpu().numpy()
        assert (a[:, 0] >= 20 - 3).all() and (a[:, 0] < w - 20 + 3).all() and (a[:, 1] >= 20 - 3).all() and (a[:, 1] < h - 20 + 3).all()


@pytest.mark.parametrize("rounds", ["1", "2"])
def test_detect_grid_finisher_path(ctx, monkeypatch, rounds):
    """With fewer multi-CU repair rounds than the fixed point needs, the single-workgroup finisher (k_select) has real work
    left; the result must not change."""
    import torch
    w, h, cell = 640, 480, 12
    g = synth.frame_gray(synth.texture_canvas(w, h, 1), 2, w, h, noise_seed=1)
    occ = np.random.RandomState(3).uniform(20, [w - 20, h - 20], (150, 2)).astype(np.float32)
    monkeypatch.setenv("ALVA_GRID_ROUNDS", rounds)
    pts, q = ctx.detect_grid(torch.from_numpy(g).cuda(), cell, occupied=torch.from_numpy(occ).cuda())
    monkeypatch.delenv("ALVA_GRID_ROUNDS")
    ref_pts, ref_q = Orc.detect_grid(g, cell, occupied=occ)
    assert q == ref_q and np.array_equal(pts.cpu().numpy().view(np.uint32), ref_pts.view(np.uint32))


@pytest.mark.parametrize("maxq", [0.0, -1.0, 1e-9, 10.0])
def test_detect_grid_unusual_quality_thresholds(ctx, maxq):
    """maxQuality <= 0 cannot occur in the reference (it starts at 0.001 and only halves) but the C ABI accepts it: the exact
    masked arg-max path (non-positive values compete) has to agree with the oracle as well; a huge threshold rejects everything."""
    import torch
    w, h, cell = 320, 240, 12
    g = synth.frame_gray(synth.texture_canvas(w, h, 4), 2, w, h, noise_seed=4)
    g[60:120, 80:200] = 90       # a flat patch: lambda_min exactly 0 there
    occ = np.random.RandomState(5).uniform(20, [w - 20, h - 20], (40, 2)).astype(np.float32)
    pts, q = ctx.detect_grid(torch.from_numpy(g).cuda(), cell, occupied=torch.from_numpy(occ).cuda(), max_quality=maxq)
    ref_pts, ref_q = Orc.detect_grid(g, cell, occupied=occ, max_quality=maxq)
    assert q == ref_q and np.array_equal(pts.cpu().numpy().view(np.uint32), ref_pts.view(np.uint32))
