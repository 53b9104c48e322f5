"""GPU parity: a6 -- ORB 7x7 blur + 256-bit steered BRIEF of supplied points (bit-exact)."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref, ref_available
from test_oracle_vs_ref import _test_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (200, 120, 2), (1280, 720, 3), (70, 66, 4)])
def test_orb_blur_bit_exact(ctx, w, h, seed):
    import torch
    g = synth.frame_gray(synth.texture_canvas(w, h, seed), 2, w, h, noise_seed=seed)
    out = ctx.orb_blur(torch.from_numpy(g).cuda()).cpu().numpy()
    assert np.array_equal(out, Orc.orb_blur(g))
    if ref_available():
        assert np.array_equal(out, Ref.orb_blur(g))


@pytest.mark.parametrize("w,h,n,seed", [(640, 480, 2120, 1), (200, 120, 100, 2), (1280, 720, 4080, 3), (640, 480, 1, 4)])
def test_describe_bit_exact(ctx, w, h, n, seed):
    import torch
    g = synth.frame_gray(synth.texture_canvas(w, h, seed), 2, w, h, noise_seed=seed)
    pts = _test_points(w, h, max(n, 16), seed)[:n] if n >= 16 else np.array([[100.3, 99.7]], np.float32)
    desc, valid = ctx.describe(torch.from_numpy(g).cuda(), torch.from_numpy(pts).cuda())
    desc, valid = desc.cpu().numpy(), valid.cpu().numpy()
    od, ov = Orc.describe(g, pts)
    assert np.array_equal(valid, ov) and np.array_equal(desc, od)
    if ref_available():
        rd, rv = Ref.describe(g, pts)
        assert np.array_equal(valid, rv) and np.array_equal(desc, rd)
