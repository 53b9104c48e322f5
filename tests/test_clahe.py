"""f4a: CLAHE.  CPU: oracle restatement == OpenCV's CLAHE in the compiled reference, bit for bit, and ==
the committed golden output.  GPU: alva_clahe == oracle, bit for bit."""
import numpy as np
import pytest
from pathlib import Path

from alvaar_amd import synth
from oracles import orc_clahe, ref_clahe, ref_available

G = Path(__file__).resolve().parent / "golden"
CASES = [(640, 480, 3.0, (12, 9), 1), (640, 480, 40.0, (8, 8), 2), (320, 240, 2.0, (6, 4), 3), (100, 76, 3.0, (2, 1), 4),
         (1280, 720, 3.0, (25, 14), 5), (64, 48, 0.0, (4, 4), 6), (52, 44, 3.0, (1, 1), 7)]


def _img(w, h, seed):
    return synth.frame_gray(synth.texture_canvas(w, h, seed), 2, w, h, noise_seed=seed)


@pytest.mark.ref
@pytest.mark.parametrize("w,h,clip,tiles,seed", CASES)
def test_oracle_matches_opencv(w, h, clip, tiles, seed):
    if not ref_available():
        pytest.skip("compiled reference not present")
    g = _img(w, h, seed)
    assert np.array_equal(orc_clahe(g, clip, tiles), ref_clahe(g, clip, tiles))


def test_oracle_matches_golden():
    z = np.load(G / "clahe.npz")
    assert np.array_equal(orc_clahe(z["g"], float(z["clip"]), tuple(z["tiles"])), z["out"])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,clip,tiles,seed", CASES)
def test_hip_matches_oracle(ctx, w, h, clip, tiles, seed):
    import torch
    g = _img(w, h, seed)
    out = ctx.clahe(torch.from_numpy(g).cuda(), clip, tiles)
    assert np.array_equal(out.cpu().numpy(), orc_clahe(g, clip, tiles))


@pytest.mark.gpu
def test_hip_matches_golden_and_default_grid(ctx):
    import torch
    z = np.load(G / "clahe.npz")
    out = ctx.clahe(torch.from_numpy(z["g"]).cuda(), float(z["clip"]), tuple(z["tiles"]))
    assert np.array_equal(out.cpu().numpy(), z["out"])
    g = _img(640, 480, 9)
    a = ctx.clahe(torch.from_numpy(g).cuda())                       # VisualFrontend's grid: size / 50 = 12 x 9, clip 3
    assert np.array_equal(a.cpu().numpy(), orc_clahe(g, 3.0, (12, 9)))
    flat = torch.full((96, 128), 200, dtype=torch.uint8, device="cuda")
    assert np.array_equal(ctx.clahe(flat, 3.0, (4, 3)).cpu().numpy(), orc_clahe(flat.cpu().numpy(), 3.0, (4, 3)))
