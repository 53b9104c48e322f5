"""GPU parity: a2 (gray), a3 (LK pyramid), a7 (Hamming brute force) -- HIP path through the C ABI
vs the CPU oracle(s).  Integer stages: bit-exact."""
import numpy as np
import pytest

from alvaar_amd import synth
from oracles import Orc, Ref, ref_available

pytestmark = pytest.mark.gpu


def _checkers():
    return [("orc", Orc)] + ([("ref", Ref)] if ref_available() else [])


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (1280, 720, 2), (64, 48, 3), (4, 1, 4)])
def test_rgba2gray_bit_exact(ctx, w, h, seed):
    import torch
    rgba = synth.random_rgba(w, h, seed)
    out = ctx.rgba2gray(torch.from_numpy(rgba).cuda()).cpu().numpy()
    for name, O in _checkers():
        assert np.array_equal(out, O.rgba2gray(rgba)), name


@pytest.mark.parametrize("w,h,levels", [(640, 480, 3), (1280, 720, 3), (100, 76, 3), (52, 44, 3), (332, 201, 2)])
def test_pyramid_bit_exact(ctx, w, h, levels):
    import torch
    import alvaar_amd
    canvas = synth.texture_canvas(w, h, seed=w + h)
    g = synth.frame_gray(canvas, 3, w, h, noise_seed=11)
    pyr = alvaar_amd.Pyramid(ctx, w, h, 9, levels)
    pyr.build_from_gray(torch.from_numpy(g).cuda())
    for name, O in _checkers():
        og, od = O.build_pyramid(g, 9, levels)
        assert pyr.num_levels == len(og)
        for l in range(pyr.num_levels):
            hg, hd = pyr.download_level(l)
            assert np.array_equal(hg, og[l]), f"{name}: gray level {l}"
            assert np.array_equal(hd, od[l]), f"{name}: deriv level {l}"
    # rebuilding with another frame must not leave stale border pixels
    g2 = synth.frame_gray(canvas, 9, w, h, noise_seed=5)
    pyr.build_from_gray(torch.from_numpy(g2).cuda())
    og, od = Orc.build_pyramid(g2, 9, levels)
    for l in range(pyr.num_levels):
        hg, hd = pyr.download_level(l)
        assert np.array_equal(hg, og[l]) and np.array_equal(hd, od[l])
    pyr.close()


@pytest.mark.parametrize("w,h,levels,with_copy", [(640, 480, 3, True), (640, 480, 3, False), (1280, 720, 3, True), (100, 76, 3, True), (52, 44, 3, False),
                                                  (332, 201, 2, True), (64, 32, 1, True), (68, 33, 1, True), (60, 31, 1, False)])
def test_pyramid_from_rgba_fused(ctx, w, h, levels, with_copy):
    """alva_pyramid_build_from_rgba: gray + the whole pyramid from the RGBA frame (k_pyr_all: ONE launch, every role converts its own
    pixels from the frame) -- the un-padded gray copy, every padded level and every derivative level bit for bit, on sizes whose tiles are
    cut by the right / bottom edge (100 x 76, 52 x 44, 332 x 201, 68 x 33, 60 x 31), and again over a second frame (stale borders)."""
    import torch
    import alvaar_amd
    canvas = synth.texture_canvas(w, h, seed=w + 3 * h)
    pyr = alvaar_amd.Pyramid(ctx, w, h, 9, levels)
    for shift, seed in ((0, 5), (7, 6)):
        rgba = synth.gray_to_rgba(synth.frame_gray(canvas, shift, w, h, noise_seed=seed), seed=seed)
        gout = torch.zeros((h, w), dtype=torch.uint8, device="cuda") if with_copy else None
        pyr.build_from_rgba(torch.from_numpy(rgba).cuda(), gout)
        gray = Orc.rgba2gray(rgba)
        if with_copy:
            assert np.array_equal(gout.cpu().numpy(), gray)
        og, od = Orc.build_pyramid(gray, 9, levels)
        assert pyr.num_levels == len(og)
        for l in range(pyr.num_levels):
            hg, hd = pyr.download_level(l)
            assert np.array_equal(hg, og[l]), f"gray level {l}"
            assert np.array_equal(hd, od[l]), f"deriv level {l}"
    pyr.close()


def test_pyramid_one_launch_equals_two(ctx):
    """the same call with ALVA_PYRAMID_TWO_LAUNCHES=1 (k_level0 + k_pyr_rest) in a child process: identical bytes"""
    import os, subprocess, sys, hashlib
    code = ("import hashlib, numpy as np, torch, alvaar_amd\n"
            "from alvaar_amd import synth\n"
            "ctx = alvaar_amd.Context(0)\n"
            "hs = hashlib.sha256()\n"
            "for w, h in ((640, 480), (100, 76), (1280, 720)):\n"
            "    rgba = synth.gray_to_rgba(synth.frame_gray(synth.texture_canvas(w, h, seed=9), 2, w, h, noise_seed=4), seed=8)\n"
            "    pyr = alvaar_amd.Pyramid(ctx, w, h, 9, 3)\n"
            "    g = torch.zeros((h, w), dtype=torch.uint8, device='cuda')\n"
            "    pyr.build_from_rgba(torch.from_numpy(rgba).cuda(), g)\n"
            "    hs.update(g.cpu().numpy().tobytes())\n"
            "    for l in range(pyr.num_levels):\n"
            "        a, b = pyr.download_level(l)\n"
            "        hs.update(a.tobytes()); hs.update(b.tobytes())\n"
            "print('DIGEST', hs.hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for two in (False, True):
        env = dict(os.environ)
        env.pop("ALVA_PYRAMID_TWO_LAUNCHES", None)
        if two:
            env["ALVA_PYRAMID_TWO_LAUNCHES"] = "1"
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert out[0] == out[1]


@pytest.mark.parametrize("nq,nt,seed", [(1, 1, 0), (17, 33, 1), (300, 257, 2), (2120, 2120, 3), (64, 0, 4), (4000, 4000, 5), (4080, 4100, 6)])   # 4000 x 4000 = BASELINE configs[2]
def test_bf_match_bit_exact(ctx, nq, nt, seed):
    import torch
    rng = np.random.RandomState(seed)
    q = rng.randint(0, 256, (nq, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (nt, 32)).astype(np.uint8)
    if nt > 2:
        t[nt // 2] = t[0]  # exact ties: the lowest train index must win
        t[nt - 1] = t[1]
        q[0] = t[0]
        if nq > 5:
            q[5] = t[1]
    idx, dist = ctx.bf_match_hamming(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    if nt == 0:
        assert (idx == -1).all() and (dist == -1).all()
        return
    oi, od = Orc.bf_match(q, t)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    if ref_available() and nq * nt < 10_000_000:
        ri, rd = Ref.bf_match(q, t)
        assert np.array_equal(idx, ri) and np.array_equal(dist, rd)


def test_bf_match_full_size_properties(ctx):
    """BASELINE config 3 size (4080 x 4080): self-match is the identity with distance 0, and the
    result is invariant to how the train set is chunked (size-independent property)."""
    import torch
    rng = np.random.RandomState(9)
    d = rng.randint(0, 256, (4080, 32)).astype(np.uint8)
    dd = torch.from_numpy(d).cuda()
    idx, dist = ctx.bf_match_hamming(dd, dd)
    assert torch.equal(idx.cpu(), torch.arange(4080, dtype=torch.int32)) and int(dist.abs().sum()) == 0
    perm = rng.permutation(4080)
    idx2, dist2 = ctx.bf_match_hamming(dd, torch.from_numpy(d[perm]).cuda())
    assert np.array_equal(perm[idx2.cpu().numpy()], np.arange(4080)) and int(dist2.abs().sum()) == 0


@pytest.mark.parametrize("B", [5, 9])
def test_batched_pyramids_equal_single_builds(ctx, B):
    """alva_pyramid_build_from_rgba_batch == one alva_pyramid_build_from_rgba per camera, bit for bit (9 cameras: the camera -> XCD
    order of the batched launches, alva_xcd_item; 5: the plain order)."""
    import torch
    import alvaar_amd
    from alvaar_amd import capi
    w, h = 320, 240
    frames = [torch.from_numpy(synth.gray_to_rgba(synth.frame_gray(synth.texture_canvas(w, h, 20 + c), c, w, h, noise_seed=c), seed=c)).cuda()
              for c in range(B)]
    single = [alvaar_amd.Pyramid(ctx, w, h, 9, 3) for _ in range(B)]
    batch = [alvaar_amd.Pyramid(ctx, w, h, 9, 3) for _ in range(B)]
    g1 = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in range(B)]
    g2 = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in range(B)]
    for c in range(B):
        single[c].build_from_rgba(frames[c], g1[c])
    capi.build_pyramids_batch(ctx, batch, frames, g2)
    ctx.sync()
    for c in range(B):
        assert torch.equal(g1[c], g2[c])
        for l in range(single[c].num_levels):
            a, b = single[c].download_level(l), batch[c].download_level(l)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (c, l)
    capi.build_pyramids_batch(ctx, batch[:1], frames[:1])          # a batch of one, no gray output
    ctx.sync()
    assert np.array_equal(batch[0].download_level(2)[1], single[0].download_level(2)[1])
