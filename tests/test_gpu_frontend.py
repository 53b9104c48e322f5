"""GPU: the native per-frame driver (alva_frontend_*) produces exactly what the stage-seam calls produce when they are
issued one by one (which the other test files pin to the oracle / reference)."""
import numpy as np
import pytest

from alvaar_amd import synth

pytestmark = pytest.mark.gpu

W, H, N = 640, 480, 600


def test_frontend_equals_stage_calls(ctx):
    import torch
    import alvaar_amd
    frames = torch.from_numpy(synth.stream_rgba(W, H, 3, seed=5, noise=True)).cuda()
    rng = np.random.RandomState(1)
    pts = torch.from_numpy(rng.uniform(40, [W - 40, H - 40], (N, 2)).astype(np.float32)).cuda()
    pb = synth.make_pnp_problem(N, 3, outlier_frac=0.15)
    bv, uv, wp = (torch.from_numpy(pb[k]).cuda() for k in ("bv", "uv", "wpt"))
    K = pb["K"]

    fe = alvaar_amd.Frontend(0, W, H, N, 500)
    orb = alvaar_amd.Orb(ctx, W, H, 500)
    pyr = [alvaar_amd.Pyramid(ctx, W, H, 9, 3) for _ in range(2)]
    gray = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    prev_desc = None
    for k in range(3):
        st, pose, nkp = fe.track(frames[k], pts, bv, uv, wp, K)
        pose = pose.copy()
        fe.sync()
        res = {a: b.clone() for a, b in fe.results().items()}
        # the same frame through the stage seam
        pyr[k % 2].build_from_rgba(frames[k], gray)
        st2, pose2, m1, m2 = ctx.compute_pose(bv, uv, wp, K)
        kp2, desc2 = orb.detect_and_compute(gray)
        assert st == st2 and np.array_equal(pose, pose2)
        assert nkp == kp2.shape[0] and nkp > 50
        assert torch.equal(res["keypoints"], kp2) and torch.equal(res["descriptors"], desc2)
        if k > 0:
            tr2, ok2 = ctx.fbklt_track(pyr[(k - 1) % 2], pyr[k % 2], pts, pts, 3)
            assert torch.equal(res["tracked"], tr2) and torch.equal(res["status"], ok2)
            assert int(ok2.sum()) > N // 4
            idx2, dist2 = ctx.bf_match_hamming(desc2, prev_desc)
            assert torch.equal(res["match_idx"], idx2) and torch.equal(res["match_dist"], dist2)
        prev_desc = desc2.clone()
    fe.close()


def test_frontend_rejects_bad_arguments():
    import alvaar_amd
    with pytest.raises(alvaar_amd.AlvaError):
        alvaar_amd.Frontend(0, 30, 30, 10, 100)


def test_lookahead_gives_identical_results():
    """alva_frontend_track_ahead (next frame preprocessed on a third stream) == alva_frontend_track, frame by frame, also when
    the announced next frame is NOT the one that arrives (the driver then rebuilds) and when look-ahead is switched on and off."""
    import torch
    import alvaar_amd
    R = 6
    frames = torch.from_numpy(synth.stream_rgba(W, H, R, seed=9, noise=True)).cuda()
    rng = np.random.RandomState(2)
    pts = torch.from_numpy(rng.uniform(40, [W - 40, H - 40], (N, 2)).astype(np.float32)).cuda()
    pb = synth.make_pnp_problem(N, 4, outlier_frac=0.15)
    bv, uv, wp = (torch.from_numpy(pb[k]).cuda() for k in ("bv", "uv", "wpt"))
    K = pb["K"]
    plain, ahead = alvaar_amd.Frontend(0, W, H, N, 500), alvaar_amd.Frontend(0, W, H, N, 500)
    order = [0, 1, 2, 3, 4, 5, 1, 3, 0, 2, 4]
    #          announce the true successor ... except at steps 4 and 7 (wrong frame) and 8 (none)
    announce = {k: order[k + 1] for k in range(len(order) - 1)}
    announce[4] = 0
    announce[7] = 5
    announce[8] = None
    for k, f in enumerate(order):
        a = plain.track(frames[f], pts, bv, uv, wp, K)
        a = (a[0], a[1].copy(), a[2])
        nxt = announce.get(k)
        b = ahead.track(frames[f], pts, bv, uv, wp, K, rgba_next=None if nxt is None else frames[nxt])
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2], k
        plain.sync()
        ahead.sync()
        ra, rb = plain.results(), ahead.results()
        for key in ("keypoints", "descriptors") + (("tracked", "status", "match_idx", "match_dist") if k > 0 else ()):
            assert torch.equal(ra[key], rb[key]), (k, key)
    plain.close()
    ahead.close()
