"""GPU: alva_track_batch_* (trackMono of B lock-step cameras, one launch per stage) gives every camera exactly what its own
alva_frontend_track / stage-seam calls give -- which the other test files pin to the oracle and the compiled reference."""
import numpy as np
import pytest

from alvaar_amd import synth

pytestmark = pytest.mark.gpu

W, H = 640, 480


def _camera(seed, n_pts, n_corr, frames=3, outlier_frac=0.15, w=W, h=H):
    import torch
    fr = torch.from_numpy(synth.stream_rgba(w, h, frames, seed=seed, noise=True)).cuda()
    rng = np.random.RandomState(seed)
    pts = torch.from_numpy(rng.uniform(40, [w - 40, h - 40], (n_pts, 2)).astype(np.float32)).cuda() if n_pts else None
    pb = synth.make_pnp_problem(max(n_corr, 8), seed + 100, outlier_frac=outlier_frac)
    bv, uv, wp = (torch.from_numpy(pb[k][:n_corr].copy()).cuda() for k in ("bv", "uv", "wpt"))
    return dict(frames=fr, pts=pts, bv=bv, uv=uv, wp=wp)


@pytest.mark.parametrize("lanes,many", [(5, False), (8, False), (16, False), (32, False), (64, False), (5, True), (8, True)])
def test_batch_equals_single_cameras(ctx, lanes, many):
    """ragged rig: different frames, keypoint counts and correspondence counts per camera, one camera without keypoints and one
    with too few correspondences for a pose.  many = 11 cameras: from 8 cameras on the batched launches use the camera -> XCD order
    (alva_xcd_item), below that the plain one"""
    import torch
    import alvaar_amd
    K = synth.make_pnp_problem(8, 1)["K"]
    spec = [(600, 600), (250, 333), (0, 90), (431, 3), (1000, 1200), (61, 4), (1, 50)]
    if many:
        spec += [(777, 40), (12, 900), (340, 340), (999, 5)]
    cams = [_camera(11 + i, a, b) for i, (a, b) in enumerate(spec)]
    B = len(cams)
    tb = alvaar_amd.TrackBatch(0, W, H, B, 1000, 1200)
    tb.set_klt_lanes(lanes)
    tb.bind([c["pts"] for c in cams], [c["bv"] for c in cams], [c["uv"] for c in cams], [c["wp"] for c in cams])
    fes = [alvaar_amd.Frontend(0, W, H, 1000, 300) for _ in range(B)]
    empty_f = torch.empty((0, 2), dtype=torch.float32, device="cuda")
    accepted = 0
    for k in range(3):
        st, poses = tb.step([c["frames"][k] for c in cams], K)
        st, poses = st.copy(), poses.copy()
        for i, c in enumerate(cams):
            pts = c["pts"] if c["pts"] is not None else empty_f
            st1, pose1, _ = fes[i].track(c["frames"][k], pts, c["bv"], c["uv"], c["wp"], K)
            assert st[i] == st1, (k, i)
            if st1 >= 1:
                assert np.array_equal(poses[i], pose1), (k, i)
            accepted += st1 == 2
            if k > 0 and spec[i][0] > 0:
                fes[i].sync()
                r = fes[i].results()
                tr, ok = tb.results(i)
                assert torch.equal(tr, r["tracked"]) and torch.equal(ok, r["status"]), (k, i)
                assert int(ok.sum()) > spec[i][0] // 4
    assert accepted >= 3 * 3
    with pytest.raises(alvaar_amd.AlvaError):
        tb.set_klt_lanes(7)
    for f in fes:
        f.close()
    tb.close()


def test_batch_of_one_and_many_identical_cameras(ctx):
    import alvaar_amd
    K = synth.make_pnp_problem(8, 1)["K"]
    c = _camera(5, 300, 300)
    one = alvaar_amd.TrackBatch(0, W, H, 1, 300, 300)
    many = alvaar_amd.TrackBatch(0, W, H, 33, 300, 300)
    one.bind([c["pts"]], [c["bv"]], [c["uv"]], [c["wp"]])
    many.bind([c["pts"]] * 33, [c["bv"]] * 33, [c["uv"]] * 33, [c["wp"]] * 33)
    for k in range(3):
        s1, p1 = one.step([c["frames"][k]], K)
        s2, p2 = many.step([c["frames"][k]] * 33, K)
        assert s1[0] == 2 and (s2 == 2).all()
        assert (p2 == p1[0]).all()
    import torch
    t1, o1 = one.results(0)
    for i in (0, 17, 32):
        t2, o2 = many.results(i)
        assert torch.equal(t1, t2) and torch.equal(o1, o2)
    one.close()
    many.close()


def test_degenerate_samples_fall_back_to_longer_prefix(ctx):
    """a camera whose correspondences are mostly one repeated point makes most P3P samples degenerate: the batch then hands that
    camera to the single-camera call, and the result is still the single-camera result"""
    import torch
    import alvaar_amd
    K = synth.make_pnp_problem(8, 1)["K"]
    c = _camera(21, 100, 60, outlier_frac=0.0)
    for x in ("bv", "uv", "wp"):
        c[x][6:] = c[x][5]   # 54 copies of one correspondence
    good = _camera(22, 100, 200)
    tb = alvaar_amd.TrackBatch(0, W, H, 2, 100, 200)
    tb.bind([c["pts"], good["pts"]], [c["bv"], good["bv"]], [c["uv"], good["uv"]], [c["wp"], good["wp"]])
    st, poses = tb.step([c["frames"][0], good["frames"][0]], K)
    st, poses = st.copy(), poses.copy()
    for i, cam in enumerate((c, good)):
        st1, pose1, _, _ = ctx.compute_pose(cam["bv"], cam["uv"], cam["wp"], K)
        assert st[i] == st1
        if st1 >= 1:
            assert np.array_equal(poses[i], pose1)
    assert st[1] == 2
    assert tb.stats() == (1, 1)   # exactly the degenerate camera went through the single-camera call
    tb.close()


def test_batch_rejects_bad_arguments():
    import alvaar_amd
    with pytest.raises(alvaar_amd.AlvaError):
        alvaar_amd.TrackBatch(0, W, H, 0, 100, 100)
    with pytest.raises(alvaar_amd.AlvaError):
        alvaar_amd.TrackBatch(0, W, H, 2, 100, 8000)
    tb = alvaar_amd.TrackBatch(0, W, H, 2, 50, 50)
    c = _camera(3, 60, 40)   # more keypoints than max_tracked
    tb.bind([c["pts"]] * 2, [c["bv"]] * 2, [c["uv"]] * 2, [c["wp"]] * 2)
    K = synth.make_pnp_problem(8, 1)["K"]
    with pytest.raises(alvaar_amd.AlvaError):
        tb.step([c["frames"][0]] * 2, K)
    tb.close()


@pytest.mark.parametrize("many", [False, True])
def test_batch_with_detector_equals_single_cameras(ctx, many):
    """the detector lane (cv::ORB + Hamming match per camera, batched over the cameras) gives every camera the keypoints, descriptors
    and matches of its own alva_frontend_track"""
    import torch
    import alvaar_amd
    K = synth.make_pnp_problem(8, 1)["K"]
    spec = [(300, 300), (120, 90), (0, 40), (500, 700), (64, 8)]
    if many:
        spec += [(222, 100), (499, 31), (17, 650), (400, 400)]
    cams = [_camera(41 + i, a, b, frames=4) for i, (a, b) in enumerate(spec)]
    B = len(cams)
    tb = alvaar_amd.TrackBatch(0, W, H, B, 500, 700)
    tb.enable_detector(500)
    tb.bind([c["pts"] for c in cams], [c["bv"] for c in cams], [c["uv"] for c in cams], [c["wp"] for c in cams])
    fes = [alvaar_amd.Frontend(0, W, H, 500, 500) for _ in range(B)]
    empty_f = torch.empty((0, 2), dtype=torch.float32, device="cuda")
    for k in range(4):
        st, poses = tb.step([c["frames"][k] for c in cams], K)
        st, poses, nkp = st.copy(), poses.copy(), tb.nkp.copy()
        for i, c in enumerate(cams):
            pts = c["pts"] if c["pts"] is not None else empty_f
            st1, pose1, nkp1 = fes[i].track(c["frames"][k], pts, c["bv"], c["uv"], c["wp"], K)
            assert st[i] == st1 and nkp[i] == nkp1 and nkp1 > 50, (k, i)
            if st1 >= 1:
                assert np.array_equal(poses[i], pose1), (k, i)
            fes[i].sync()
            r = fes[i].results()
            d = tb.detections(i)
            assert torch.equal(d["keypoints"], r["keypoints"]) and torch.equal(d["descriptors"], r["descriptors"]), (k, i)
            if k > 0:
                assert torch.equal(d["match_idx"], r["match_idx"]) and torch.equal(d["match_dist"], r["match_dist"]), (k, i)
                assert int((d["match_dist"] >= 0).sum()) == nkp1
    for f in fes:
        f.close()
    tb.close()


def test_batch_720p_with_detector(ctx):
    """BASELINE configs[2] geometry (1280x720): two cameras, detector lane on, against their own alva_frontend_track"""
    import torch
    import alvaar_amd
    w, h = 1280, 720
    K = synth.make_pnp_problem(8, 1)["K"]
    cams = [_camera(61 + i, 700, 500, frames=3, w=w, h=h) for i in range(2)]
    tb = alvaar_amd.TrackBatch(0, w, h, 2, 700, 500)
    tb.enable_detector(1000)
    tb.bind([c["pts"] for c in cams], [c["bv"] for c in cams], [c["uv"] for c in cams], [c["wp"] for c in cams])
    fes = [alvaar_amd.Frontend(0, w, h, 700, 1000) for _ in range(2)]
    for k in range(3):
        st, poses = tb.step([c["frames"][k] for c in cams], K)
        st, poses, nkp = st.copy(), poses.copy(), tb.nkp.copy()
        for i, c in enumerate(cams):
            st1, pose1, nkp1 = fes[i].track(c["frames"][k], c["pts"], c["bv"], c["uv"], c["wp"], K)
            assert st[i] == st1 == 2 and nkp[i] == nkp1 and np.array_equal(poses[i], pose1), (k, i)
            fes[i].sync()
            r, d = fes[i].results(), tb.detections(i)
            assert torch.equal(d["keypoints"], r["keypoints"]) and torch.equal(d["descriptors"], r["descriptors"]), (k, i)
            if k > 0:
                tr, ok = tb.results(i)
                assert torch.equal(tr, r["tracked"]) and torch.equal(ok, r["status"]), (k, i)
                assert torch.equal(d["match_idx"], r["match_idx"]) and torch.equal(d["match_dist"], r["match_dist"]), (k, i)
    for f in fes:
        f.close()
    tb.close()
