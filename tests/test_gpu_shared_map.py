"""GPU: the shared-map merge APPLIED to sessions (north_star's optional extra; the reference has one map -- parity unpinned; semantics of the
per-session part from MapManager::mergeMapPoints, /root/reference/src/slam/src/map_manager.cpp:428-513).  Two alva::System sessions in one
process stand for two ranks: their record blocks are concatenated the way all_gather_into_tensor lays them out (the collective itself is
covered on RCCL by test_gpu_rccl_merge.py), fused on the GPU (alva_fuse_map_points) and the result applied with multi.apply_merge."""
import numpy as np
import pytest

from alvaar_amd import synth

pytestmark = pytest.mark.gpu
W, H = 640, 480


def _run(ar, frames, k0, k1):
    st = 3
    for k in range(k0, k1):
        st = ar.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
    return st


def _round(sessions, ctx):
    import torch
    from alvaar_amd import multi
    blocks = [multi.system_map_records(ar, s, 8192)[0] for s, ar in enumerate(sessions)]
    allrec = torch.cat(blocks, 0)
    stream, ids, keep, absorbed = multi.fuse_duplicates(allrec, ctx)
    rec = allrec.reshape(-1, multi.RECORD_BYTES)
    rec = rec[rec[:, 4:8].contiguous().view(torch.int32).reshape(-1) >= 0]
    key = rec[:, 0:4].contiguous().view(torch.int32).reshape(-1).to(torch.int64) * (1 << 32) + rec[:, 4:8].contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    xyz = rec[torch.argsort(key, stable=True)][:, 8:32].contiguous().view(torch.float64).reshape(-1, 3).cpu().numpy()
    args = (stream.cpu().numpy(), ids.cpu().numpy(), keep.cpu().numpy(), absorbed.cpu().numpy().astype(np.int64), xyz)
    return args, [multi.apply_merge(ar, s, *args) for s, ar in enumerate(sessions)]


def test_two_sessions_of_one_scene_share_ids_and_keep_tracking():
    import torch
    import alvaar_amd
    from alvaar_amd.system import AlvaAR
    canvas = synth.texture_canvas(W, H, 7)
    frames = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H)) for k in range(90)])).cuda()
    a, b, solo = (AlvaAR(W, H, cell_size=24, random_sampling=False) for _ in range(3))
    for ar in (a, b, solo):
        assert _run(ar, frames, 0, 60) == 1
    ctx = alvaar_amd.Context(0)
    (stream, ids, keep, absorbed, xyz), res = _round([a, b], ctx)
    n3d = int((stream == 0).sum())
    # the same scene from the same frames in the same world frame: every point of stream 1 IS a point of stream 0
    assert n3d > 100 and int((~keep.astype(bool)).sum()) == int((stream == 1).sum()) == n3d
    assert res[0]["registered"] and res[1]["registered"] and res[1]["registration"]["rms_m"] < 1e-9
    assert res[0]["applied"] == 0 and res[1]["applied"] == n3d and res[1]["local_merges"] == 0
    loc, sst, sid = b.shared_ids()
    assert len(loc) == n3d and (sst == 0).all() and np.array_equal(loc, sid)     # identical runs: the shared id is the twin's id
    assert len(a.shared_ids()[0]) == 0
    # sharing ids changes nothing a session tracks with: both go on exactly like the session that never merged
    for k in range(60, 90):
        sa, sb, ss = (ar.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k) for ar in (a, b, solo))
        assert sa == sb == ss == 1
        pa, pb, ps = a.pose7()[0], b.pose7()[0], solo.pose7()[0]
        assert np.array_equal(pa.view(np.uint64), ps.view(np.uint64)) and np.array_equal(pb.view(np.uint64), ps.view(np.uint64)), k
    # culled map points leave the table; the ones still there keep their shared ids
    loc2, sst2, sid2 = b.shared_ids()
    assert 0 < len(loc2) <= n3d and np.array_equal(loc2, sid2)
    for ar in (a, b, solo):
        ar.close()


def test_points_of_one_session_that_are_the_same_shared_point_are_merged():
    """two map points of session 1 absorbed by ONE point of session 0 (hand-made round result): the newer is merged into the older through
    MapManager::mergeMapPoints' path, the survivor carries the shared id, the session keeps tracking"""
    import torch
    from alvaar_amd import multi
    from alvaar_amd.system import AlvaAR
    canvas = synth.texture_canvas(W, H, 9)
    frames = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H)) for k in range(80)])).cuda()
    ar = AlvaAR(W, H, cell_size=24, random_sampling=False)
    assert _run(ar, frames, 0, 60) == 1
    ids, xyz, flags, inv, desc = ar.map_points(cap=262144)
    # a pair the reference's routine may merge: a 3-D point the current frame no longer sees and a newer one it does, never co-observed
    # by a keyframe (alva_system_merge_map_points refuses the others -- see include/alvaar_system.h); take the first pair that goes through
    lost = [int(i) for i in ids[(flags[:, 0] != 0) & (flags[:, 1] == 0)]]
    seen = [int(i) for i in ids[(flags[:, 0] != 0) & (flags[:, 1] != 0)]][::-1]
    both = [int(i) for i in ids[(flags[:, 0] != 0) & (flags[:, 1] != 0)]][:2]
    res = multi.apply_merge(ar, 1, np.array([0, 1, 1]), np.array([7, both[0], both[1]]), np.array([True, False, False]), np.array([0, 0, 0]))
    assert res["applied"] == 2 and res["local_merges"] == 0          # co-observed by the current frame: both keep their ids, same shared id
    older = newer = None
    before = ar.counters()["merges"]
    for p, q in zip(lost, seen):
        res = multi.apply_merge(ar, 1, np.array([0, 1, 1]), np.array([4242, p, q]), np.array([True, False, False]), np.array([0, 0, 0]))
        assert res["applied"] == 2
        if res["local_merges"] == 1:
            older, newer = min(p, q), max(p, q)
            break
    assert older is not None and ar.counters()["merges"] == before + 1
    loc, sst, sid = ar.shared_ids()
    assert older in loc and newer not in loc and sid[list(loc).index(older)] == 4242 and sst[list(loc).index(older)] == 0
    ids2 = ar.map_points(cap=262144)[0]
    assert older in ids2 and newer not in ids2
    assert _run(ar, frames, 60, 80) == 1
    ar.close()


def test_maps_in_different_frames_are_not_merged():
    """a second session with its own gauge (its map scaled by 1.2 and shifted: what an independent monocular initialisation gives): whatever
    pairs the 5 cm rule still finds, a similarity fitted to them is not the identity and nothing is applied"""
    from alvaar_amd import multi
    rng = np.random.RandomState(2)
    p = rng.uniform(-0.04, 0.04, (60, 3))                      # a small cloud: the 5 cm ball still pairs points of the scaled copy
    q = 1.2 * p + np.array([0.01, -0.005, 0.0])
    ok, reg = multi.frames_are_registered(q, p)
    assert not ok and abs(reg["scale"] - 1 / 1.2) < 1e-6

    class Dummy:
        def set_shared_ids(self, *a):
            raise AssertionError("nothing may be applied")
        merge_map_points = set_shared_ids
    n = len(p)
    stream = np.r_[np.zeros(n, int), np.ones(n, int)]
    ids = np.r_[np.arange(n), np.arange(n)]
    keep = np.r_[np.ones(n, bool), np.zeros(n, bool)]
    absorbed = np.r_[np.arange(n), np.arange(n)]
    res = multi.apply_merge(Dummy(), 1, stream, ids, keep, absorbed, np.vstack([p, q]))
    assert res == {"applied": 0, "local_merges": 0, "registered": False, "registration": reg}
    s, R, t, rms = multi.similarity_fit(p, q)
    assert abs(s - 1.2) < 1e-9 and rms < 1e-12


def _round_registered(sessions, ctx):
    import torch
    from alvaar_amd import multi
    blocks = [multi.system_map_records(ar, s, 8192)[0] for s, ar in enumerate(sessions)]
    allrec = torch.cat(blocks, 0)
    reg = multi.register_streams(allrec, ctx)
    moved = multi.apply_registration(allrec, reg)
    stream, ids, keep, absorbed = multi.fuse_duplicates(moved, ctx)
    _, _, xyz, _ = multi._split_records(moved)
    args = (stream.cpu().numpy(), ids.cpu().numpy(), keep.cpu().numpy(), absorbed.cpu().numpy().astype(np.int64), xyz.cpu().numpy())
    return reg, args, [multi.apply_merge(ar, s, *args) for s, ar in enumerate(sessions)]


def _centre(ar):
    return ar.pose7()[0][:3].copy()


@pytest.mark.parametrize("case", ["scaled_gauge", "late_start"])
def test_independent_monocular_maps_are_registered_then_merged(case):
    """Two sessions of one scene with DIFFERENT gauges.  scaled_gauge: session b's two-view translation is 1.3x session a's (test hook), so
    its whole map and trajectory are 1.3x larger.  late_start: b starts 8 frames later and initialises by itself -- its world origin is
    another camera position.  Without registration the one-world-frame check refuses the round (or nothing fuses); register_streams finds
    the similarity from descriptor correspondences alone, after which the maps fuse, b's points get a's ids as shared ids, and b's camera
    centre mapped through the similarity lands on a's."""
    import torch
    import alvaar_amd
    from alvaar_amd.system import AlvaAR
    canvas = synth.texture_canvas(W, H, 7)
    frames = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H)) for k in range(100)])).cuda()
    a, b = (AlvaAR(W, H, cell_size=24, random_sampling=False) for _ in range(2))
    off = 8 if case == "late_start" else 0
    prev = 3
    for k in range(90):
        sa = a.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
        if case == "scaled_gauge" and sa == 1 and prev != 1:
            p = a.pose7()[0].copy()
            p[:3] *= 1.3
            b.set_init_pose(p)          # b's two-view pose: a's, with a 1.3x longer baseline
        prev = sa
        if k >= off:
            sb = b.find_camera_pose_device(int(frames[k].data_ptr()), 33.0 * k)
    assert sa == 1 and sb == 1
    ctx = alvaar_amd.Context(0)
    (_, _, keep0, _, _), res0 = _round([a, b], ctx)
    assert res0[1]["applied"] == 0 and (not res0[1]["registered"] or int((~keep0.astype(bool)).sum()) < 4), "different gauges must not merge as they are"
    reg, (stream, ids, keep, absorbed, xyz), res = _round_registered([a, b], ctx)
    r = reg[1]
    assert r["inliers"] >= 30 and r["inliers"] > 0.5 * r["candidates"], r
    if case == "scaled_gauge":
        assert abs(r["scale"] - 1 / 1.3) < 0.02, r["scale"]
    else:
        assert abs(r["scale"] - 1) < 0.1 and np.linalg.norm(r["t"]) > 0.05, r
    fused = int((~keep.astype(bool)).sum())
    assert fused >= 30 and res[1]["registered"] and res[1]["applied"] == fused and res[0]["applied"] == 0
    loc, sst, sid = b.shared_ids()
    assert len(loc) == fused and (sst == 0).all()
    # the similarity maps b's camera onto a's: centres agree to a few percent of the distance travelled
    ca, cb = _centre(a), _centre(b)
    mapped = r["scale"] * (np.asarray(r["R"]) @ cb) + np.asarray(r["t"])
    assert np.linalg.norm(mapped - ca) < 0.05 * max(np.linalg.norm(ca), 1e-3), (mapped, ca)
    a.close()
    b.close()
