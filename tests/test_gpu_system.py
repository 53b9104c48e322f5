"""GPU: the drop-in surface (a1 / a10 / a13 / f1) against the REFERENCE ITSELF.

alva_system_* (host-side map layer over the HIP stages) and the reference's own System (oracle/_ref: System -> VisualFrontend ->
MapManager -> Mapper -> Optimizer over the vendored OpenCV / OpenGV / Ceres; determinism switches of SURVEY.md §8c: explicit
timestamps, fixed-seed sampling, Ceres' wall-clock caps frozen) are fed the same frames.  Asserted per frame: identical status,
identical counters (keypoints 2-D / 3-D, occupied cells, keyframes, map points, ids handed out, p3pReq_, poseFailedCounter_), identical
keypoint ids IN CONTAINER ORDER with identical 3-D / descriptor flags, identical keyframe ids, identical map point tables
(3-D flag, observed flag, observer count, anchor, descriptor count) and identical descriptor medoids; per keyframe the keypoint set,
covisibility and local-map size.  Floats: keypoint pixels <= 1e-2 px (the tracker itself is bitwise for equal priors; priors come from
poses that agree to ~1e-10), poses: RMSE <= 1e-5 (BASELINE.json north_star), map points <= 1e-5.

The two-view initialisation is OpenGV's forward-difference refinement working at its rounding-noise floor (one ulp
on one input bearing moves the reference's own result by up to 1e-4), so the pose it returns cannot be reproduced to 1e-5 by ANY other
build of the same algorithm.  The tests therefore (a) compare the initialisation pose at 5e-3 and (b) start both maps from the
reference's two-view pose (alva_system_debug_set_init_pose) for the 1e-5 comparison of everything that follows; a run WITHOUT the
hook is compared as well (same discrete trajectory over 160 frames, raw poses to 1e-3, after a Sim(3) alignment to 1e-4, pixels to 0.1)."""
import numpy as np
import pytest

from alvaar_amd import synth
import sysdiff

pytestmark = [pytest.mark.gpu, pytest.mark.ref]
# the unhooked run after Sim(3) alignment of the trajectories (gauge rotation from the orientations, scale + translation from the camera
# centres): measured on MI355X 3.6e-5 of the trajectory's extent (centres) and 1.1e-5 rad (rotations), scale 1.0000064 -- most of the raw
# difference (RMSE 3.7e-5, worst 1.6e-4) IS gauge; what is left is the two-view pose's noise floor
ALIGNED_TOL = 1e-4


def _reference_run(frames, w, h, cell, reset_at=(), **kw):
    ref = sysdiff.RefSystem(w, h, cell, **kw)
    out, init_pose = [], None
    for k, rgba in enumerate(frames):
        if k in reset_at:
            ref.reset()
        st, p7, p16 = ref.step(rgba, 33.0 * k)
        if init_pose is None and st == 1:
            init_pose = p7.copy()
        out.append(dict(status=st, pose7=p7.copy(), pose16=p16.copy(), state=ref.state().copy(), kps=tuple(a.copy() for a in ref.frame_keypoints()),
                        kfs=ref.keyframe_ids().copy(), mps=tuple(a.copy() for a in ref.map_points())))   # copies: the slices would pin the capacity-sized arrays
    return ref, out, init_pose


def _differential(frames, w, h, cell, inject, pose_tol, min_kf, min_ba, px_tol=1e-2, reset_at=(), aligned_tol=None, min_kf_created=0, reference=None, **kw):
    """reference: (records, init_pose, final) of tests/ref_runner.py (ONE reproducible run of the reference in a process of its own: the long
    streams); default: the reference runs here, in this process"""
    frames = list(frames)
    traj_ref, traj_gpu = [], []
    if reference is None:
        ref, rec, init_pose = _reference_run(frames, w, h, cell, reset_at=reset_at, **kw)
    else:
        rec, init_pose, ref = reference
    gpu = sysdiff.GpuSystem(w, h, cell, **kw)
    try:
        sq, cnt, worst_px, worst_x, worst_pose = 0.0, 0, 0.0, 0.0, 0.0
        for k, rgba in enumerate(frames):
            r = rec[k]
            if k in reset_at:   # the caller's System::reset between two frames (system.cpp:42-55)
                gpu.reset()
            if inject and r["status"] == 1 and (k == 0 or rec[k - 1]["status"] != 1):
                gpu.set_init_pose(r["pose7"])   # the frame on which the reference (re-)initialises its map: start ours from the same two-view pose
            st, p7, p16 = gpu.step(rgba, 33.0 * k)
            assert st == r["status"], f"frame {k}: status {st} != reference {r['status']}"
            assert list(gpu.state()) == list(r["state"]), f"frame {k}: state {list(gpu.state())} != reference {list(r['state'])}"
            ids, px, un, i3, hd = gpu.frame_keypoints()
            rids, rpx, run, ri3, rhd = r["kps"]
            assert np.array_equal(ids, rids), f"frame {k}: keypoint ids / container order differ"
            assert np.array_equal(i3, ri3) and np.array_equal(hd, rhd), f"frame {k}: keypoint flags differ"
            if len(ids):
                worst_px = max(worst_px, float(np.abs(px - rpx).max()), float(np.abs(un - run).max()))
            assert np.array_equal(gpu.keyframe_ids(), r["kfs"]), f"frame {k}: keyframe ids differ"
            mi, mx, mf, minv, md = gpu.map_points()
            ri, rx, rf, rinv, rd = r["mps"]
            assert np.array_equal(mi, ri) and np.array_equal(mf, rf), f"frame {k}: map point table differs"
            assert np.array_equal(md, rd), f"frame {k}: descriptor medoids differ"
            if len(mi):
                worst_x = max(worst_x, float(np.abs(mx - rx).max()))
            if st == 1:
                traj_ref.append(r["pose7"].copy())
                traj_gpu.append(p7.copy())
                d = sysdiff.pose_diff(r["pose7"], p7)
                worst_pose = max(worst_pose, d)
                q = p7[3:] if np.dot(p7[3:], r["pose7"][3:]) >= 0 else -p7[3:]
                sq += float(np.sum((p7[:3] - r["pose7"][:3]) ** 2) + np.sum((q - r["pose7"][3:]) ** 2))
                cnt += 7
                assert np.abs(p16 - r["pose16"]).max() <= max(10 * pose_tol, 1e-5), f"frame {k}: pose array differs"
        rmse = (sq / max(cnt, 1)) ** 0.5
        kf_worst = sysdiff.compare_keyframes(ref, gpu, 10 * pose_tol, what="end of stream")
        c = gpu.counters()
        assert len(ref.keyframe_ids()) >= min_kf and c["ba_solves"] >= min_ba, (len(ref.keyframe_ids()), c)
        assert int(gpu.state()[11]) >= min_kf_created, f"only {int(gpu.state()[11])} keyframes were created"
        assert worst_px <= px_tol, worst_px
        assert rmse <= pose_tol, f"pose RMSE {rmse} (worst {worst_pose})"
        assert worst_x <= max(10 * pose_tol, 1e-5), worst_x
        init_gpu = gpu.pose7()[1]
        if aligned_tol is not None:
            scale, dpos, drot = sysdiff.sim3_aligned_diff(traj_ref, traj_gpu)
            print(f"\n  after Sim(3) alignment of the trajectory: scale {scale:.8f}, centres {dpos:.2e} of the extent, rotations {drot:.2e} rad")
            assert dpos <= aligned_tol and drot <= aligned_tol, (scale, dpos, drot)
        print(f"\n  frames {len(frames)}  keyframes {len(ref.keyframe_ids())}  BA solves {c['ba_solves']}  merges {c['merges']}  "
              f"pose RMSE {rmse:.2e} (worst {worst_pose:.2e})  keyframe poses {kf_worst:.2e}  map points {worst_x:.2e}  pixels {worst_px:.2e}  "
              f"own two-view pose vs reference {sysdiff.pose_diff(init_pose, init_gpu):.2e}")
        return rmse, sysdiff.pose_diff(init_pose, init_gpu)
    finally:
        ref.close()
        gpu.close()


def test_system_equals_reference_150_frames():
    """shipped configuration (cell 40): cold start, two-view initialisation, 8 keyframes, local BA, merges, culling; pose RMSE <= 1e-5"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(150)]
    rmse, dinit = _differential(frames, w, h, 40, True, 1e-5, 8, 6)
    assert dinit <= 5e-3


def test_system_equals_reference_without_the_hook():
    """the same stream with the map started from OUR OWN five-point result (no alva_system_debug_set_init_pose): identical DISCRETE
    state on every one of 160 frames (statuses, counters, keypoint ids in container order, keyframes, map tables, medoids).  The two maps
    start from two-view poses that differ at OpenGV's refinement noise floor, so the raw poses are compared at 1e-3
    (round 2: 2e-2) and, because most of that difference is the map's gauge (scale / world frame), again after a Sim(3) alignment of the
    trajectories at 1e-4."""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(160)]
    _differential(frames, w, h, 40, False, 1e-3, 8, 6, px_tol=0.1, aligned_tol=ALIGNED_TOL)


def test_system_equals_reference_2000_keypoints():
    """BASELINE configs[1] geometry (cell 12 => 2120 keypoints)"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(60)]
    _differential(frames, w, h, 12, True, 1e-5, 3, 2)


def _against_recording(frames, w, h, cell, recording, pose_tol):
    """The HIP path against the COMMITTED recording of one majority run of the reference on this stream
    (tests/golden/make_system_long_golden.py: per-frame status, counters, digests of keypoint ids / flags, keypoint pixels, keyframe
    ids, map-point table and descriptor medoids; poses): discrete state exact on every frame, pose RMSE <= pose_tol."""
    import os
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_system_long_golden", os.path.join(sys_path, "make_system_long_golden.py"))
    gold = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gold)
    R = np.load(os.path.join(sys_path, recording))
    assert len(R["status"]) == len(frames)
    gpu = sysdiff.GpuSystem(w, h, cell)
    try:
        sq, cnt = 0.0, 0
        for k, rgba in enumerate(frames):
            if R["status"][k] == 1 and (k == 0 or R["status"][k - 1] != 1):
                gpu.set_init_pose(R["pose7"][k])
            st, p7, _ = gpu.step(rgba, 33.0 * k)
            st_, state_, dig_, _ = gold.frame_record(gpu, st, p7)
            assert st_ == int(R["status"][k]), f"frame {k}: status {st_} != recording {int(R['status'][k])}"
            assert np.array_equal(state_, R["state"][k]), f"frame {k}: state differs from the recording"
            for what, a, b in zip(("keypoint ids / flags", "keypoint pixels", "keyframe ids", "map-point table", "descriptor medoids"), dig_, R["digests"][k]):
                assert a == b, f"frame {k}: {what} differ from the recording"
            if st == 1:
                r7 = R["pose7"][k]
                q = p7[3:] if np.dot(p7[3:], r7[3:]) >= 0 else -p7[3:]
                sq += float(np.sum((p7[:3] - r7[:3]) ** 2) + np.sum((q - r7[3:]) ** 2))
                cnt += 7
        rmse = (sq / max(cnt, 1)) ** 0.5
        assert rmse <= pose_tol, f"pose RMSE {rmse} against the recording"
        return rmse
    finally:
        gpu.close()


def _long_stream(w, h, n, steps, canvas_seed, noise_seed):
    """a crop sequence of n frames with noise, played forwards and backwards for `steps` steps: (base gray frames, base index per step, frames)"""
    canvas = synth.texture_canvas(w, h, canvas_seed)
    base = np.stack([synth.frame_gray(canvas, k, w, h, noise_seed=noise_seed) for k in range(n)])
    period = 2 * (n - 1)
    index = [(k % period) if (k % period) < n else period - (k % period) for k in range(steps)]
    rgba = [synth.gray_to_rgba(g) for g in base]
    return base, index, [rgba[i] for i in index]


def _differential_long(base, index, frames, w, h, cell, pose_tol, min_kf, min_ba, **kw):
    """A long stream against ONE run of the reference.  Inside this process the reference is not reproducible on these streams (Ceres keeps
    its parameter blocks ordered by ADDRESS, so its reduction orders follow the heap layout; two runs differ from the first local BA on and
    about one in five of the 560-frame, 2500-keypoint stream ends on another discrete path).  oracle/_ref/ref_run (tests/ref_runner.py) runs
    it in a process that holds nothing else, without address-space randomisation: its records are a function of the frames, bit for bit,
    and the comparison needs no second attempt."""
    import ref_runner
    return _differential(frames, w, h, cell, True, pose_tol, min_kf, min_ba, reference=ref_runner.run_reference(base, index, w, h, cell), **kw)


def test_system_equals_reference_long_stream():
    """660 frames (200-frame crop sequence with noise, forwards / backwards): more than 30 keyframes -- the 30-keyframe window, the keyframe
    filter of Mapper::optimize (from keyframe 20 on), the second local-map round; >= 120 tracked frames after initialisation"""
    w, h = 640, 480
    base, index, frames = _long_stream(w, h, 200, 660, 7, 11)
    _differential_long(base, index, frames, w, h, 40, 1e-5, 8, 30)


def test_system_equals_reference_long_stream_2000_keypoints():
    """the same stream at BASELINE configs[1]'s geometry (cell 12 => ~2500 keypoints), 560 frames: > 30 keyframes, ~15 culled by the keyframe
    filter, ~3000 map-point merges"""
    w, h = 640, 480
    base, index, frames = _long_stream(w, h, 200, 560, 7, 11)
    _differential_long(base, index, frames, w, h, 12, 1e-5, 8, 25)


def test_system_equals_reference_rotating_camera_with_noise():
    w, h = 640, 480
    f = sysdiff.intrinsics(w, h)[0]
    canvas = synth.texture_canvas(w, h, 5)
    frames = [synth.plane_stream_frame(canvas, k, w, h, f, noise_seed=100) for k in range(130)]
    _differential(frames, w, h, 40, True, 1e-5, 4, 2)


def test_system_equals_reference_tracking_loss_and_reset():
    """scene cut -> KLT / pose failures -> resetFrame, poseFailedCounter_, status 2, re-initialisation (visual_frontend.cpp:73-92,
    :318-330, :383-399): the status sequence and every counter follow the reference"""
    w, h = 640, 480
    canvas, other = synth.texture_canvas(w, h, 7), synth.texture_canvas(w, h, 99)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(40)]
    frames += [synth.gray_to_rgba(synth.frame_gray(other, 3 * (k % 2) * 20 + k, w, h)) for k in range(40)]
    frames += [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(40)]
    _differential(frames, w, h, 40, True, 1e-5, 0, 0)


def test_system_equals_reference_distortion_and_clahe():
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(60)]
    _differential(frames, w, h, 40, True, 1e-5, 2, 1, clahe=True, dist=(-0.12, 0.03, 0.0006, -0.0004))


def test_js_wrapper_conventions():
    """AlvaAR (src/system.js) call pattern: status 3 while initialising, pose only on status 1, getFramePoints = 2-D keypoints,
    findCameraPoseWithIMU always returns a pose, findPlane needs 32 observed 3-D points"""
    from alvaar_amd.system import AlvaAR
    w, h = 640, 480
    ar = AlvaAR.Initialize(w, h)
    canvas = synth.texture_canvas(w, h, 7)
    frame = lambda k: synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
    pose, status = ar.findCameraPose(frame(0))
    assert status == 3 and pose is None
    pts2d = ar.getFramePoints()
    assert 50 < len(pts2d) <= 2048 and all(20 <= p["x"] < w - 20 + 3 for p in pts2d)
    assert ar.findPlane() is None
    statuses = []
    for k in range(1, 40):
        pose, status = ar.findCameraPose(frame(k), 33.0 * k)
        statuses.append(status)
    k0 = statuses.index(1)
    assert statuses[:k0] == [3] * k0 and set(statuses[k0:]) == {1}
    assert pose[15] == 1.0 and pose[3] == pose[7] == pose[11] == 0.0
    assert abs(np.linalg.norm(ar.pose7()[1][:3]) - 1.0) < 1e-9     # twc.normalize() at initialisation (visual_frontend.cpp:547)
    ids, px, is3d = ar.keypoints()
    assert is3d.sum() >= 30
    plane = ar.findPlane()
    assert plane is not None and plane[15] == 1.0
    R = plane.reshape(4, 4)[:3, :3].T
    Rx = np.array([[1, 0, 0], [0, np.cos(1.0), -np.sin(1.0)], [0, np.sin(1.0), np.cos(1.0)]])
    n = (R @ Rx.T)[:, 0]
    assert abs(abs(n[2]) - 1.0) < 0.05, n                              # the scene IS a fronto-parallel plane
    ar.reset()
    pose, status = ar.findCameraPose(frame(0))
    assert status == 3
    imu_pose = ar.findCameraPoseWithIMU(frame(1), (1.0, 0.0, 0.0, 0.0))
    assert imu_pose is not None and np.allclose(imu_pose.reshape(4, 4)[:3, :3], np.eye(3)) and imu_pose[15] == 1.0
    ar.close()


def test_configure_failure_leaves_the_object_unconfigured():
    from alvaar_amd.system import AlvaAR
    from alvaar_amd.capi import AlvaError
    with pytest.raises(AlvaError):
        AlvaAR(30, 20)


def test_cpp_program_through_the_int_offset_methods(tmp_path):
    """A C++ program that LINKS libalvaar_hip.so and calls alva::System through the reference's own signatures -- every buffer a 32-bit
    heap offset passed as int (system.hpp:30-36), buffers from mmap(MAP_32BIT): same statuses and poses as the Python caller."""
    import subprocess
    from pathlib import Path
    from alvaar_amd.system import AlvaAR
    root = Path(__file__).resolve().parent.parent
    w, h, n = 640, 480, 40
    canvas = synth.texture_canvas(w, h, 7)
    frames = np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(n)])
    (tmp_path / "frames.bin").write_bytes(frames.tobytes())
    exe = tmp_path / "system_int_offsets"
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-I{root / 'include'}", "-o", str(exe), str(root / "tests" / "cpp" / "system_int_offsets.cpp"),
                           f"-L{root / 'alvaar_amd'}", "-lalvaar_hip", f"-Wl,-rpath,{root / 'alvaar_amd'}"])
    out = subprocess.check_output([str(exe), str(w), str(h), str(tmp_path / "frames.bin"), str(n)], text=True).strip().splitlines()
    assert len(out) == n + 1 and out[-1].startswith("plane ")
    statuses = [int(l.split()[1]) for l in out[:n]]
    k0 = statuses.index(1)
    assert statuses[:k0] == [3] * k0 and set(statuses[k0:]) == {1} and 15 <= k0 <= 25
    # the clock-seeded default configuration is not bit-reproducible between runs; the discrete behaviour and the geometry are
    poses = np.array([[float(v) for v in l.split()[3:]] for l in out[:n]])
    assert np.all(poses[:, 15] == 1.0) and np.all(poses[:, [3, 7, 11]] == 0.0)
    assert abs(np.linalg.norm(poses[k0, 12:15]) - 1.0) < 1e-6          # unit baseline at initialisation
    d = np.array([2.0, 1.0, 0.0]) / np.sqrt(5.0)
    assert np.abs(poses[k0, 12:15] - d).max() < 0.03
    assert int(out[-1].split()[1]) in (0, 1) and int(out[-1].split()[3]) == 1


def test_tracking_step_variants_agree_bitwise(monkeypatch):
    """the tracking step three ways -- slot-wise (default: tracker launch, retry launch, compaction), with explicit keypoint lists
    (ALVA_TRACK_LISTS=1: the reference's two lists built on the device) and composed from the fine-grained stages
    (ALVA_TRACK_UNFUSED=1: host round trips between them, the reference's own statement order): same statuses, states, keypoints and
    poses -- the two fused forms to the last bit, the composed one to 1e-9 -- through initialisation, keyframes, merges and local BA; and
    the default form once more with stream waits instead of completion words and without the warm start (ALVA_NO_POLL, ALVA_NO_WARMUP)"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(70)]
    runs = []
    for env in ({}, {"ALVA_TRACK_LISTS": "1"}, {"ALVA_TRACK_UNFUSED": "1"}, {"ALVA_NO_POLL": "1", "ALVA_NO_WARMUP": "1"}):
        for key in ("ALVA_TRACK_LISTS", "ALVA_TRACK_UNFUSED", "ALVA_NO_POLL", "ALVA_NO_WARMUP"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        gpu = sysdiff.GpuSystem(w, h, 12)
        rec = []
        for k, f in enumerate(frames):
            st, p7, p16 = gpu.step(f, 33.0 * k)
            ids, px, un, i3, hd = gpu.frame_keypoints()
            rec.append((st, list(gpu.state()), ids.copy(), px.copy(), un.copy(), i3.copy(), p7.copy()))
        assert gpu.counters()["ba_solves"] >= 2 and sum(r[0] == 1 for r in rec) >= 30
        gpu.close()
        runs.append(rec)
    worst = 0.0
    for other, name in ((runs[1], "lists"), (runs[2], "unfused"), (runs[3], "lists")):   # the last: stream waits, cold start -- bitwise too
        for k, (a, b) in enumerate(zip(runs[0], other)):
            assert a[0] == b[0] and a[1] == b[1], f"{name} frame {k}: status / state"
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[5], b[5]), f"{name} frame {k}: keypoint ids / flags"
            if name == "lists":
                assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), f"{name} frame {k}: pixels"
                assert np.array_equal(a[6].view(np.uint64), b[6].view(np.uint64)), f"{name} frame {k}: pose"
            else:
                # the composed step hands the P3P pose to the refinement through the host as t | q (the reference's Sophus::SE3d hand-off,
                # visual_frontend.cpp:306-375), the fused steps keep it on the device: last-bit differences, no more
                assert len(a[3]) == 0 or np.abs(a[3] - b[3]).max() <= 1e-2, f"{name} frame {k}: pixels"
                worst = max(worst, float(np.abs(a[6] - b[6]).max()))
    assert worst <= 1e-9, worst
    print(f"\n  composed vs fused step: worst pose difference {worst:.2e}")


def test_pose_launch_forms_agree_bitwise(monkeypatch):
    """the frame's tail four ways: ONE launch queued behind the tracker (k_pose_all: compaction -> P3P -> PnP, the default); the compaction
    kernel + the fused P3P -> PnP launch (ALVA_NO_POSE_ALL=1); three separate launches (ALVA_POSE_UNFUSED=1, round 5's chain); and the
    default with the host's answer made LATE on purpose (ALVA_POSE_ALL_LATE_US: the queued launch gives up, says so, and the host
    falls back to the separate launches -- what happens under a tool that makes launches synchronous).  Same arithmetic in all of them:
    statuses, states, keypoints and poses equal to the last bit through initialisation, keyframes and local BA."""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(60)]
    keys = ("ALVA_NO_POSE_ALL", "ALVA_POSE_UNFUSED", "ALVA_POSE_ALL_LATE_US")
    runs = []
    for env in ({}, {"ALVA_NO_POSE_ALL": "1"}, {"ALVA_POSE_UNFUSED": "1"}, {"ALVA_POSE_ALL_LATE_US": "1500"}):
        for key in keys:
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        gpu = sysdiff.GpuSystem(w, h, 12)
        rec = []
        for k, f in enumerate(frames):
            st, p7, p16 = gpu.step(f, 33.0 * k)
            ids, px, un, i3, hd = gpu.frame_keypoints()
            rec.append((st, list(gpu.state()), ids.copy(), px.copy(), un.copy(), i3.copy(), p7.copy()))
        assert sum(r[0] == 1 for r in rec) >= 25
        gpu.close()
        runs.append(rec)
    for key in keys:
        monkeypatch.delenv(key, raising=False)
    for other, name in zip(runs[1:], ("compaction + fused pose", "three launches", "late answer -> fallback")):
        for k, (a, b) in enumerate(zip(runs[0], other)):
            assert a[0] == b[0] and a[1] == b[1], f"{name}, frame {k}: status / state"
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[5], b[5]), f"{name}, frame {k}: keypoint ids / flags"
            assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), f"{name}, frame {k}: pixels"
            assert np.array_equal(a[6].view(np.uint64), b[6].view(np.uint64)), f"{name}, frame {k}: pose"


def test_carried_slot_table_equals_assembled(monkeypatch):
    """Between two keyframes the tracker's slot table is CARRIED on the device (the host names, per slot, the slot it was; track_slots.hpp)
    instead of assembled from the map and written over the bus.  Three runs of one stream through initialisation, keyframes and local BA:
    carried (the default), every frame assembled (ALVA_NO_CARRY=1), and carried with the map layer comparing every carried table against
    the assembled one (ALVA_CHECK_CARRY=1: positions, flags, world points; it aborts on a difference).  Statuses, states, keypoints
    and poses equal to the last bit, and the default run did carry most of its frames."""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(70)]
    keys = ("ALVA_NO_CARRY", "ALVA_CHECK_CARRY")
    runs, carried = [], []
    for env in ({}, {"ALVA_NO_CARRY": "1"}, {"ALVA_CHECK_CARRY": "1"}):
        for key in keys:
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        gpu = sysdiff.GpuSystem(w, h, 12)
        gpu.ar.timing_fine()
        rec = []
        for k, f in enumerate(frames):
            st, p7, p16 = gpu.step(f, 33.0 * k)
            ids, px, un, i3, hd = gpu.frame_keypoints()
            rec.append((st, list(gpu.state()), ids.copy(), px.copy(), un.copy(), i3.copy(), p7.copy()))
        fine = gpu.ar.timing_fine()
        carried.append((int(fine["#frames: slot table carried"]), int(fine["#frames: slot table assembled"])))
        assert sum(r[0] == 1 for r in rec) >= 30
        gpu.close()
        runs.append(rec)
    for key in keys:
        monkeypatch.delenv(key, raising=False)
    assert carried[0][0] >= 25 and carried[0][1] >= 2, carried     # most tracked frames carried; the frames behind a keyframe assembled
    assert carried[1][0] == 0, carried
    assert carried[2][0] == carried[0][0], carried
    for other, name in zip(runs[1:], ("assembled every frame", "carried + checked")):
        for k, (a, b) in enumerate(zip(runs[0], other)):
            assert a[0] == b[0] and a[1] == b[1], f"{name}, frame {k}: status / state"
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[5], b[5]), f"{name}, frame {k}: keypoint ids / flags"
            assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), f"{name}, frame {k}: pixels"
            assert np.array_equal(a[6].view(np.uint64), b[6].view(np.uint64)), f"{name}, frame {k}: pose"


def test_imu_surface_equals_reference_composition():
    """System::findCameraPoseWithIMU (system.cpp:57-104): orientation from the IMU quaternion (w, -x, y, z), inverted; translation = the
    visual translation integrated over the tracked frames (reset of the increment on every frame that is not tracked).  Expected arrays come
    from the reference's own classes (oracle/ref_shim_system.cpp: ref_imu_pose = Eigen quaternion -> matrix -> inverse, Sophus::SE3d,
    Utils::toPoseArray) fed with the reference System's status and translation per frame."""
    import ctypes as C
    import oracles
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(70)]
    ref, rec, _ = _reference_run(frames, w, h, 40)
    ref.close()
    L = oracles.ref_lib()
    L.ref_imu_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.RandomState(4)
    gpu = sysdiff.GpuSystem(w, h, 40)
    cur, prev = np.zeros(3), np.zeros(3)
    worst, tracked = 0.0, 0
    for k, f in enumerate(frames):
        r = rec[k]
        if r["status"] == 1 and (k == 0 or rec[k - 1]["status"] != 1):
            gpu.set_init_pose(r["pose7"])
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        got = gpu.ar.findCameraPoseWithIMU(f, q, motion=[(33.0 * k, 0.01, 0.02, 0.03, 0.0, 9.8, 0.1)], timestamp_ms=33.0 * k)
        assert got is not None                       # the call always reports 1 (system.cpp:103)
        if r["status"] == 1:                         # :89-99
            t = r["pose7"][:3]
            cur = cur + t - prev
            prev = t.copy()
            tracked += 1
        else:
            prev = np.zeros(3)
        want = np.zeros(16, np.float32)
        qq, cc = np.ascontiguousarray(q), np.ascontiguousarray(cur)
        L.ref_imu_pose(qq.ctypes.data, cc.ctypes.data, want.ctypes.data)
        worst = max(worst, float(np.abs(got - want).max()))
        assert np.abs(got - want).max() <= 2e-6, (k, got, want)
    gpu.close()
    assert tracked >= 30 and np.linalg.norm(cur) > 0.1
    print(f"\n  IMU surface: worst difference to the reference composition {worst:.1e} over {len(frames)} frames ({tracked} tracked)")


def test_system_equals_reference_explicit_reset():
    """System::reset called by the host in the middle of a tracked stream: frame, map and counters cleared, motion model / p3pReq_ / the
    detector's adaptive threshold kept; re-initialisation and 60 more frames compared like the ones before"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(110)]
    _differential(frames, w, h, 40, True, 1e-5, 2, 1, reset_at=(48,))


def test_system_equals_reference_1280x720():
    """BASELINE configs[4]'s geometry through the surface: 1280x720, cell 15 (4080 cells)"""
    w, h = 1280, 720
    canvas = synth.texture_canvas(w, h, 9)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=3)) for k in range(60)]
    _differential(frames, w, h, 15, True, 1e-5, 3, 2)


def test_system_equals_reference_1280x720_long_stream():
    """configs[4]'s geometry over a LONG stream: 1280x720, cell 15 (~5500 keypoints), 440 frames forwards / backwards => more than 20
    keyframes: the keyframe filter of Mapper::optimize (from keyframe 20 on, mapper.cpp:66-142), map-point culling and local BA over a
    full covisibility window at the size configs[4] names (reference: visual_frontend.cpp:517-552, mapper.cpp:9-64)"""
    w, h = 1280, 720
    base, index, frames = _long_stream(w, h, 150, 440, 9, 3)
    _differential_long(base, index, frames, w, h, 15, 1e-5, 8, 20, min_kf_created=21)


def test_concurrent_sessions_equal_their_solo_runs():
    """four alva::System sessions driven from four host threads at once (different grids, different streams): every session's statuses,
    poses and keypoints are those of the same session run alone -- nothing in the library is shared between sessions but the device"""
    import threading
    w, h = 640, 480
    specs = [(40, 7, False), (12, 5, True), (24, 9, False), (12, 7, False)]   # cell size, canvas seed, noise

    def frames_of(seed, noise):
        canvas = synth.texture_canvas(w, h, seed)
        return [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11 if noise else None)) for k in range(70)]

    def run(cell, frames, out):
        gpu = sysdiff.GpuSystem(w, h, cell)
        for k, f in enumerate(frames):
            st, p7, p16 = gpu.step(f, 33.0 * k)
            ids, px, un, i3, hd = gpu.frame_keypoints()
            out.append((st, p7.copy(), ids.copy(), px.copy()))
        out.append(gpu.counters())
        gpu.close()

    streams = [frames_of(seed, noise) for _, seed, noise in specs]
    solo = []
    for (cell, _, _), fr in zip(specs, streams):
        rec = []
        run(cell, fr, rec)
        solo.append(rec)
    together = [[] for _ in specs]
    threads = [threading.Thread(target=run, args=(cell, fr, rec)) for (cell, _, _), fr, rec in zip(specs, streams, together)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s, (a, b) in enumerate(zip(solo, together)):
        assert len(a) == len(b) == 71, s
        assert a[-1] == b[-1] and a[-1]["ba_solves"] >= 1, (s, a[-1], b[-1])
        for k in range(70):
            assert a[k][0] == b[k][0], (s, k)
            assert np.array_equal(a[k][1].view(np.uint64), b[k][1].view(np.uint64)), (s, k)
            assert np.array_equal(a[k][2], b[k][2]) and np.array_equal(a[k][3].view(np.uint32), b[k][3].view(np.uint32)), (s, k)


@pytest.mark.parametrize("mode", ["one_lane", "three_lanes", "two_lanes_shared_session_streams", "no_lockstep"])
def test_group_sessions_equal_their_solo_runs(mode):
    """alva_system_group: six sessions (different grids / streams, one 1280x720) advanced frame by frame on TWO host threads -- three
    fibers per thread, every GPU wait of a session running the thread's other sessions: statuses, poses (bitwise), keypoint ids and
    pixels (bitwise) and the counters are those of each session's solo run.  With lanes (lane.hpp) the launches of the sessions' tracking
    frames are issued once per kind for all sessions of a lane -- sessions of different image sizes and keypoint counts in one launch,
    deposits arriving from both worker threads -- and the launch statistics must show that they really were shared.  one_lane: all six
    sessions; three_lanes: two each (one per worker); two_lanes_shared_session_streams: the sessions' OWN streams are shared too."""
    import torch
    from alvaar_amd.system import AlvaAR, SystemGroup
    specs = [(640, 480, 40, 7, False), (640, 480, 12, 5, True), (640, 480, 24, 9, False), (640, 480, 12, 7, False), (1280, 720, 15, 9, True),
             (640, 480, 40, 3, True)]
    n = 60

    def frames_of(w, h, seed, noise):
        canvas = synth.texture_canvas(w, h, seed)
        fr = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11 if noise else None)) for k in range(n)]
        if seed == 3:   # the LAST session loses tracking in the middle (a scene cut for 14 frames: KLT / pose failures, status 2, reset,
            other = synth.texture_canvas(w, h, 99)   # re-initialisation) while its lane mates keep tracking: frames that deposit images
            for k in range(30, 44):                  # and then track nothing, sessions that leave the lock-step
                fr[k] = synth.gray_to_rgba(synth.frame_gray(other, 3 * (k % 2) * 20 + k, w, h))
        return np.stack(fr)

    dev = [torch.from_numpy(frames_of(w, h, seed, noise)).cuda() for w, h, cell, seed, noise in specs]

    def record(ar):
        ids, px, i3 = ar.keypoints()
        return ar.pose7()[0].copy(), ids.copy(), px.copy(), list(ar.state())

    solo = []
    for (w, h, cell, seed, noise), fr in zip(specs, dev):
        ar = AlvaAR(w, h, cell_size=cell, random_sampling=False)
        rec = []
        for k in range(n):
            st = ar.find_camera_pose_device(int(fr[k].data_ptr()), 33.0 * k)
            rec.append((st,) + record(ar))
        rec.append(ar.counters())
        ar.close()
        solo.append(rec)
    assert 2 in [r[0] for r in solo[-1][:n]], "the scene cut was meant to cost the last session its tracking"
    group = SystemGroup([], 2)
    group.set_lockstep(mode != "no_lockstep")
    group.set_lanes({"one_lane": 1, "three_lanes": 3}.get(mode, 2))
    shared = mode == "two_lanes_shared_session_streams"
    sessions = [AlvaAR(w, h, cell_size=cell, random_sampling=False, hip_stream=group.stream(i % 2) if shared else None)
                for i, (w, h, cell, seed, noise) in enumerate(specs)]
    group.set_sessions(sessions)
    together = [[] for _ in specs]
    for k in range(n):
        st = group.step_device([int(fr[k].data_ptr()) for fr in dev], 33.0 * k)
        for i, ar in enumerate(sessions):
            together[i].append((int(st[i]),) + record(ar))
    for i, ar in enumerate(sessions):
        together[i].append(ar.counters())
    launches, carried = group.launch_stats()
    for ar in sessions:
        ar.close()
    group.close()
    if mode == "no_lockstep":
        assert launches == 0                                       # no lane: every session launches for itself
    else:
        assert carried > 1.2 * launches > 0, (mode, launches, carried)   # (sessions of different sizes, not all tracking at once)
    for i, (a, b) in enumerate(zip(solo, together)):
        assert a[-1] == b[-1] and (a[-1]["ba_solves"] >= 1 or i == len(specs) - 1), (i, a[-1], b[-1])   # (the last session was reset late)
        for k in range(n):
            assert a[k][0] == b[k][0] and a[k][4] == b[k][4], (i, k)
            assert np.array_equal(a[k][1].view(np.uint64), b[k][1].view(np.uint64)), (i, k)
            assert np.array_equal(a[k][2], b[k][2]) and np.array_equal(a[k][3].view(np.uint32), b[k][3].view(np.uint32)), (i, k)


def test_next_frame_hints_do_not_change_results():
    """alva_system_hint_next_frame_device: the next frame's gray image + pyramid built ahead on a second stream.  A 70-frame stream
    (initialisation, keyframes with local BA, a forwards / backwards sweep) with a correct hint on most frames, a WRONG hint on every
    seventh (dropped, the frame is rebuilt), none on every fifth, and one frame passed twice: statuses, poses (bitwise), keypoint ids and
    pixels (bitwise) and the counters equal the run without hints"""
    import torch
    from alvaar_amd.system import AlvaAR
    w, h, n = 640, 480, 70
    canvas = synth.texture_canvas(w, h, 7)
    dev = torch.from_numpy(np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(40)])).cuda()
    order = [k if k < 40 else 79 - k for k in range(n)]
    order[50] = order[49]   # the same frame twice in a row (hinted correctly: the hint names the buffer it is reading from)

    def run(hints: bool):
        ar = AlvaAR(w, h, cell_size=12, random_sampling=False)
        rec = []
        for k in range(n):
            nxt = None
            if hints and k + 1 < n and k % 5 != 4:
                nxt = int(dev[order[k + 1] if k % 7 != 6 else (order[k + 1] + 3) % 40].data_ptr())
            st = ar.find_camera_pose_device(int(dev[order[k]].data_ptr()), 33.0 * k, nxt)
            ids, px, i3 = ar.keypoints()
            rec.append((st, ar.pose7()[0].copy(), ids.copy(), px.copy(), list(ar.state())))
        rec.append(ar.counters())
        ar.close()
        return rec

    a, b = run(False), run(True)
    assert a[-1] == b[-1] and a[-1]["ba_solves"] >= 1, (a[-1], b[-1])
    for k in range(n):
        assert a[k][0] == b[k][0] and a[k][4] == b[k][4], k
        assert np.array_equal(a[k][1].view(np.uint64), b[k][1].view(np.uint64)), k
        assert np.array_equal(a[k][2], b[k][2]) and np.array_equal(a[k][3].view(np.uint32), b[k][3].view(np.uint32)), k


@pytest.mark.parametrize("name", ["560_cell12", "660_cell40", "440_720p"])
def test_long_stream_equals_the_recorded_reference_run(name):
    """Each of the three long streams against the COMMITTED recording of a majority run of the reference on it (tests/golden/
    make_system_long_golden.py; the reproducible run of oracle/_ref/ref_run walks the same discrete path): a pin that needs no reference
    at run time -- every frame's status,
    counters, keypoint ids / flags / pixels (digests of the bytes), keyframe ids, map-point table and descriptor medoids equal the
    recording, pose RMSE <= 1e-5 (from the recording's two-view pose: the init-pose hook, see README "pose parity")."""
    import importlib.util
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_system_long_golden", os.path.join(gdir, "make_system_long_golden.py"))
    gold = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gold)
    w, h, _, n_frames, cell, _, _ = gold.STREAMS[name]
    rmse = _against_recording(gold.stream(name), w, h, cell, gold.FILES[name], 1e-5)
    print(f"\n  {n_frames} frames ({name}) against the recording: pose RMSE {rmse:.2e}")
