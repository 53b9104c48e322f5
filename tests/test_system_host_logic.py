"""Host logic of the System surface (a1, a10, a13, f1 of SURVEY.md §8) WITHOUT a GPU: the product's map layer
(alvaar_amd/csrc/slam/ -- keyframe policy, map bookkeeping, container orders, BA graph construction + write-back, culling, descriptor
medoids) compiled over the reference's own L1 stages (oracle/sys_cpu.cpp) against the reference's System (oracle/ref_shim_system.cpp)
on the same frames.  The stage arithmetic is the same code on both sides, so every difference would be a bookkeeping difference:
status sequence, state counters, keypoints in container order (bitwise pixels), keyframes, covisibility, map point tables and
descriptor medoids must be IDENTICAL; poses / points agree to rounding (the pose algebra is ours: <= 1e-9)."""
import numpy as np
import pytest

from alvaar_amd import synth
import sysdiff

pytestmark = pytest.mark.ref


def _run(frames, w, h, cell, n_min_kf, clahe=False, dist=(0.0, 0.0, 0.0, 0.0), ts=lambda k: 33.0 * k, reset_at=()):
    ref, cpu = sysdiff.RefSystem(w, h, cell, clahe, dist), sysdiff.CpuSystem(w, h, cell, clahe, dist)
    try:
        statuses, worst_pose, worst_x = [], 0.0, 0.0
        for k, rgba in enumerate(frames):
            if k in reset_at:   # the caller's System::reset (system.cpp:42-55) between two frames
                ref.reset()
                cpu.reset()
            s1, p1, a1 = ref.step(rgba, ts(k))
            s2, p2, a2 = cpu.step(rgba, ts(k))
            assert s1 == s2, f"frame {k}: status {s1} != {s2}"
            statuses.append(s1)
            worst_x = max(worst_x, sysdiff.compare(ref, cpu, 1e-9, px_exact=True, xyz_tol=1e-9, what=f"frame {k}"))
            d = sysdiff.pose_diff(p1, p2)
            assert d <= 1e-9, f"frame {k}: pose differs by {d}"
            assert np.abs(a1 - a2).max() <= 1e-6
            worst_pose = max(worst_pose, d)
            if k % 10 == 0:
                sysdiff.compare_keyframes(ref, cpu, 1e-9, what=f"frame {k}")
        sysdiff.compare_keyframes(ref, cpu, 1e-9, what="last frame")
        assert len(ref.keyframe_ids()) >= n_min_kf or ref.state()[11] >= n_min_kf
        return statuses, cpu.counters(), worst_pose, worst_x
    finally:
        ref.close()
        cpu.close()


def test_shipped_configuration_translating_camera():
    """cell 40 (192 keypoints): cold start, five-point initialisation, 8 keyframes, local BA from keyframe 2, merges, culling"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = (synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(150))
    statuses, cnt, _, _ = _run(frames, w, h, 40, 8)
    k0 = statuses.index(1)
    assert statuses[:k0] == [3] * k0 and set(statuses[k0:]) == {1} and 15 <= k0 <= 25
    assert cnt["ba_solves"] >= 6 and cnt["merges"] > 0


def test_2000_keypoint_workload():
    """BASELINE configs[1] geometry: cell 12 => 2120 cells"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = (synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(45))
    statuses, cnt, _, _ = _run(frames, w, h, 12, 3)
    assert statuses[-1] == 1 and cnt["ba_solves"] >= 1


def test_rotating_camera_with_noise():
    """camera rotating and translating in front of a plane, +-4 gray noise: KLT failures, P3P / PnP outliers, keyframes by parallax"""
    w, h = 640, 480
    f = sysdiff.intrinsics(w, h)[0]
    canvas = synth.texture_canvas(w, h, 5)
    frames = (synth.plane_stream_frame(canvas, k, w, h, f, noise_seed=100) for k in range(130))
    statuses, cnt, _, _ = _run(frames, w, h, 40, 4)
    assert 1 in statuses and cnt["ba_solves"] >= 2


def test_tracking_loss_and_reset():
    """a scene cut (unrelated texture) makes KLT / pose estimation fail: keypoints are dropped, poseFailedCounter_ accumulates, the
    tracker resets (status 2) and re-initialises -- the failure paths of visual_frontend.cpp:73-92, :318-330, :383-399"""
    w, h = 640, 480
    canvas, other = synth.texture_canvas(w, h, 7), synth.texture_canvas(w, h, 99)

    def frames():
        for k in range(40):
            yield synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
        for k in range(40):
            yield synth.gray_to_rgba(synth.frame_gray(other, 3 * (k % 2) * 20 + k, w, h))
        for k in range(40):
            yield synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
    statuses, _, _, _ = _run(frames(), w, h, 40, 0)
    assert 2 in statuses or 3 in statuses[41:], statuses


def test_distortion_and_clahe_wired():
    """non-zero radial / tangential coefficients and CLAHE through the whole path (camera_calibration.cpp:34-72, visual_frontend.cpp:678-681)"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = (synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(60))
    statuses, _, _, _ = _run(frames, w, h, 40, 2, clahe=True, dist=(-0.12, 0.03, 0.0006, -0.0004))
    assert 1 in statuses


@pytest.mark.parametrize("cell,frames_total", [(40, 660), (12, 560)])
def test_long_stream_keyframe_window_and_filter(cell, frames_total):
    """660 frames at cell 40 / 560 frames at cell 12 = BASELINE configs[1]'s 2000 keypoints (the 200-frame crop sequence forwards /
    backwards): more than 30 keyframes, so the 30-keyframe window
    (mapper.cpp:24-28), the keyframe filter of Mapper::optimize from keyframe 20 on (:74-141) and the second local-map round
    (:316-330) all run; every mirror the map layer keeps beside the reference's containers is checked on every read
    (ALVA_CHECK_OBS_MIRROR=1)"""
    import os
    w, h, n = 640, 480, 200
    canvas = synth.texture_canvas(w, h, 7)
    base = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(n)]
    period = 2 * (n - 1)

    def frames():
        for k in range(frames_total):
            r = k % period
            yield base[r if r < n else period - r]
    os.environ["ALVA_CHECK_OBS_MIRROR"] = "1"
    try:
        # both sides run the reference's own Ceres here, whose reduction orders follow heap addresses: two runs of the REFERENCE differ
        # from each other by ~1e-13 from the first local BA on (tools/ref_determinism_probe.py) and the pipeline amplifies that to a
        # changed discrete decision within a few hundred frames about once in five runs of the 2500-keypoint stream -- so a mismatch
        # is retried; agreement with some run of the reference over the whole stream is the claim
        last = None
        for _ in range(4):
            try:
                statuses, cnt, _, _ = _run(frames(), w, h, cell, 31)
                last = None
                break
            except AssertionError as e:
                last = e
        if last is not None:
            raise last
    finally:
        del os.environ["ALVA_CHECK_OBS_MIRROR"]
    assert statuses[-1] == 1 and cnt["culled_keyframes"] >= 1 and cnt["ba_solves"] >= 25, cnt


def test_explicit_reset_between_frames():
    """System::reset called by the host in the middle of a tracked stream (system.cpp:42-55: frame, map, counters cleared; the motion
    model, p3pReq_ and the detector's adaptive threshold survive): the stream re-initialises, and everything after the reset is compared
    like everything before it"""
    w, h = 640, 480
    canvas = synth.texture_canvas(w, h, 7)
    frames = (synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(110))
    statuses, cnt, _, _ = _run(frames, w, h, 40, 0, reset_at=(48,))
    assert statuses[47] == 1 and statuses[48] == 3 and statuses[-1] == 1, (statuses[44:52], statuses[-5:])


def test_stage_result_tape_replays_to_the_same_state():
    """tools/host_replay_cpu.py's instrument (oracle/sys_cpu.cpp MemoStages): a second map layer fed the RECORDED results of every stage
    call of a first one must pass through the same states -- status, counters, the frame's keypoints in container order, the map-point
    table with its medoids -- which also says that the map layer's behaviour is a function of what its stages return and nothing else"""
    import ctypes as C
    w, h, n = 640, 480, 90
    canvas = synth.texture_canvas(w, h, 7)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(n)]
    rec, rep = sysdiff.CpuSystem(w, h, 25), sysdiff.CpuSystem(w, h, 25)
    L = rec.L
    L.syscpu_tape_new.restype = C.c_void_p
    L.syscpu_tape_bytes.restype = C.c_longlong
    L.syscpu_tape_bytes.argtypes = [C.c_void_p]
    L.syscpu_attach_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.syscpu_tape_free.argtypes = [C.c_void_p]
    tape = C.c_void_p(L.syscpu_tape_new())
    try:
        L.syscpu_attach_tape(rec.h, tape, 0)
        seen = []
        for k in range(n):
            st, pose, _ = rec.step(frames[k], 33.0 * k)
            seen.append((st, pose.copy(), [int(v) for v in rec.state()], [a.copy() for a in rec.frame_keypoints()]))
        assert L.syscpu_tape_bytes(tape) > 1000000 and any(s[0] == 1 for s in seen)
        L.syscpu_attach_tape(rep.h, tape, 1)
        for k in range(n):
            st, pose, _ = rep.step(frames[k], 33.0 * k)
            assert st == seen[k][0] and np.array_equal(pose, seen[k][1]), k
            assert [int(v) for v in rep.state()] == seen[k][2], k
            for a, b in zip(rep.frame_keypoints(), seen[k][3]):
                assert np.array_equal(a, b), k
        sysdiff.compare(rec, rep, 0.0, what="after the replay")
    finally:
        rec.close()
        rep.close()
        L.syscpu_tape_free(tape)
