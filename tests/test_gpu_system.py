"""GPU: the reference's System surface (a1) driven like the JS wrapper drives it.
Scene: the synthetic texture is a fronto-parallel plane at depth Z; the canvas crop moves (2, 1) px per frame, which is
exactly a camera translation of (2 Z / f, Z / f, 0) per frame with no rotation.  Map points are attached once (the
mapper rows that would triangulate them are SURVEY.md §8f "next"); from then on every pose comes from the GPU hot loop
gray -> pyramid -> fb-KLT -> P3P-LMedS -> PnP behind findCameraPose."""
import numpy as np
import pytest

from alvaar_amd import synth

pytestmark = pytest.mark.gpu


def test_find_camera_pose_tracks_a_translating_camera():
    from alvaar_amd.system import AlvaAR
    w, h, Z = 640, 480, 4.0
    ar = AlvaAR.Initialize(w, h)                      # src/system.js:47-56
    K = ar.intrinsics
    canvas = synth.texture_canvas(w, h, 7)
    frame = lambda k: synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
    pose, status = ar.findCameraPose(frame(0))
    assert status == 3 and pose is None               # initialising: keypoints extracted, no map yet
    pts2d = ar.getFramePoints()
    assert 50 < len(pts2d) <= 2048 and all(20 <= p["x"] < w - 20 + 3 for p in pts2d)
    ids, px, is3d = ar.keypoints()
    assert len(ids) == len(pts2d) and not is3d.any()
    X = np.stack([(px[:, 0] - K["cx"]) / K["fx"] * Z, (px[:, 1] - K["cy"]) / K["fy"] * Z, np.full(len(px), Z)], 1)
    assert ar.set_map_points(ids, X) == len(ids)
    assert len(ar.getFramePoints()) == 0              # getFramePoints reports 2-D (untriangulated) keypoints only
    errs = []
    for k in range(1, 13):
        pose, status = ar.findCameraPose(frame(k))
        assert status == 1 and pose is not None, k
        R = pose.reshape(4, 4)[:3, :3]
        t = pose[12:15]
        assert pose[15] == 1.0 and pose[3] == pose[7] == pose[11] == 0.0
        assert np.abs(R - np.eye(3)).max() < 2e-3
        expect = np.array([2 * k * Z / K["fx"], k * Z / K["fy"], 0.0])
        errs.append(np.abs(t - expect).max())
    assert max(errs) < 0.02, errs                     # KLT is sub-pixel; 1 px = Z / f = 0.0069 m here
    ar.reset()
    pose, status = ar.findCameraPose(frame(0))
    assert status == 3
    ar.close()


def test_imu_variant_and_plane_conventions():
    from alvaar_amd.system import AlvaAR
    ar = AlvaAR.Initialize(640, 480)
    rgba = synth.gray_to_rgba(synth.frame_gray(synth.texture_canvas(640, 480, 7), 0, 640, 480))
    pose = ar.findCameraPoseWithIMU(rgba, (1.0, 0.0, 0.0, 0.0))     # always status 1 (system.cpp:103)
    assert pose is not None and np.allclose(pose.reshape(4, 4)[:3, :3], np.eye(3)) and pose[15] == 1.0
    assert ar.findPlane() is None                                     # < 32 observed 3-D points -> 0 (system.cpp:181)
    ar.close()


def test_find_plane_after_cold_start():
    """findPlane on the map the cold start built: the scene IS a fronto-parallel plane, so the plane normal must come out along the
    world z axis and the origin inside the triangulated points (the plane fit is the intended processPlane, parity unpinned)."""
    from alvaar_amd.system import AlvaAR
    w, h = 640, 480
    ar = AlvaAR.Initialize(w, h)
    canvas = synth.texture_canvas(w, h, 7)
    for k in range(0, 30):
        pose, status = ar.findCameraPose(synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)))
    assert status == 1
    plane = ar.findPlane()
    assert plane is not None and plane[15] == 1.0
    R = plane.reshape(4, 4)[:3, :3].T
    Rx = np.array([[1, 0, 0], [0, np.cos(1.0), -np.sin(1.0)], [0, np.sin(1.0), np.cos(1.0)]])
    n = (R @ Rx.T)[:, 0]
    assert abs(abs(n[2]) - 1.0) < 0.05, n                              # clock-seeded sampling, as in the reference (system.cpp:203)
    assert plane[14] > 0.5                                             # in front of the first camera (unit-baseline scale)
    ar.close()


def test_cold_start_initialises_its_own_map():
    """No host-fed map: keyframe 0, parallax gate, 5-point initialisation (unit baseline), triangulation of keyframe 1, then
    P3P + PnP tracking and new keyframes by the reference's policy -- the path checkReadyForInit -> createKeyframe ->
    triangulateTemporal -> computePose of the reference, on the same fronto-parallel scene as above."""
    from alvaar_amd.system import AlvaAR
    w, h = 640, 480
    ar = AlvaAR.Initialize(w, h)
    canvas = synth.texture_canvas(w, h, 7)
    frame = lambda k: synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h))
    statuses, poses = [], {}
    for k in range(0, 60):
        pose, status = ar.findCameraPose(frame(k))
        statuses.append(status)
        if status == 1:
            poses[k] = pose
        assert status in (1, 3), (k, status)
    k0 = statuses.index(1)
    assert statuses[:k0] == [3] * k0 and all(s == 1 for s in statuses[k0:])
    assert 17 <= k0 <= 22                                   # (2, 1) px per frame => > 40 px of parallax after 18 frames
    d = np.array([2.0, 1.0, 0.0]) / np.sqrt(5.0)
    t0 = poses[k0][12:15]
    assert abs(np.linalg.norm(t0) - 1.0) < 1e-6             # twc.normalize() (visual_frontend.cpp:547)
    assert np.abs(t0 - d).max() < 0.03                      # direction of the true translation
    ids, px, is3d = ar.keypoints()
    assert is3d.sum() >= 30                                 # the initial map (mapper.cpp:29: fewer than 30 would reset)
    for k, pose in poses.items():
        R, t = pose.reshape(4, 4)[:3, :3], pose[12:15]
        assert np.abs(R - np.eye(3)).max() < 0.02, k
        assert np.abs(t - d * k / k0).max() < 0.06 * k / k0, (k, t)     # the map's scale is the first baseline
    ar.close()
