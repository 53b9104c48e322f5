"""TEST INFRASTRUCTURE: one interface over three implementations of the reference's System surface, for differential tests.

  RefSystem  the reference's own System (oracle/_ref/libalva_ref.so: ref_system_*, oracle/ref_shim_system.cpp)
  CpuSystem  the product's host-side map layer over the reference's L1 stages (syscpu_*, oracle/sys_cpu.cpp) -- host logic, no GPU
  GpuSystem  the product: alva_system_* (host-side map layer over the HIP stages)
All three take explicit timestamps and fixed-seed sampling; `compare` checks everything the map layers hold."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

import oracles

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
CAP_KP, CAP_MP = 16384, 65536


def intrinsics(w, h, fov=45.0):
    """AlvaAR.getCameraIntrinsics (src/system.js:84-141) -- the same rule alvaar_amd.system.AlvaAR applies"""
    from alvaar_amd.system import camera_intrinsics
    k = camera_intrinsics(w, h, fov)
    return k["fx"], k["fy"], k["cx"], k["cy"]


class _ShimSystem:
    prefix = ""

    def __init__(self, w, h, cell=40, clahe=False, dist=(0.0, 0.0, 0.0, 0.0)):
        L = oracles.ref_lib()
        self.L = L
        p = self.prefix
        getattr(L, p + "_create").restype = _vp
        getattr(L, p + "_create").argtypes = [_i] * 2 + [_d] * 8 + [_i] * 3
        getattr(L, p + "_destroy").argtypes = [_vp]
        getattr(L, p + "_find_camera_pose").argtypes = [_vp, _vp, _d, _vp, _vp]
        getattr(L, p + "_state").argtypes = [_vp, _vp]
        getattr(L, p + "_frame_keypoints").argtypes = [_vp, _i] + [_vp] * 5
        getattr(L, p + "_keyframe_ids").argtypes = [_vp, _i, _vp]
        getattr(L, p + "_keyframe").argtypes = [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]
        getattr(L, p + "_covisibility").argtypes = [_vp, _i, _i, _vp]
        getattr(L, p + "_map_points").argtypes = [_vp, _i] + [_vp] * 5
        L.ref_freeze_clock(1)  # Ceres' wall-clock caps never fire
        fx, fy, cx, cy = intrinsics(w, h)
        self.K = (fx, fy, cx, cy)
        self.h = getattr(L, p + "_create")(w, h, fx, fy, cx, cy, *map(float, dist), cell, int(clahe), 0)

    def close(self):
        if self.h:
            getattr(self.L, self.prefix + "_destroy")(self.h)
            self.h = None

    def reset(self):
        """System::reset (system.cpp:42-55)"""
        getattr(self.L, self.prefix + "_reset").argtypes = [_vp]
        getattr(self.L, self.prefix + "_reset")(self.h)

    def step(self, rgba, ts):
        pose16, pose7 = np.zeros(16, np.float32), np.zeros(7)
        rgba = np.ascontiguousarray(rgba)
        st = getattr(self.L, self.prefix + "_find_camera_pose")(self.h, rgba.ctypes.data, float(ts), pose16.ctypes.data, pose7.ctypes.data)
        return st, pose7, pose16

    def state(self):
        out = np.zeros(16, np.int32)
        getattr(self.L, self.prefix + "_state")(self.h, out.ctypes.data)
        return out

    def frame_keypoints(self):
        ids, px, un = np.zeros(CAP_KP, np.int32), np.zeros((CAP_KP, 2), np.float32), np.zeros((CAP_KP, 2), np.float32)
        i3, hd = np.zeros(CAP_KP, np.uint8), np.zeros(CAP_KP, np.uint8)
        n = getattr(self.L, self.prefix + "_frame_keypoints")(self.h, CAP_KP, ids.ctypes.data, px.ctypes.data, un.ctypes.data, i3.ctypes.data,
                                                               hd.ctypes.data)
        return ids[:n], px[:n], un[:n], i3[:n], hd[:n]

    def keyframe_ids(self):
        ids = np.zeros(256, np.int32)
        n = getattr(self.L, self.prefix + "_keyframe_ids")(self.h, 256, ids.ctypes.data)
        return ids[:n]

    def keyframe(self, kfid):
        pose, info = np.zeros(7), np.zeros(6, np.int32)
        ids, px, i3 = np.zeros(CAP_KP, np.int32), np.zeros((CAP_KP, 2), np.float32), np.zeros(CAP_KP, np.uint8)
        n = getattr(self.L, self.prefix + "_keyframe")(self.h, int(kfid), pose.ctypes.data, info.ctypes.data, CAP_KP, ids.ctypes.data,
                                                        px.ctypes.data, i3.ctypes.data)
        return pose, info, ids[:n], px[:n], i3[:n]

    def covisibility(self, kfid=-1):
        pairs = np.zeros((256, 2), np.int32)
        n = getattr(self.L, self.prefix + "_covisibility")(self.h, int(kfid), 256, pairs.ctypes.data)
        return pairs[:max(n, 0)]

    def map_points(self):
        ids, xyz, fl = np.zeros(CAP_MP, np.int32), np.zeros((CAP_MP, 3)), np.zeros((CAP_MP, 5), np.int32)
        inv, desc = np.zeros(CAP_MP), np.zeros((CAP_MP, 32), np.uint8)
        n = getattr(self.L, self.prefix + "_map_points")(self.h, CAP_MP, ids.ctypes.data, xyz.ctypes.data, fl.ctypes.data, inv.ctypes.data,
                                                          desc.ctypes.data)
        return ids[:n], xyz[:n], fl[:n], inv[:n], desc[:n]


class RefSystem(_ShimSystem):
    prefix = "ref_system"


class CpuSystem(_ShimSystem):
    prefix = "syscpu"

    def set_init_pose(self, pose7):
        self.L.syscpu_set_init_pose.argtypes = [_vp, _vp]
        p = None if pose7 is None else np.ascontiguousarray(pose7, np.float64)
        self.L.syscpu_set_init_pose(self.h, None if p is None else p.ctypes.data)

    def counters(self):
        out = (C.c_long * 3)()
        self.L.syscpu_counters.argtypes = [_vp, _vp]
        self.L.syscpu_counters(self.h, out)
        return dict(ba_solves=out[0], merges=out[1], culled_keyframes=out[2])


class GpuSystem:
    """alva_system_* through alvaar_amd.system.AlvaAR (fixed-seed sampling, explicit timestamps)"""

    def __init__(self, w, h, cell=40, clahe=False, dist=(0.0, 0.0, 0.0, 0.0)):
        from alvaar_amd.system import AlvaAR
        self.ar = AlvaAR(w, h, cell_size=cell, clahe=clahe, random_sampling=False, distortion=dist)

    def close(self):
        self.ar.close()

    def reset(self):
        self.ar.reset()

    def step(self, rgba, ts):
        pose, st = self.ar.findCameraPose(rgba, ts)
        return st, self.ar.pose7()[0], self.ar._pose.copy()

    def __getattr__(self, name):
        return getattr(self.ar, name)


def compare(a, b, pose_tol, px_exact=True, xyz_tol=None, what=""):
    """everything two map layers hold after a frame: state counters, the frame's keypoints IN CONTAINER ORDER, keyframe ids and the
    map point table.  Returns the largest pose / map point differences seen."""
    sa, sb = a.state(), b.state()
    assert list(sa) == list(sb), f"{what}: state {list(sa)} != {list(sb)}"
    ka, kb = a.frame_keypoints(), b.frame_keypoints()
    assert np.array_equal(ka[0], kb[0]), f"{what}: keypoint ids / container order differ"
    assert np.array_equal(ka[3], kb[3]) and np.array_equal(ka[4], kb[4]), f"{what}: keypoint flags differ"
    if px_exact:
        assert np.array_equal(ka[1].view(np.uint32), kb[1].view(np.uint32)), f"{what}: keypoint pixels differ (max {np.abs(ka[1] - kb[1]).max()})"
        assert np.array_equal(ka[2].view(np.uint32), kb[2].view(np.uint32)), f"{what}: undistorted pixels differ"
    else:
        assert np.abs(ka[1] - kb[1]).max(initial=0) < 1e-2, f"{what}: keypoint pixels differ (max {np.abs(ka[1] - kb[1]).max()})"
    assert np.array_equal(a.keyframe_ids(), b.keyframe_ids()), f"{what}: keyframe ids differ"
    ma, mb = a.map_points(), b.map_points()
    assert np.array_equal(ma[0], mb[0]), f"{what}: map point ids differ"
    assert np.array_equal(ma[2], mb[2]), f"{what}: map point flags (3-D, observed, #observers, anchor, #descriptors) differ"
    assert np.array_equal(ma[4], mb[4]), f"{what}: map point descriptors (medoids) differ"
    dx = float(np.abs(ma[1] - mb[1]).max(initial=0))
    if xyz_tol is not None:
        assert dx <= xyz_tol, f"{what}: map points differ by {dx}"
    return dx


def compare_keyframes(a, b, pose_tol, what=""):
    worst = 0.0
    for kf in a.keyframe_ids():
        pa, ia, idsa, pxa, i3a = a.keyframe(kf)
        pb, ib, idsb, pxb, i3b = b.keyframe(kf)
        assert list(ia) == list(ib), f"{what}: keyframe {kf} info {list(ia)} != {list(ib)}"
        assert np.array_equal(idsa, idsb) and np.array_equal(i3a, i3b), f"{what}: keyframe {kf} keypoint set / order differs"
        assert np.array_equal(a.covisibility(kf), b.covisibility(kf)), f"{what}: keyframe {kf} covisibility differs"
        q = pb[3:] if np.dot(pa[3:], pb[3:]) >= 0 else -pb[3:]
        d = max(np.abs(pa[:3] - pb[:3]).max(), np.abs(pa[3:] - q).max())
        assert d <= pose_tol, f"{what}: keyframe {kf} pose differs by {d}"
        worst = max(worst, d)
    return worst


def pose_diff(pa, pb):
    q = pb[3:] if np.dot(pa[3:], pb[3:]) >= 0 else -pb[3:]
    return max(np.abs(pa[:3] - pb[:3]).max(), np.abs(pa[3:] - q).max())


def _quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def sim3_aligned_diff(ref_poses7, got_poses7):
    """Two trajectories of the same camera in two maps whose gauges (scale, world frame) may differ: align `got` to `ref` with the
    similarity transform of the gauge (rotation = chordal mean of the relative orientations, scale and translation by least squares on
    the camera centres), then report
      (scale, worst |centre difference| / trajectory extent, worst rotation difference in radians)
    of the aligned trajectory.  pose7 = (t, q_xyzw) of Twc."""
    A = np.array([p[:3] for p in got_poses7], np.float64)
    B = np.array([p[:3] for p in ref_poses7], np.float64)
    ma, mb = A.mean(0), B.mean(0)
    Ac, Bc = A - ma, B - mb
    # rotation of the gauge from the ORIENTATIONS (chordal mean of Rr_i Rg_i^T): the centres alone do not fix it when the camera moves
    # along a line (the synthetic streams do: a rotation about that line leaves every centre where it is)
    M = sum(_quat_to_rot(pr[3:]) @ _quat_to_rot(pg[3:]).T for pr, pg in zip(ref_poses7, got_poses7))
    U, S, Vt = np.linalg.svd(M)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    RA = (R @ Ac.T).T
    den = float((RA ** 2).sum())
    c = float((Bc * RA).sum() / den) if den > 0 else 1.0
    t = mb - c * R @ ma
    extent = float(np.linalg.norm(Bc, axis=1).max()) or 1.0
    dpos = float(np.linalg.norm((c * (R @ A.T).T + t) - B, axis=1).max()) / extent
    drot = 0.0
    for pr, pg in zip(ref_poses7, got_poses7):
        Rr, Rg = _quat_to_rot(pr[3:]), R @ _quat_to_rot(pg[3:])
        drot = max(drot, float(np.arccos(np.clip((np.trace(Rr.T @ Rg) - 1) / 2, -1, 1))))
    return c, dpos, drot
