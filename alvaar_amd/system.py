"""Python mirror of the reference's JS wrapper class `AlvaAR` (src/system.js:45-238) over the alva_system_* C ABI
(include/alvaar_system.h): same method names, same intrinsics-from-FOV rule, same return conventions
(pose array or None).  It is the host-side caller of the drop-in boundary, used by tests."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .capi import lib, AlvaError

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
lib.alva_system_create.argtypes = [_i, C.POINTER(_vp)]
lib.alva_system_destroy.argtypes = [_vp]
lib.alva_system_destroy.restype = None
lib.alva_system_configure.argtypes = [_vp, _i, _i] + [_d] * 8
lib.alva_system_reset.argtypes = [_vp]
lib.alva_system_reset.restype = None
lib.alva_system_find_camera_pose.argtypes = [_vp, _vp, _vp]
lib.alva_system_find_camera_pose_with_imu.argtypes = [_vp, _vp, _vp, _vp]
lib.alva_system_find_plane.argtypes = [_vp, _vp, _i]
lib.alva_system_get_frame_points.argtypes = [_vp, _vp]
lib.alva_system_get_keypoints.argtypes = [_vp, _vp, _vp, _vp, _i]
lib.alva_system_set_map_points.argtypes = [_vp, _vp, _vp, _i]
lib.alva_system_set_pose.argtypes = [_vp, _vp]
lib.alva_system_last_error.restype = C.c_char_p


def camera_intrinsics(width: int, height: int, fov: float = 45.0):
    """AlvaAR.getCameraIntrinsics (src/system.js:84-141)"""
    aspect = width / height
    fov_h, fov_v = (fov * aspect, fov) if width > height else (fov, fov * aspect)
    deg2rad = 0.01745329251994329576
    fx = (width * 0.5) / math.tan(fov_h * 0.5 * deg2rad)
    fy = (height * 0.5) / math.tan(fov_v * 0.5 * deg2rad)
    f = min(fx, fy)
    return dict(width=width, height=height, fx=f, fy=f, cx=width * 0.5, cy=height * 0.5, k1=0.0, k2=0.0, p1=0.0, p2=0.0)


class AlvaAR:
    def __init__(self, width: int, height: int, fov: float = 45.0, device: int = 0):
        self.intrinsics = camera_intrinsics(width, height, fov)
        h = _vp()
        rc = lib.alva_system_create(device, C.byref(h))
        if rc:
            raise AlvaError(lib.alva_system_last_error().decode())
        self.h = h
        k = self.intrinsics
        rc = lib.alva_system_configure(h, width, height, k["fx"], k["fy"], k["cx"], k["cy"], k["k1"], k["k2"], k["p1"], k["p2"])
        if rc:
            raise AlvaError(lib.alva_system_last_error().decode())
        self._pose = np.zeros(16, np.float32)

    @staticmethod
    def Initialize(width: int, height: int, fov: float = 45.0) -> "AlvaAR":  # noqa: N802 (reference name)
        return AlvaAR(width, height, fov)

    def close(self):
        if getattr(self, "h", None):
            lib.alva_system_destroy(self.h)
            self.h = None

    __del__ = close

    def findCameraPose(self, frame_rgba: np.ndarray):  # noqa: N802
        """returns (pose[16] or None, status) -- the JS wrapper returns the pose only on status 1"""
        frame = np.ascontiguousarray(frame_rgba, np.uint8)
        status = lib.alva_system_find_camera_pose(self.h, frame.ctypes.data, self._pose.ctypes.data)
        if status < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return (self._pose.copy() if status == 1 else None), status

    def findCameraPoseWithIMU(self, frame_rgba, orientation_wxyz, motion=()):  # noqa: N802
        frame = np.ascontiguousarray(frame_rgba, np.uint8)
        imu = np.zeros(256, np.float64)
        imu[:4] = orientation_wxyz
        imu[4] = len(motion)
        for k, smp in enumerate(motion):
            imu[5 + 7 * k:12 + 7 * k] = smp
        status = lib.alva_system_find_camera_pose_with_imu(self.h, frame.ctypes.data, imu.ctypes.data, self._pose.ctypes.data)
        return self._pose.copy() if status == 1 else None

    def findPlane(self, num_iterations: int = 250):  # noqa: N802
        out = np.zeros(16, np.float32)
        return out if lib.alva_system_find_plane(self.h, out.ctypes.data, num_iterations) == 1 else None

    def getFramePoints(self):  # noqa: N802
        buf = np.zeros(4096, np.int32)
        n = lib.alva_system_get_frame_points(self.h, buf.ctypes.data)
        m = min(n, 2048)
        return [dict(x=int(buf[2 * i]), y=int(buf[2 * i + 1])) for i in range(m)]

    def reset(self):
        lib.alva_system_reset(self.h)

    # bootstrap helpers (until the mapper rows of SURVEY.md §8f are built)
    def keypoints(self, cap: int = 8192):
        ids = np.zeros(cap, np.int32)
        px = np.zeros((cap, 2), np.float32)
        is3d = np.zeros(cap, np.uint8)
        n = min(lib.alva_system_get_keypoints(self.h, ids.ctypes.data, px.ctypes.data, is3d.ctypes.data, cap), cap)
        return ids[:n], px[:n], is3d[:n].astype(bool)

    def set_map_points(self, ids, xyz):
        ids = np.ascontiguousarray(ids, np.int32)
        xyz = np.ascontiguousarray(xyz, np.float64)
        return lib.alva_system_set_map_points(self.h, ids.ctypes.data, xyz.ctypes.data, len(ids))
