"""Python mirror of the reference's JS wrapper class `AlvaAR` (src/system.js:45-238) over the alva_system_* C ABI
(include/alvaar_system.h): same method names, same intrinsics-from-FOV rule, same return conventions
(pose array or None).  It is the host-side caller of the drop-in boundary, used by tests."""
from __future__ import annotations

import ctypes as C
import math
import os

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
lib.alva_system_find_camera_pose_with_imu_ts.argtypes = [_vp, _vp, _vp, C.c_double, _vp]
lib.alva_system_find_plane.argtypes = [_vp, _vp, _i]
lib.alva_system_get_frame_points.argtypes = [_vp, _vp]
lib.alva_system_get_keypoints.argtypes = [_vp, _vp, _vp, _vp, _i]
lib.alva_system_configure_ex.argtypes = [_vp, _i, _i] + [_d] * 8 + [_i] * 3
lib.alva_system_find_camera_pose_ts.argtypes = [_vp, _vp, _d, _vp]
lib.alva_system_register_frame_buffer.argtypes = [_vp, _vp, C.c_size_t]
lib.alva_system_unregister_frame_buffer.argtypes = [_vp]
if hasattr(lib, "alva_system_alloc_frame_buffer"):
    lib.alva_system_alloc_frame_buffer.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
lib.alva_system_find_camera_pose_device.argtypes = [_vp, _vp, _d, _vp]
lib.alva_system_hint_next_frame_device.argtypes = [_vp, _vp]
lib.alva_system_debug_state.argtypes = [_vp, _vp]
lib.alva_system_debug_pose7.argtypes = [_vp, _vp, _vp]
lib.alva_system_debug_frame_keypoints.argtypes = [_vp, _i] + [_vp] * 5
lib.alva_system_debug_keyframe_ids.argtypes = [_vp, _i, _vp]
lib.alva_system_debug_keyframe.argtypes = [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]
lib.alva_system_debug_covisibility.argtypes = [_vp, _i, _i, _vp]
lib.alva_system_debug_map_points.argtypes = [_vp, _i] + [_vp] * 5
lib.alva_system_debug_counters.argtypes = [_vp, _vp]
lib.alva_system_merge_map_points.argtypes = [_vp, _i, _i]
lib.alva_system_set_shared_ids.argtypes = [_vp, _i, _vp, _vp, _vp]
lib.alva_system_get_shared_ids.argtypes = [_vp, _i, _vp, _vp, _vp]
lib.alva_system_debug_klt_work.argtypes = [_vp, _vp, _i]
lib.alva_system_debug_set_init_pose.argtypes = [_vp, _vp]
lib.alva_system_debug_timing.argtypes = [_vp, _vp, _i]
lib.alva_system_debug_timing_keyframe.argtypes = [_vp, _vp, _i]
lib.alva_system_debug_timing_fine.argtypes = [_vp, _vp, _i]
lib.alva_system_last_error.restype = C.c_char_p
lib.alva_system_group_create.argtypes = [_i, C.POINTER(_vp)]
lib.alva_system_group_destroy.argtypes = [_vp]
lib.alva_system_group_destroy.restype = None
lib.alva_system_group_find_camera_pose_device.argtypes = [_vp, _i, _vp, _vp, _d, _vp, _vp]
lib.alva_system_group_stream.argtypes = [_vp, _i, _i, C.POINTER(_vp)]
lib.alva_system_set_stream.argtypes = [_vp, _vp]
lib.alva_system_group_set_lockstep.argtypes = [_vp, _i]
lib.alva_system_group_launch_stats.argtypes = [_vp, _vp]
lib.alva_system_group_set_lanes.argtypes = [_vp, _i]
lib.alva_system_group_time_stats.argtypes = [_vp, _vp, _i]


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
    def __init__(self, width: int, height: int, fov: float = 45.0, device: int = 0, cell_size: int | None = None, clahe: bool = False,
                 random_sampling: bool = True, distortion=(0.0, 0.0, 0.0, 0.0), hip_stream=None):
        """cell_size / clahe / random_sampling: the settings System::configure hard-codes (system.cpp:15-19, state.hpp:67);
        None = the shipped configuration through alva_system_configure."""
        self.intrinsics = camera_intrinsics(width, height, fov)
        self.intrinsics.update(dict(zip(("k1", "k2", "p1", "p2"), map(float, distortion))))
        h = _vp()
        rc = lib.alva_system_create(device, C.byref(h))
        if rc:
            raise AlvaError(lib.alva_system_last_error().decode())
        self.h = h
        if hip_stream is not None:   # a stream shared with other sessions (SystemGroup.stream)
            lib.alva_system_set_stream(h, hip_stream)
        k = self.intrinsics
        if cell_size is None and not clahe and random_sampling:
            rc = lib.alva_system_configure(h, width, height, k["fx"], k["fy"], k["cx"], k["cy"], k["k1"], k["k2"], k["p1"], k["p2"])
        else:
            rc = lib.alva_system_configure_ex(h, width, height, k["fx"], k["fy"], k["cx"], k["cy"], k["k1"], k["k2"], k["p1"], k["p2"],
                                              cell_size or 40, int(clahe), int(random_sampling))
        if rc:
            msg = lib.alva_system_last_error().decode()
            lib.alva_system_destroy(h)
            self.h = None
            raise AlvaError(msg)
        self._pose = np.zeros(16, np.float32)
        self._pose_ptr = self._pose.ctypes.data   # (ndarray.ctypes builds a helper object per access: ~1 us of every per-frame call)
        self._fcp_device = lib.alva_system_find_camera_pose_device
        # src/system.js:63-67: ONE frame buffer (memImg) allocated at construction and reused for every frame; here it is page-locked and
        # mapped once (alva_system_register_frame_buffer) so that the gray / pyramid kernel reads it in place.  4096-byte aligned.
        # Round 5: the buffer lives in DEVICE memory that the host writes over the PCIe BAR (alva_system_alloc_frame_buffer): memImg.write IS
        # the upload and the kernels read the frame out of HBM.  Never read mem_img back (a load crosses the bus).  ALVA_NO_BAR_FRAME=1 or a
        # system without such memory: the page-locked host buffer of rounds 3 - 4.
        nbytes = width * height * 4
        self.mem_img = None
        self._bar_frame = False
        if not os.environ.get("ALVA_NO_BAR_FRAME") and hasattr(lib, "alva_system_alloc_frame_buffer"):
            p = _vp()
            if lib.alva_system_alloc_frame_buffer(h, nbytes, C.byref(p)) == 0 and p.value:
                self.mem_img = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).reshape(height, width, 4)
                self._bar_frame = True
        if self.mem_img is None:
            self._mem_raw = np.empty(nbytes + 4096, np.uint8)
            off = (-self._mem_raw.ctypes.data) % 4096
            self.mem_img = self._mem_raw[off:off + nbytes].reshape(height, width, 4)
        self._registered = self._bar_frame or lib.alva_system_register_frame_buffer(h, self.mem_img.ctypes.data, nbytes) == 0
        self._mem_ptr = self.mem_img.ctypes.data

    @staticmethod
    def Initialize(width: int, height: int, fov: float = 45.0) -> "AlvaAR":  # noqa: N802 (reference name)
        return AlvaAR(width, height, fov)

    def close(self):
        if getattr(self, "h", None):
            lib.alva_system_destroy(self.h)   # releases the frame-buffer registration before mem_img goes away
            self.h = None

    __del__ = close

    def findCameraPose(self, frame_rgba: np.ndarray, timestamp_ms: float | None = None):  # noqa: N802
        """returns (pose[16] or None, status) -- the JS wrapper returns the pose only on status 1.  timestamp_ms = None reads the
        system clock like the reference (system.cpp:114)."""
        self._stage(frame_rgba)
        if timestamp_ms is None:
            status = lib.alva_system_find_camera_pose(self.h, self._mem_ptr, self._pose_ptr)
        else:
            status = lib.alva_system_find_camera_pose_ts(self.h, self._mem_ptr, timestamp_ms, self._pose_ptr)
        if status < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return (self._pose.copy() if status == 1 else None), status

    def _stage(self, frame_rgba) -> np.ndarray:
        """memImg.write(frame.data) (src/system.js:175): the frame goes into the wrapper's own registered buffer -- unless the caller
        already wrote it there (passes self.mem_img itself)."""
        if frame_rgba is self.mem_img:
            return self.mem_img
        np.copyto(self.mem_img, np.asarray(frame_rgba, np.uint8).reshape(self.mem_img.shape))
        return self.mem_img

    def find_camera_pose_device(self, d_rgba_ptr: int, timestamp_ms: float, next_d_rgba_ptr: int | None = None):
        """frame already in device memory (torch tensor .data_ptr()); returns the status, the pose is in self._pose.
        next_d_rgba_ptr: the frame the NEXT call will pass (alva_system_hint_next_frame_device: its pyramid is built ahead)"""
        if next_d_rgba_ptr:
            lib.alva_system_hint_next_frame_device(self.h, next_d_rgba_ptr)
        status = self._fcp_device(self.h, d_rgba_ptr, timestamp_ms, self._pose_ptr)
        if status < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return status

    def findCameraPoseWithIMU(self, frame_rgba, orientation_wxyz, motion=(), timestamp_ms: float | None = None):  # noqa: N802
        frame = self._stage(frame_rgba)
        imu = np.zeros(256, np.float64)
        imu[:4] = orientation_wxyz
        imu[4] = len(motion)
        for k, smp in enumerate(motion):
            imu[5 + 7 * k:12 + 7 * k] = smp
        if timestamp_ms is None:
            status = lib.alva_system_find_camera_pose_with_imu(self.h, frame.ctypes.data, imu.ctypes.data, self._pose.ctypes.data)
        else:
            status = lib.alva_system_find_camera_pose_with_imu_ts(self.h, frame.ctypes.data, imu.ctypes.data, float(timestamp_ms), self._pose.ctypes.data)
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

    def keypoints(self, cap: int = 16384):
        ids = np.zeros(cap, np.int32)
        px = np.zeros((cap, 2), np.float32)
        is3d = np.zeros(cap, np.uint8)
        n = min(lib.alva_system_get_keypoints(self.h, ids.ctypes.data, px.ctypes.data, is3d.ctypes.data, cap), cap)
        return ids[:n], px[:n], is3d[:n].astype(bool)

    # ---- inspection (alva_system_debug_*), same layouts as the reference-side shim of the tests
    def pose7(self):
        p, q = np.zeros(7), np.zeros(7)
        lib.alva_system_debug_pose7(self.h, p.ctypes.data, q.ctypes.data)
        return p, q

    def state(self):
        out = np.zeros(16, np.int32)
        lib.alva_system_debug_state(self.h, out.ctypes.data)
        return out

    def frame_keypoints(self, cap: int = 16384):
        ids, px, un = np.zeros(cap, np.int32), np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        i3, hd = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        n = lib.alva_system_debug_frame_keypoints(self.h, cap, ids.ctypes.data, px.ctypes.data, un.ctypes.data, i3.ctypes.data, hd.ctypes.data)
        return ids[:n], px[:n], un[:n], i3[:n], hd[:n]

    def keyframe_ids(self, cap: int = 256):
        ids = np.zeros(cap, np.int32)
        n = lib.alva_system_debug_keyframe_ids(self.h, cap, ids.ctypes.data)
        return ids[:n]

    def keyframe(self, kfid: int, cap: int = 16384):
        pose, info = np.zeros(7), np.zeros(6, np.int32)
        ids, px, i3 = np.zeros(cap, np.int32), np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint8)
        n = lib.alva_system_debug_keyframe(self.h, kfid, pose.ctypes.data, info.ctypes.data, cap, ids.ctypes.data, px.ctypes.data, i3.ctypes.data)
        return pose, info, ids[:n], px[:n], i3[:n]

    def covisibility(self, kfid: int = -1, cap: int = 256):
        pairs = np.zeros((cap, 2), np.int32)
        n = lib.alva_system_debug_covisibility(self.h, kfid, cap, pairs.ctypes.data)
        return pairs[:max(n, 0)]

    def map_points(self, cap: int = 65536):
        ids, xyz, fl = np.zeros(cap, np.int32), np.zeros((cap, 3)), np.zeros((cap, 5), np.int32)
        inv, desc = np.zeros(cap), np.zeros((cap, 32), np.uint8)
        n = lib.alva_system_debug_map_points(self.h, cap, ids.ctypes.data, xyz.ctypes.data, fl.ctypes.data, inv.ctypes.data, desc.ctypes.data)
        return ids[:n], xyz[:n], fl[:n], inv[:n], desc[:n]

    def pack_map_records(self, stream_id: int, capacity: int, out):
        """alva_system_pack_map_records: this session's 3-D map points as 64-byte exchange records written on the device into `out`
        (cuda uint8 tensor [capacity, 64]); returns the number of points found (may exceed capacity: then `out` holds a subset)"""
        n = C.c_int(0)
        lib.alva_system_pack_map_records.argtypes = [_vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)]
        rc = lib.alva_system_pack_map_records(self.h, int(stream_id), int(capacity), out.data_ptr(), C.byref(n))
        if rc < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return n.value

    def merge_map_points(self, prev_id: int, new_id: int) -> bool:
        """MapManager::mergeMapPoints(prev_id, new_id) on this session's map (include/alvaar_system.h); False = the reference's early return"""
        rc = lib.alva_system_merge_map_points(self.h, int(prev_id), int(new_id))
        if rc < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return rc == 1

    def set_shared_ids(self, local_ids, shared_stream, shared_id) -> int:
        a, b, c = (np.ascontiguousarray(v, np.int32) for v in (local_ids, shared_stream, shared_id))
        rc = lib.alva_system_set_shared_ids(self.h, len(a), a.ctypes.data, b.ctypes.data, c.ctypes.data)
        if rc < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        return rc

    def shared_ids(self, cap: int = 65536):
        a, b, c = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = lib.alva_system_get_shared_ids(self.h, cap, a.ctypes.data, b.ctypes.data, c.ctypes.data)
        if n < 0:
            raise AlvaError(lib.alva_system_last_error().decode())
        n = min(n, cap)
        return a[:n], b[:n], c[:n]

    def counters(self):
        out = (C.c_long * 3)()
        lib.alva_system_debug_counters(self.h, out)
        return dict(ba_solves=out[0], merges=out[1], culled_keyframes=out[2])

    def klt_work(self, reset: bool = True):
        """(keypoint-levels, slots) of the tracking steps since the last reset"""
        out = (C.c_long * 2)()
        lib.alva_system_debug_klt_work(self.h, out, int(reset))
        return int(out[0]), int(out[1])

    def timing(self, reset: bool = True):
        """seconds per section of the frame loop since the last reset (see alva_system_debug_timing)"""
        out = np.zeros(8)
        lib.alva_system_debug_timing(self.h, out.ctypes.data, int(reset))
        names = ("upload+pyramid", "gather", "track_step", "track_apply", "pose_wait", "pose_apply+kf_check", "keyframe_create", "mapping")
        return dict(zip(names, out))

    def timing_keyframe(self, reset: bool = True):
        out = np.zeros(16)
        lib.alva_system_debug_timing_keyframe(self.h, out.ctypes.data, int(reset))
        names = ("prepare", "describe_tracked", "detect", "describe_new", "insert+copy", "triangulate", "covisibility", "local_map_matching",
                 "optimize", "(match stage)", "(BA stage)", "(BA build)", "(BA solves+sweep)", "(BA write-back)", "(BA culling)", "(descriptor medoids)")
        return dict(zip(names, out[:16]))

    FINE_NAMES = ("match: local-map union", "match: flatten+stage", "match: merges", "flatten: kf table + cells", "flatten: local list",
                  "flatten: per map point", "BA build: keyframes + point set", "BA build: observations", "filter: keyframe removals",
                  "window: remove kf-30", "copy: frame", "copy: order mirror", "copy: observation mirror", "new keypoints + map points",
                  "covis: counts", "covis: local ids", "covis: into local map", "parallax pairs (all frames)", "parallax (all frames)", "parallax sort (all frames)",
                  "#flattened map points", "#flattened observations", "#local candidates", "#BA points", "#BA residual blocks", "#BA keyframes",
                  "#new keypoints", "#local ids", "#covisible keyframes", "#frames: slot table carried", "#frames: slot table assembled", "probe 31")   # (31: scratch for measurements)

    def timing_fine(self, reset: bool = True):
        out = np.zeros(32)
        lib.alva_system_debug_timing_fine(self.h, out.ctypes.data, int(reset))
        return {n: v for n, v in zip(self.FINE_NAMES, out) if n}

    def set_init_pose(self, pose7):
        if pose7 is None:
            lib.alva_system_debug_set_init_pose(self.h, None)
        else:
            p = np.ascontiguousarray(pose7, np.float64)
            lib.alva_system_debug_set_init_pose(self.h, p.ctypes.data)


class SystemGroup:
    """alva_system_group: S AlvaAR sessions advanced one frame per call on `n_threads` host threads (sessions are fibers: a session's
    waits for the GPU run the thread's other sessions).  No reference counterpart (the reference is one System per worker)."""

    def __init__(self, sessions, n_threads: int):
        h = _vp()
        rc = lib.alva_system_group_create(int(n_threads), C.byref(h))
        if rc:
            raise AlvaError("alva_system_group_create failed")
        self.h = h
        self.set_sessions(sessions)

    def stream(self, index: int, device: int = 0):
        """the group's shared HIP stream number `index` (created on first use): pass it to AlvaAR(..., hip_stream=...)"""
        st = _vp()
        if lib.alva_system_group_stream(self.h, device, index, C.byref(st)):
            raise AlvaError("alva_system_group_stream failed")
        return st

    def set_lockstep(self, on: bool):
        """lock-step launches (include/alvaar_system.h): one launch per kernel kind for the sessions of a worker that share a stream"""
        if lib.alva_system_group_set_lockstep(self.h, 1 if on else 0):
            raise AlvaError("alva_system_group_set_lockstep failed")

    def set_lanes(self, lanes: int):
        """number of lanes (group-owned streams that carry the shared tracking-chain launches); session i -> lane (i + i // n_threads) % lanes"""
        if lib.alva_system_group_set_lanes(self.h, int(lanes)):
            raise AlvaError("alva_system_group_set_lanes failed")

    def time_stats(self, reset: bool = True):
        """workers' seconds inside group steps, seconds of them inside session slices that did work (not polls), slices, working slices"""
        out = (C.c_double * 4)()
        if lib.alva_system_group_time_stats(self.h, out, 1 if reset else 0):
            raise AlvaError("alva_system_group_time_stats failed")
        return float(out[0]), float(out[1]), int(out[2]), int(out[3])

    def launch_stats(self):
        """(combined launches issued, session launches they carried) since the group was created"""
        out = (C.c_long * 2)()
        if lib.alva_system_group_launch_stats(self.h, out):
            raise AlvaError("alva_system_group_launch_stats failed")
        return int(out[0]), int(out[1])

    def set_sessions(self, sessions):
        self.sessions = list(sessions)
        n = len(self.sessions)
        self._sys = (_vp * n)(*[s.h for s in self.sessions])
        self._ptr = (_vp * n)()
        self.poses = np.zeros((n, 16), np.float32)
        self.status = np.zeros(n, np.int32)

    def step_device(self, d_rgba_ptrs, timestamp_ms: float):
        """one frame per session (device pointers, one per session); returns the status array (view)"""
        for i, p in enumerate(d_rgba_ptrs):
            self._ptr[i] = p
        rc = lib.alva_system_group_find_camera_pose_device(self.h, len(self.sessions), self._sys, self._ptr, float(timestamp_ms),
                                                           self.poses.ctypes.data, self.status.ctypes.data)
        if rc:
            raise AlvaError("alva_system_group_find_camera_pose_device failed")
        if (self.status < 0).any():
            raise AlvaError(lib.alva_system_last_error().decode() or "a session of the group failed")
        return self.status

    def close(self):
        if getattr(self, "h", None):
            lib.alva_system_group_destroy(self.h)
            self.h = None

    __del__ = close
