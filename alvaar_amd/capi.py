"""ctypes binding of include/alvaar_hip.h.  Device buffers are torch CUDA tensors."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# torch first: it ships its own ROCm runtime (torch/lib/libamdhip64.so).  Loading our library before
# torch would bind it to /opt/rocm's copy and leave two HIP runtimes in one process.
import torch  # noqa: F401  (plumbing: device memory + streams)

import os
# ALVA_LIB: another build of the same library (A/B measurements on one GPU box); the default is the in-tree build
_LIB_PATH = Path(os.environ["ALVA_LIB"]) if os.environ.get("ALVA_LIB") else Path(__file__).resolve().parent / "libalvaar_hip.so"


class AlvaError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not _LIB_PATH.exists():
        raise AlvaError(
            f"{_LIB_PATH} is missing: build it with `python -m alvaar_amd.build` "
            "(there is deliberately no CPU fallback)")
    return C.CDLL(str(_LIB_PATH))


lib = _load()

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t


class PyrLevel(C.Structure):
    _fields_ = [("width", _i), ("height", _i), ("d_gray", _vp), ("gray_pitch", _sz),
                ("d_deriv", _vp), ("deriv_pitch", _sz)]


def _sig(name, argtypes, restype=_i):
    if os.environ.get("ALVA_LIB") and not hasattr(lib, name):
        return None   # an OLDER build loaded for an A/B measurement may lack the newest entry points (the in-tree build never does)
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


_sig("alva_last_error", [], C.c_char_p)
_sig("alva_version", [], C.c_char_p)
_sig("alva_ctx_create", [_i, _vp, _i, C.POINTER(_vp)])
_sig("alva_ctx_create_with_priority", [_i, _i, C.POINTER(_vp)])
_sig("alva_ctx_destroy", [_vp], None)
_sig("alva_ctx_sync", [_vp])
_sig("alva_rgba2gray", [_vp, _vp, _sz, _i, _i, _vp, _sz])
_sig("alva_pyramid_create", [_vp, _i, _i, _i, _i, C.POINTER(_vp)])
_sig("alva_pyramid_destroy", [_vp], None)
_sig("alva_pyramid_num_levels", [_vp])
_sig("alva_pyramid_level", [_vp, _i, C.POINTER(PyrLevel)])
_sig("alva_pyramid_build_from_gray", [_vp, _vp, _vp, _sz])
_sig("alva_pyramid_download_level", [_vp, _vp, _i, _vp, _vp])
_sig("alva_pyramid_build_from_rgba", [_vp, _vp, _vp, _sz, _vp, _sz])
_sig("alva_pyramid_build_from_rgba_batch", [_vp, _vp, _vp, _sz, _vp, _sz, _i])
_sig("alva_lk_track", [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i])
_sig("alva_fbklt_track", [_vp, _vp, _vp, _i, _f, _f, _i, _f, _vp, _vp, _vp, _i])
_sig("alva_p3p_draw_samples", [_i, _i, _i, C.c_uint32, _vp])
_sig("alva_p3p_lmeds", [_vp, _vp, _vp, _i, _i, _f, _i, C.c_uint32, _f, _f, _vp, _vp, _vp, _vp, _vp])
_sig("alva_pnp_refine", [_vp, _vp, _vp, _i, _vp, _i, _f, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp])
_sig("alva_local_ba", [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp])
_sig("alva_detect_grid", [_vp, _vp, _sz, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp])
_sig("alva_fast", [_vp, _vp, _sz, _i, _i, _i, _vp, _vp, _i, _vp])
_sig("alva_orb_create", [_vp, _i, _i, _i, _f, _i, _i, C.POINTER(_vp)])
_sig("alva_orb_destroy", [_vp], None)
_sig("alva_orb_detect_and_compute", [_vp, _vp, _vp, _sz, _vp, _vp, _i, _vp])
_sig("alva_ctx_wait", [_vp, _vp])
_sig("alva_prof_enable", [_i])
_sig("alva_prof_report", [C.c_char_p, _sz])
_sig("alva_orb_collect", [_vp, _vp, _vp])
_sig("alva_orb_debug_level", [_vp, _vp, _i, _i, _vp, _sz, _vp, _vp])
_sig("alva_match_to_map_records", [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _f, _f, _vp])
_sig("alva_match_to_map", [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _f, _vp])
_sig("alva_undistort_points", [_vp, _vp, _i] + [C.c_double] * 8 + [_vp])
_sig("alva_project_dist", [_vp, _vp, _i] + [C.c_double] * 8 + [_vp])
_sig("alva_clahe", [_vp, _vp, _sz, _i, _i, C.c_double, _i, _i, _vp, _sz])
_sig("alva_triangulate", [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _f, _vp, _vp, _vp, _vp, _vp])
_sig("alva_frontend_create", [_i, _i, _i, _i, _i, C.POINTER(_vp)])
_sig("alva_frontend_destroy", [_vp], None)
_sig("alva_frontend_track", [_vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp])
_sig("alva_frontend_track_ahead", [_vp, _vp, _sz, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp])
_sig("alva_frontend_results", [_vp] + [C.POINTER(_vp)] * 6)
_sig("alva_frontend_sync", [_vp])
_sig("alva_frontend_run_many", [_vp, _i, _i, _i, _vp, _i, _sz, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp])
_sig("alva_fbklt_track_batch", [_vp, _vp, _vp, _i, _i, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _i])
_sig("alva_track_batch_create", [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)])
_sig("alva_track_batch_destroy", [_vp], None)
_sig("alva_track_batch_step", [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp])
_sig("alva_track_batch_results", [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)])
_sig("alva_track_batch_ctx", [_vp], _vp)
_sig("alva_track_batch_stats", [_vp, _vp, _vp])
_sig("alva_track_batch_enable_detector", [_vp, _i])
_sig("alva_track_batch_step_detect", [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp])
_sig("alva_track_batch_detections", [_vp, _i] + [C.POINTER(_vp)] * 4)
_sig("alva_orb_detect_and_compute_batch", [_vp, _vp, _i, _vp, _sz, _vp, _vp, _i])
_sig("alva_orb_collect_batch", [_vp, _vp, _i, _vp])
_sig("alva_orb_device_count", [_vp], _vp)
_sig("alva_orb_ambiguous_rotations", [_vp, _i])
_sig("alva_bf_match_hamming_batch", [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i])
_sig("alva_track_batch_set_klt_lanes", [_vp, _i])
_sig("alva_compute_pose_enqueue", [_vp, _vp, _vp, _vp, _i, _i, _f, _i, C.c_uint32, _i, _f, _f, _f, _f, _f])
_sig("alva_compute_pose_collect", [_vp, _vp, _vp, _vp, _vp])
_sig("alva_compute_pose", [_vp, _vp, _vp, _vp, _i, _i, _f, _i, C.c_uint32, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp])
_sig("alva_describe", [_vp, _vp, _sz, _i, _i, _vp, _i, _vp, _vp])
_sig("alva_orb_blur", [_vp, _vp, _sz, _i, _i, _vp, _sz])
_sig("alva_bf_match_hamming", [_vp, _vp, _i, _vp, _i, _vp, _vp])
_sig("alva_find_plane", [_vp, _vp, _i, _vp, _i, _i, C.c_uint32, _vp, _vp, _vp])
_sig("alva_relpose_draw_samples", [_i, _i, _i, C.c_uint32, _vp])
_sig("alva_relpose_hypotheses", [_vp, _vp, _vp, _i, _vp, _i, _f, _f, _f, _vp, _vp])
_sig("alva_compute_5pt_essential", [_vp, _vp, _vp, _i, _i, _f, _i, _i, C.c_uint32, _f, _f, _vp, _vp, _vp, _vp, _vp])


class RelposeInfo(C.Structure):
    _fields_ = [("iterations", _i), ("n_inliers", _i), ("draws", _i), ("lm_iterations", _i), ("lm_status", _i), ("lm_nfev", _i),
                ("ransac_model", C.c_double * 12)]


def check(rc: int) -> None:
    if rc != 0:
        raise AlvaError(f"alvaar_hip error {rc}: {lib.alva_last_error().decode()}")


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


class Context:
    """One HIP stream on one device (alva_ctx).  By default enqueues on torch's current stream so
    torch.cuda.synchronize()/Events see the work."""

    def __init__(self, device: int = 0, stream: int | None = None, own_stream: bool = False):
        if not torch.cuda.is_available():
            raise AlvaError("no HIP device visible: the alvaar_amd hot path has no CPU fallback")
        self.device = device
        if stream is None and not own_stream:
            stream = torch.cuda.current_stream(device).cuda_stream
        h = _vp()
        check(lib.alva_ctx_create(device, _vp(stream or 0), 1 if own_stream else 0, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:   # `lib` is already gone at interpreter shutdown
            lib.alva_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def sync(self):
        check(lib.alva_ctx_sync(self.h))

    # a2
    def rgba2gray(self, rgba, out=None):
        h, w, c = rgba.shape
        assert c == 4 and rgba.dtype == torch.uint8 and rgba.is_contiguous()
        if out is None:
            out = torch.empty((h, w), dtype=torch.uint8, device=rgba.device)
        check(lib.alva_rgba2gray(self.h, _ptr(rgba), w * 4, w, h, _ptr(out), out.stride(0)))
        return out


    # a4
    def lk_track(self, prev: "Pyramid", nxt: "Pyramid", pts, init, num_levels=3, max_iters=30, eps=0.01):
        n = pts.shape[0]
        out = init.clone().contiguous()
        status = torch.empty(n, dtype=torch.uint8, device=pts.device)
        err = torch.empty(n, dtype=torch.float32, device=pts.device)
        check(lib.alva_lk_track(self.h, prev.h, nxt.h, num_levels, max_iters, eps, _ptr(pts), _ptr(out), _ptr(status), _ptr(err), n))
        return out, status, err

    def fbklt_track(self, prev: "Pyramid", curr: "Pyramid", pts, prior, num_levels=3, err_thresh=30.0, fb_dist=0.5,
                    max_iters=30, eps=0.01):
        """FeatureTracker::fbKltTracking: returns (updated prior [n,2], status [n] u8)."""
        n = pts.shape[0]
        out = prior.clone().contiguous()
        status = torch.empty(n, dtype=torch.uint8, device=pts.device)
        check(lib.alva_fbklt_track(self.h, prev.h, curr.h, num_levels, err_thresh, fb_dist, max_iters, eps, _ptr(pts), _ptr(out),
                                   _ptr(status), n))
        return out, status

    # a8
    def p3p_lmeds(self, bearings, wpts, max_iters=100, err=3.0, fx=579.4, fy=579.4, do_random=False, seed=12345):
        """MultiViewGeometry::p3pRansac: returns (ok, R [3,3] numpy, t [3] numpy, outlier indices numpy)."""
        import numpy as np
        n = bearings.shape[0]
        assert bearings.dtype == torch.float64 and wpts.dtype == torch.float64
        R = np.zeros((3, 3))
        t = np.zeros(3)
        out = np.zeros(max(n, 1), np.int32)
        nout = C.c_int(0)
        ok = C.c_int(0)
        check(lib.alva_p3p_lmeds(self.h, _ptr(bearings), _ptr(wpts), n, max_iters, err, int(do_random), seed, fx, fy,
                                 R.ctypes.data, t.ctypes.data, out.ctypes.data, C.byref(nout), C.byref(ok)))
        return bool(ok.value), R, t, out[:nout.value].copy()

    # f2b
    def compute_5pt_essential(self, bv1, bv2, max_iters=100, err=3.0, optimize=True, fx=579.4, fy=579.4, do_random=False, seed=12345):
        """MultiViewGeometry::compute5ptEssentialMatrix: returns (ok, Rwc [3,3], twc [3], inlier mask [n] bool, RelposeInfo)."""
        import numpy as np
        n = bv1.shape[0]
        assert bv1.dtype == torch.float64 and bv2.dtype == torch.float64 and bv2.shape[0] == n
        R = np.zeros((3, 3))
        t = np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        info = RelposeInfo()
        ok = C.c_int(0)
        check(lib.alva_compute_5pt_essential(self.h, _ptr(bv1), _ptr(bv2), n, int(max_iters), float(err), int(optimize), int(do_random),
                                             int(seed), float(fx), float(fy), R.ctypes.data, t.ctypes.data, mask.ctypes.data,
                                             C.addressof(info), C.addressof(ok)))
        return bool(ok.value), R, t, mask[:n].astype(bool), info

    # f3 (parity unpinned)
    def find_plane(self, points, pose7, num_iterations=250, samples3=None, do_random=False, seed=12345):
        """The intended System::processPlane: returns the plane pose [16] float32 or None."""
        import numpy as np
        assert points.dtype == torch.float64
        pose = np.ascontiguousarray(pose7, np.float64)
        out = np.zeros(16, np.float32)
        found = C.c_int(0)
        s = None if samples3 is None else np.ascontiguousarray(samples3, np.int32)
        check(lib.alva_find_plane(self.h, _ptr(points), points.shape[0], pose.ctypes.data, int(num_iterations if s is None else len(s)),
                                  int(do_random), int(seed), None if s is None else s.ctypes.data, out.ctypes.data, C.addressof(found)))
        return out if found.value else None

    def relpose_hypotheses(self, bv1, bv2, samples8, err=3.0, fx=579.4, fy=579.4):
        """One RANSAC hypothesis per 8-index sample: returns (models [H,12] = R row-major | t, inlier counts [H], -1 = no model)."""
        import numpy as np
        s = np.ascontiguousarray(samples8, np.int32)
        H = s.shape[0]
        models = np.zeros((H, 12))
        counts = np.zeros(H, np.int32)
        check(lib.alva_relpose_hypotheses(self.h, _ptr(bv1), _ptr(bv2), bv1.shape[0], s.ctypes.data, H, float(err), float(fx), float(fy),
                                          models.ctypes.data, counts.ctypes.data))
        return models, counts

    # a9
    def pnp_refine(self, uv, wpts, pose7, K, max_iters=5, chi2th=5.9915, robust=True, l2=True):
        """MultiViewGeometry::ceresPnP: returns (ok, pose7 numpy, outlier indices numpy, info[8])."""
        import numpy as np
        n = uv.shape[0]
        assert uv.dtype == torch.float64 and wpts.dtype == torch.float64
        pose = np.ascontiguousarray(pose7, np.float64).copy()
        out = np.zeros(max(n, 1), np.int32)
        nout = C.c_int(0)
        ok = C.c_int(0)
        info = np.zeros(8)
        check(lib.alva_pnp_refine(self.h, _ptr(uv), _ptr(wpts), n, pose.ctypes.data, max_iters, chi2th, int(robust), int(l2),
                                  K[0], K[1], K[2], K[3], out.ctypes.data, C.byref(nout), info.ctypes.data, C.byref(ok)))
        return bool(ok.value), pose, out[:nout.value].copy(), info

    # a10-a13
    def local_ba(self, pb, max_iters=5, ftol=0.0, huber_chi2=5.9915, inv_depth=True):
        """The solve of Optimizer::localBA on a flat problem dict (see synth.make_ba_problem)."""
        import numpy as np
        poses = np.ascontiguousarray(pb["poses"], np.float64).copy()
        kfc = np.ascontiguousarray(pb["kf_const"], np.uint8)
        calib = np.ascontiguousarray(pb["calib"], np.float64)
        akf = np.ascontiguousarray(pb["anchor_kf"], np.int32)
        auv = np.ascontiguousarray(pb["anchor_uv"], np.float64)
        pts = np.ascontiguousarray(pb["inv_depth"] if inv_depth else pb["pts_xyz"], np.float64).copy()
        okf = np.ascontiguousarray(pb["obs_kf"], np.int32)
        opt = np.ascontiguousarray(pb["obs_pt"], np.int32)
        ouv = np.ascontiguousarray(pb["obs_uv"], np.float64)
        nobs = len(okf)
        chi2 = np.zeros(max(nobs, 1))
        depth = np.zeros(max(nobs, 1), np.uint8)
        info = np.zeros(9)
        ok = C.c_int(0)
        check(lib.alva_local_ba(self.h, len(poses), poses.ctypes.data, kfc.ctypes.data, calib.ctypes.data, int(inv_depth), len(akf),
                                akf.ctypes.data, auv.ctypes.data, pts.ctypes.data, nobs, okf.ctypes.data, opt.ctypes.data,
                                ouv.ctypes.data, max_iters, ftol, huber_chi2, chi2.ctypes.data, depth.ctypes.data, info.ctypes.data,
                                C.byref(ok)))
        return dict(ok=bool(ok.value), poses=poses, pts=pts, chi2=chi2[:nobs], depth=depth[:nobs], info=info)

    def local_ba_csr(self, pb, max_iters=5, ftol=0.0, huber_chi2=5.9915, chi2_threshold=5.9915):
        """alva_local_ba_csr: the same solve for observations grouped by point (the problem dict's observations are stably sorted by
        point here); the sweep's test per residual block comes back as booleans in the SORTED order, `order` maps back."""
        import numpy as np
        poses = np.ascontiguousarray(pb["poses"], np.float64).copy()
        kfc = np.ascontiguousarray(pb["kf_const"], np.uint8)
        calib = np.ascontiguousarray(pb["calib"], np.float64)
        akf = np.ascontiguousarray(pb["anchor_kf"], np.int32)
        auv = np.ascontiguousarray(pb["anchor_uv"], np.float64)
        pts = np.ascontiguousarray(pb["inv_depth"], np.float64).copy()
        opt0 = np.asarray(pb["obs_pt"], np.int64)
        order = np.argsort(opt0, kind="stable")
        okf = np.ascontiguousarray(np.asarray(pb["obs_kf"], np.int32)[order])
        ouv = np.ascontiguousarray(np.asarray(pb["obs_uv"], np.float64)[order])
        nobs, npt = len(okf), len(akf)
        ptr = np.zeros(npt + 1, np.int32)
        np.cumsum(np.bincount(opt0, minlength=npt), out=ptr[1:])
        bits = np.zeros((nobs + 63) // 64 + 1, np.uint64)
        nbad, ok, info = C.c_int(0), C.c_int(0), np.zeros(9)
        lib.alva_local_ba_csr.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _d, _d, _d, _vp, _vp, _vp, _vp]
        check(lib.alva_local_ba_csr(self.h, len(poses), poses.ctypes.data, kfc.ctypes.data, calib.ctypes.data, npt, ptr.ctypes.data, akf.ctypes.data,
                                    auv.ctypes.data, pts.ctypes.data, nobs, okf.ctypes.data, ouv.ctypes.data, max_iters, ftol, huber_chi2,
                                    chi2_threshold, bits.ctypes.data, C.byref(nbad), info.ctypes.data, C.byref(ok)))
        bad = np.unpackbits(bits.view(np.uint8), bitorder="little")[:nobs].astype(bool)
        return dict(ok=bool(ok.value), poses=poses, pts=pts, bad=bad, n_bad=nbad.value, order=order, info=info)

    def local_ba_batch(self, pbs, max_iters=5, ftol=0.0, huber_chi2=5.9915):
        """alva_local_ba_batch on a list of flat problem dicts (anchored inverse depth); returns one result dict per problem"""
        import numpy as np
        n = len(pbs)
        keep = []

        def arr(x, dt, copy=False):
            a = np.ascontiguousarray(x, dt)
            a = a.copy() if copy else a
            keep.append(a)
            return a
        poses = [arr(pb["poses"], np.float64, True) for pb in pbs]
        kfc = [arr(pb["kf_const"], np.uint8) for pb in pbs]
        akf = [arr(pb["anchor_kf"], np.int32) for pb in pbs]
        auv = [arr(pb["anchor_uv"], np.float64) for pb in pbs]
        pts = [arr(pb["inv_depth"], np.float64, True) for pb in pbs]
        okf = [arr(pb["obs_kf"], np.int32) for pb in pbs]
        opt = [arr(pb["obs_pt"], np.int32) for pb in pbs]
        ouv = [arr(pb["obs_uv"], np.float64) for pb in pbs]
        chi2 = [np.zeros(len(o)) for o in okf]
        depth = [np.zeros(len(o), np.uint8) for o in okf]
        calib = arr(pbs[0]["calib"], np.float64)
        info = np.zeros((n, 4))
        ok = np.zeros(n, np.int32)
        ptrs = lambda lst: (C.c_void_p * n)(*[a.ctypes.data for a in lst])
        ints = lambda vals: (C.c_int * n)(*vals)
        lib.alva_local_ba_batch.argtypes = [_vp, _i] + [_vp] * 3 + [_vp] + [_vp] * 4 + [_vp] * 4 + [_i, _d, _d] + [_vp] * 4
        check(lib.alva_local_ba_batch(self.h, n, ints([len(p) for p in poses]), ptrs(poses), ptrs(kfc), calib.ctypes.data, ints([len(a) for a in akf]),
                                      ptrs(akf), ptrs(auv), ptrs(pts), ints([len(o) for o in okf]), ptrs(okf), ptrs(opt), ptrs(ouv), max_iters, ftol,
                                      huber_chi2, ptrs(chi2), ptrs(depth), info.ctypes.data, ok.ctypes.data))
        return [dict(ok=bool(ok[b]), poses=poses[b], pts=pts[b], chi2=chi2[b], depth=depth[b], info=info[b]) for b in range(n)]

    # a5
    def detect_grid(self, gray, cell, occupied=None, roi=None, max_quality=0.001, cap=None):
        """FeatureExtractor::detectFeaturePoints: returns (pts [n,2] float32 cuda tensor, new max_quality)."""
        h, w = gray.shape
        if roi is None:
            roi = (20, 20, w - 40, h - 40)
        if cap is None:
            cap = 2 * (w // cell) * (h // cell) + 1
        nocc = 0 if occupied is None else occupied.shape[0]
        out = torch.zeros((cap, 2), dtype=torch.float32, device=gray.device)
        mq = C.c_double(max_quality)
        cnt = C.c_int(0)
        check(lib.alva_detect_grid(self.h, _ptr(gray), gray.stride(0), w, h, cell, _ptr(occupied) if nocc else None, nocc,
                                   roi[0], roi[1], roi[2], roi[3], C.byref(mq), _ptr(out), cap, C.byref(cnt)))
        return out[:min(cnt.value, cap)], mq.value

    # a5'
    def fast(self, gray, threshold=20, cap=200000):
        """cv::FAST(TYPE_9_16, NMS): returns (xy [n,2] int32, score [n] int32) cuda tensors, row-major order."""
        h, w = gray.shape
        xy = torch.zeros((cap, 2), dtype=torch.int32, device=gray.device)
        sc = torch.zeros(cap, dtype=torch.int32, device=gray.device)
        cnt = C.c_int(0)
        check(lib.alva_fast(self.h, _ptr(gray), gray.stride(0), w, h, threshold, _ptr(xy), _ptr(sc), cap, C.byref(cnt)))
        n = min(cnt.value, cap)
        return xy[:n], sc[:n]

    # f1
    def match_to_map(self, calib10, cell_size, num_cells_w, grid_cells, cell_ptr, cell_mp, kf_q, kf_t, mp_wpt, mp_is3d, obs_ptr, obs_kf,
                     obs_px, obs_desc, frame_kf, num_kp3d, local, max_proj_err=2.0, dist_ratio=0.2):
        """Mapper::matchToMap on a flattened map (see include/alvaar_hip.h); all arrays cuda tensors, calib10 a host sequence.
        Returns match_of_mp [n_mp] int32 (cuda)."""
        import numpy as np
        n_mp = mp_wpt.shape[0]
        out = torch.empty(n_mp, dtype=torch.int32, device=mp_wpt.device)
        cal = np.ascontiguousarray(calib10, np.float64)
        check(lib.alva_match_to_map(self.h, cal.ctypes.data, int(cell_size), int(num_cells_w), int(grid_cells), _ptr(cell_ptr), _ptr(cell_mp),
                                    kf_q.shape[0], _ptr(kf_q), _ptr(kf_t), n_mp, _ptr(mp_wpt), _ptr(mp_is3d), _ptr(obs_ptr), _ptr(obs_kf),
                                    _ptr(obs_px), _ptr(obs_desc), int(frame_kf), int(num_kp3d), local.shape[0], _ptr(local),
                                    float(max_proj_err), float(dist_ratio), _ptr(out)))
        return out

    MP_ENT_CAP = 40   # csrc/slam/mp_rec.hpp

    @staticmethod
    def mp_record_dtype():
        """numpy layout of one map-point record (csrc/slam/mp_rec.hpp, 1024 bytes)"""
        import numpy as np
        ent = np.dtype([("kf", "<i4"), ("flags", "u1"), ("pad", "u1", 3), ("px", "<f4", 2), ("unpx", "<f4", 2)])
        rec = np.dtype([("X", "<f8", 3), ("inv_depth", "<f8"), ("id", "<i4"), ("anchor_kf", "<i4"), ("is3d", "u1"), ("has_desc", "u1"),
                        ("observed", "u1"), ("n_ent", "u1"), ("n_obs", "u1"), ("overflow", "u1"), ("pad8", "u1", 2), ("dev_slot", "<i4"),
                        ("pad32", "<i4", 3), ("ent", ent, Context.MP_ENT_CAP)])
        assert ent.itemsize == 24 and rec.itemsize == 1024
        return rec

    def match_to_map_records(self, calib10, cell_size, num_cells_w, grid_cells, cell_ptr, cell_mp, kf_ids, kf_q, kf_t, frame_kf_index, mp_slot,
                             chunk_table, medoid_store, num_kp3d, local, max_proj_err=2.0, dist_ratio=0.2):
        """Mapper::matchToMap on map-point RECORDS (alva_match_to_map_records): mp_slot [n_mp] names each row's record; chunk_table is a cuda
        int64 tensor of the record chunks' addresses (pinned host memory, 4096 records per chunk); descriptors come from medoid_store's
        tables.  Returns match_of_mp [n_mp] int32 (cuda)."""
        import numpy as np
        lib.alva_medoid_tables.restype = C.c_void_p
        lib.alva_medoid_tables.argtypes = [C.c_void_p]
        n_mp = mp_slot.shape[0]
        out = torch.empty(n_mp, dtype=torch.int32, device=mp_slot.device)
        cal = np.ascontiguousarray(calib10, np.float64)
        frame_kf_id = int(kf_ids[int(frame_kf_index)].item())
        check(lib.alva_match_to_map_records(self.h, cal.ctypes.data, int(cell_size), int(num_cells_w), int(grid_cells), _ptr(cell_ptr), _ptr(cell_mp),
                                            kf_ids.shape[0], _ptr(kf_ids), _ptr(kf_q), _ptr(kf_t), int(frame_kf_index), frame_kf_id, n_mp, _ptr(mp_slot),
                                            _ptr(chunk_table), lib.alva_medoid_tables(medoid_store.h), int(num_kp3d), local.shape[0],
                                            _ptr(local), float(max_proj_err), float(dist_ratio), _ptr(out)))
        return out

    # f4b
    def undistort_points(self, px, K, dist):
        """CameraCalibration::undistortImagePoint for n pixels [n,2] f32; K = (fx,fy,cx,cy), dist = (k1,k2,p1,p2)"""
        out = torch.empty_like(px)
        check(lib.alva_undistort_points(self.h, _ptr(px), px.shape[0], *[float(v) for v in K], *[float(v) for v in dist], _ptr(out)))
        return out

    def project_dist(self, cam_pts, K, dist):
        """CameraCalibration::projectCamToImageDist for n camera-frame points [n,3] f64 -> [n,2] f32 pixels"""
        out = torch.empty((cam_pts.shape[0], 2), dtype=torch.float32, device=cam_pts.device)
        check(lib.alva_project_dist(self.h, _ptr(cam_pts), cam_pts.shape[0], *[float(v) for v in K], *[float(v) for v in dist], _ptr(out)))
        return out

    # f4a
    def clahe(self, gray, clip_limit=3.0, tiles=None, tile_size=50):
        """cv::createCLAHE(clip, tiles)->apply; default grid = size / 50 as VisualFrontend builds it"""
        h, w = gray.shape
        tx, ty = tiles if tiles is not None else (w // tile_size, h // tile_size)
        out = torch.empty_like(gray)
        check(lib.alva_clahe(self.h, _ptr(gray), gray.stride(0), w, h, float(clip_limit), int(tx), int(ty), _ptr(out), out.stride(0)))
        return out

    # f2a
    def triangulate(self, T, group, bvl, bvr, unpxl, unpxr, K, max_reproj_err=3.0):
        """Mapper::triangulateTemporal per keypoint: returns dict(lpt, wpt, inv_depth, status, parallax) of cuda tensors."""
        n = bvl.shape[0]
        dev = bvl.device
        out = dict(lpt=torch.empty((n, 3), dtype=torch.float64, device=dev), wpt=torch.empty((n, 3), dtype=torch.float64, device=dev),
                   inv_depth=torch.empty(n, dtype=torch.float64, device=dev), status=torch.empty(n, dtype=torch.uint8, device=dev),
                   parallax=torch.empty(n, dtype=torch.float64, device=dev))
        check(lib.alva_triangulate(self.h, n, _ptr(T), T.shape[0], _ptr(group), _ptr(bvl), _ptr(bvr), _ptr(unpxl), _ptr(unpxr),
                                   K[0], K[1], K[2], K[3], max_reproj_err, _ptr(out["lpt"]), _ptr(out["wpt"]), _ptr(out["inv_depth"]),
                                   _ptr(out["status"]), _ptr(out["parallax"])))
        return out

    def wait_for(self, producer: "Context"):
        """stream-order dependency on another context's enqueued work (no host wait)"""
        check(lib.alva_ctx_wait(self.h, producer.h))

    # a8 + a9 chained
    def compute_pose(self, bearings, uv, wpts, K, p3p_iters=100, p3p_err=3.0, pnp_iters=5, chi2th=5.9915, do_random=False, seed=12345):
        """VisualFrontend::computePose: returns (status 0/1/2, pose7, p3p_outlier mask, pnp_outlier mask)."""
        import numpy as np
        n = bearings.shape[0]
        pose = np.zeros(7)
        m1 = np.zeros(max(n, 1), np.uint8)
        m2 = np.zeros(max(n, 1), np.uint8)
        st = C.c_int(0)
        check(lib.alva_compute_pose(self.h, _ptr(bearings), _ptr(uv), _ptr(wpts), n, p3p_iters, p3p_err, int(do_random), seed, pnp_iters,
                                    chi2th, K[0], K[1], K[2], K[3], pose.ctypes.data, m1.ctypes.data, m2.ctypes.data, C.byref(st)))
        return st.value, pose, m1[:n].astype(bool), m2[:n].astype(bool)

    def compute_pose_enqueue(self, bearings, uv, wpts, K, p3p_iters=100, p3p_err=3.0, pnp_iters=5, chi2th=5.9915, do_random=False,
                             seed=12345):
        self._pose_n = bearings.shape[0]
        self._pose_keep = (bearings, uv, wpts)
        check(lib.alva_compute_pose_enqueue(self.h, _ptr(bearings), _ptr(uv), _ptr(wpts), self._pose_n, p3p_iters, p3p_err, int(do_random),
                                            seed, pnp_iters, chi2th, K[0], K[1], K[2], K[3]))

    def compute_pose_collect(self):
        import numpy as np
        n = self._pose_n
        pose = np.zeros(7)
        m1 = np.zeros(max(n, 1), np.uint8)
        m2 = np.zeros(max(n, 1), np.uint8)
        st = C.c_int(0)
        check(lib.alva_compute_pose_collect(self.h, pose.ctypes.data, m1.ctypes.data, m2.ctypes.data, C.byref(st)))
        self._pose_keep = None
        return st.value, pose, m1[:n].astype(bool), m2[:n].astype(bool)

    # a6
    def orb_blur(self, gray):
        h, w = gray.shape
        out = torch.empty((h, w), dtype=torch.uint8, device=gray.device)
        check(lib.alva_orb_blur(self.h, _ptr(gray), gray.stride(0), w, h, _ptr(out), out.stride(0)))
        return out

    def describe(self, gray, pts):
        """FeatureExtractor::describeFeaturePoints: returns (desc [n,32] u8, valid [n] u8)."""
        h, w = gray.shape
        n = pts.shape[0]
        assert pts.dtype == torch.float32 and pts.is_contiguous()
        desc = torch.empty((n, 32), dtype=torch.uint8, device=gray.device)
        valid = torch.empty(n, dtype=torch.uint8, device=gray.device)
        check(lib.alva_describe(self.h, _ptr(gray), gray.stride(0), w, h, _ptr(pts), n, _ptr(desc), _ptr(valid)))
        return desc, valid

    # a7
    def bf_match_hamming(self, query, train):
        nq, nt = query.shape[0], train.shape[0]
        assert query.dtype == torch.uint8 and query.shape[1] == 32 and query.is_contiguous()
        assert train.dtype == torch.uint8 and train.shape[1] == 32 and train.is_contiguous()
        idx = torch.empty(nq, dtype=torch.int32, device=query.device)
        dist = torch.empty(nq, dtype=torch.int32, device=query.device)
        check(lib.alva_bf_match_hamming(self.h, _ptr(query), nq, _ptr(train), nt, _ptr(idx), _ptr(dist)))
        return idx, dist


class Pyramid:
    """alva_pyramid: padded gray + Scharr-derivative levels resident in HBM."""

    def __init__(self, ctx: Context, width: int, height: int, win: int = 9, max_level: int = 3):
        self.ctx = ctx
        self.win = win
        self.width, self.height = width, height
        h = _vp()
        check(lib.alva_pyramid_create(ctx.h, width, height, win, max_level, C.byref(h)))
        self.h = h
        self.num_levels = lib.alva_pyramid_num_levels(self.h)

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.alva_pyramid_destroy(self.h)
            self.h = None

    __del__ = close

    def build_from_gray(self, gray):
        check(lib.alva_pyramid_build_from_gray(self.ctx.h, self.h, _ptr(gray), gray.stride(0)))

    def build_from_rgba(self, rgba, gray_out=None):
        h, w, _ = rgba.shape
        check(lib.alva_pyramid_build_from_rgba(self.ctx.h, self.h, _ptr(rgba), w * 4, _ptr(gray_out),
                                               0 if gray_out is None else gray_out.stride(0)))

    def level(self, l: int) -> PyrLevel:
        info = PyrLevel()
        check(lib.alva_pyramid_level(self.h, l, C.byref(info)))
        return info

    def download_level(self, l: int):
        """Returns (gray_padded u8 [H+2w, W+2w], deriv_padded i16 [H+2w, W+2w, 2]) as numpy (tests)."""
        import numpy as np
        info = self.level(l)
        W, H = info.width + 2 * self.win, info.height + 2 * self.win
        g = np.empty((H, W), np.uint8)
        d = np.empty((H, W, 2), np.int16)
        check(lib.alva_pyramid_download_level(self.ctx.h, self.h, l, g.ctypes.data, d.ctypes.data))
        return g, d


class Orb:
    """alva_orb: cv::ORB::create(nfeatures, scale, nlevels, 31, 0, 2, HARRIS_SCORE, 31, fast_threshold)."""

    def __init__(self, ctx: Context, width: int, height: int, nfeatures: int = 2000, scale: float = 1.2, nlevels: int = 8,
                 fast_threshold: int = 20):
        self.ctx = ctx
        self.nfeatures = nfeatures
        h = _vp()
        check(lib.alva_orb_create(ctx.h, width, height, nfeatures, scale, nlevels, fast_threshold, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.alva_orb_destroy(self.h)
            self.h = None

    __del__ = close

    def enqueue(self, gray, kp_buf, desc_buf, ctx=None):
        """detectAndCompute without the host wait, into caller-owned buffers (kp_buf [cap,6] f32, desc_buf [cap,32] u8)."""
        self._pending = (ctx or self.ctx, kp_buf, desc_buf)
        check(lib.alva_orb_detect_and_compute(self._pending[0].h, self.h, _ptr(gray), gray.stride(0), _ptr(kp_buf), _ptr(desc_buf),
                                              kp_buf.shape[0], None))

    def collect(self):
        """wait for enqueue(): returns (kp[:n], desc[:n]) views of the caller's buffers"""
        ctx, kp, desc = self._pending
        cnt = C.c_int(0)
        check(lib.alva_orb_collect(ctx.h, self.h, C.byref(cnt)))
        n = min(cnt.value, kp.shape[0])
        return kp[:n], (desc[:n] if desc is not None else None)

    def level(self, l: int, blurred: bool = False):
        """level l of the pyramid the last run built (blurred: its 7x7 blur) as a cuda uint8 tensor [h, w] (alva_orb_debug_level)"""
        w, h = C.c_int(0), C.c_int(0)
        check(lib.alva_orb_debug_level(self.ctx.h, self.h, l, int(blurred), None, 0, C.byref(w), C.byref(h)))
        out = torch.empty((h.value, w.value), dtype=torch.uint8, device=f"cuda:{self.ctx.device}")
        check(lib.alva_orb_debug_level(self.ctx.h, self.h, l, int(blurred), _ptr(out), out.stride(0), None, None))
        return out

    def detect_and_compute(self, gray, describe=True, cap=None):
        """returns (kp [n,6] float32 {x,y,size,angle,response,octave}, desc [n,32] u8) cuda tensors"""
        if cap is None:
            cap = 4 * self.nfeatures + 1024
        kp = torch.zeros((cap, 6), dtype=torch.float32, device=gray.device)
        desc = torch.zeros((cap, 32), dtype=torch.uint8, device=gray.device) if describe else None
        cnt = C.c_int(0)
        check(lib.alva_orb_detect_and_compute(self.ctx.h, self.h, _ptr(gray), gray.stride(0), _ptr(kp), _ptr(desc), cap, C.byref(cnt)))
        n = min(cnt.value, cap)
        return kp[:n], (desc[:n] if describe else None)


class Frontend:
    """alva_frontend: native per-frame driver (VisualFrontend::trackMono order) over two HIP streams."""

    def __init__(self, device: int, width: int, height: int, max_tracked: int, orb_features: int = 2000):
        import numpy as np
        h = _vp()
        check(lib.alva_frontend_create(device, width, height, max_tracked, orb_features, C.byref(h)))
        self.h = h
        self.device = device
        self.cap = 4 * orb_features + 1024
        self.max_tracked = max_tracked
        self._pose = np.zeros(7)
        self._st = C.c_int(0)
        self._nkp = C.c_int(0)

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.alva_frontend_destroy(self.h)
            self.h = None

    __del__ = close

    def track(self, rgba, pts, bearings, uv, wpts, K, rgba_next=None):
        """returns (pose status 0/1/2, pose7 numpy (a view, overwritten by the next call), number of ORB keypoints).
        rgba_next: the frame the next call will pass as rgba (its gray + pyramid are built on a third stream meanwhile)."""
        check(lib.alva_frontend_track_ahead(self.h, _ptr(rgba), rgba.stride(0), None if rgba_next is None else _ptr(rgba_next), _ptr(pts),
                                            pts.shape[0], _ptr(bearings), _ptr(uv), _ptr(wpts), bearings.shape[0], K[0], K[1], K[2], K[3],
                                            self._pose.ctypes.data, C.byref(self._st), C.byref(self._nkp)))
        self._npts = pts.shape[0]
        return self._st.value, self._pose, self._nkp.value

    def sync(self):
        check(lib.alva_frontend_sync(self.h))

    def results(self):
        """dict of torch views (no copy) of the device-resident results of the last track(); call sync() before reading"""
        ptrs = [_vp() for _ in range(6)]
        check(lib.alva_frontend_results(self.h, *[C.byref(p) for p in ptrs]))
        n, m = self._npts, self._nkp.value

        def view(ptr, shape, dtype):
            import numpy as np
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            if nbytes == 0:
                return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")
            class _W:
                pass
            w = _W()
            typestr = {torch.float32: "<f4", torch.uint8: "|u1", torch.int32: "<i4"}[dtype]
            w.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr.value, False), "version": 2}
            return torch.as_tensor(w, device=f"cuda:{self.device}")
        return {"tracked": view(ptrs[0], (n, 2), torch.float32), "status": view(ptrs[1], (n,), torch.uint8),
                "keypoints": view(ptrs[2], (m, 6), torch.float32), "descriptors": view(ptrs[3], (m, 32), torch.uint8),
                "match_idx": view(ptrs[4], (m,), torch.int32), "match_dist": view(ptrs[5], (m,), torch.int32)}


def fbklt_track_batch(ctx: "Context", prevs, currs, pts, priors, num_levels=3, lanes=5, err_thresh=30.0, fb_dist=0.5, max_iters=30, eps=0.01):
    """alva_fbklt_track_batch: returns (list of tracked [n,2] f32 tensors, list of status [n] u8 tensors); priors are not modified."""
    n = len(prevs)
    outs = [p.clone() for p in priors]
    sts = [torch.empty((p.shape[0],), dtype=torch.uint8, device=p.device) for p in pts]
    arr = lambda v: (_vp * n)(*v)
    check(lib.alva_fbklt_track_batch(ctx.h, arr([p.h for p in prevs]), arr([p.h for p in currs]), n, num_levels, err_thresh, fb_dist, max_iters, eps,
                                     arr([_ptr(p) for p in pts]), arr([_ptr(o) for o in outs]), arr([_ptr(s_) for s_ in sts]),
                                     (_i * n)(*[p.shape[0] for p in pts]), lanes))
    return outs, sts


class TrackBatch:
    """alva_track_batch: trackMono (preprocessImage -> kltTracking -> computePose) of B lock-step cameras, one launch per stage."""

    def __init__(self, device: int, width: int, height: int, cameras: int, max_tracked: int, max_corr: int):
        import numpy as np
        h = _vp()
        check(lib.alva_track_batch_create(device, width, height, cameras, max_tracked, max_corr, C.byref(h)))
        self.h, self.device, self.B = h, device, cameras
        self.poses = np.zeros((cameras, 7))
        self.status = np.zeros(cameras, np.int32)
        self._n_pts = [0] * cameras
        self._args = None

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.alva_track_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def bind(self, pts, bearings, uv, wpts):
        """pointer tables of the per-camera inputs that stay put between frames (lists of cuda tensors, one per camera)"""
        B = self.B
        arr = lambda ts: (_vp * B)(*[_ptr(t) if t is not None and t.numel() else None for t in ts])
        self._args = (arr(pts), (_i * B)(*[0 if t is None else t.shape[0] for t in pts]), arr(bearings), arr(uv), arr(wpts),
                      (_i * B)(*[0 if t is None else t.shape[0] for t in bearings]))
        self._n_pts = [0 if t is None else t.shape[0] for t in pts]
        self._keep = (pts, bearings, uv, wpts)

    def step(self, rgbas, K):
        """rgbas: list of B [H,W,4] u8 cuda tensors (same pitch).  Returns (status[B], poses[B,7]) -- views, overwritten by the next step."""
        a_pts, a_np, a_bv, a_uv, a_wp, a_nc = self._args
        fr = (_vp * self.B)(*[_ptr(r) for r in rgbas])
        check(lib.alva_track_batch_step_detect(self.h, fr, rgbas[0].stride(0), a_pts, a_np, a_bv, a_uv, a_wp, a_nc, K[0], K[1], K[2], K[3],
                                               self.poses.ctypes.data, self.status.ctypes.data,
                                               self.nkp.ctypes.data if getattr(self, "nkp", None) is not None else None))
        return self.status, self.poses

    def enable_detector(self, orb_features: int = 2000):
        import numpy as np
        check(lib.alva_track_batch_enable_detector(self.h, orb_features))
        self.nkp = np.zeros(self.B, np.int32)
        self.cap = 4 * orb_features + 1024

    def detections(self, cam: int):
        """dict of torch views of one camera's detector results of the last step (keypoints [n,6], descriptors [n,32], match idx/dist [n])"""
        ptrs = [_vp() for _ in range(4)]
        check(lib.alva_track_batch_detections(self.h, cam, *[C.byref(p) for p in ptrs]))
        n = int(self.nkp[cam])
        return {"keypoints": _dev_view(ptrs[0], (n, 6), torch.float32, self.device), "descriptors": _dev_view(ptrs[1], (n, 32), torch.uint8, self.device),
                "match_idx": _dev_view(ptrs[2], (n,), torch.int32, self.device), "match_dist": _dev_view(ptrs[3], (n,), torch.int32, self.device)}

    def frame_table(self, rgbas):
        """pointer table of one set of B frames for step_table() (build once per resident frame set: a Python list comprehension
        over 64 tensors costs more than the whole GPU step)"""
        return (_vp * self.B)(*[_ptr(r) for r in rgbas]), rgbas[0].stride(0), rgbas

    def step_table(self, table, K):
        a_pts, a_np, a_bv, a_uv, a_wp, a_nc = self._args
        check(lib.alva_track_batch_step_detect(self.h, table[0], table[1], a_pts, a_np, a_bv, a_uv, a_wp, a_nc, K[0], K[1], K[2], K[3],
                                               self.poses.ctypes.data, self.status.ctypes.data,
                                               self.nkp.ctypes.data if getattr(self, "nkp", None) is not None else None))
        return self.status, self.poses

    def results(self, cam: int):
        """(tracked [n,2] f32, status [n] u8) of one camera: torch views of device memory, valid until the next step"""
        p, q = _vp(), _vp()
        check(lib.alva_track_batch_results(self.h, cam, C.byref(p), C.byref(q)))
        n = self._n_pts[cam]
        return _dev_view(p, (n, 2), torch.float32, self.device), _dev_view(q, (n,), torch.uint8, self.device)

    def ctx_handle(self):
        return lib.alva_track_batch_ctx(self.h)

    def set_klt_lanes(self, lanes: int):
        check(lib.alva_track_batch_set_klt_lanes(self.h, lanes))

    def stats(self):
        """(steps done, cameras re-solved through the single-camera call)"""
        a, b = C.c_long(0), C.c_long(0)
        check(lib.alva_track_batch_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


def _dev_view(ptr, shape, dtype, device):
    import numpy as np
    if int(np.prod(shape)) == 0:
        return torch.empty(shape, dtype=dtype, device=f"cuda:{device}")

    class _W:
        pass
    w = _W()
    typestr = {torch.float32: "<f4", torch.uint8: "|u1", torch.int32: "<i4"}[dtype]
    w.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr.value, False), "version": 2}
    return torch.as_tensor(w, device=f"cuda:{device}")


def build_pyramids_batch(ctx: "Context", pyrs, rgbas, grays=None):
    """alva_pyramid_build_from_rgba_batch: all cameras' gray + LK pyramid in five launches."""
    n = len(pyrs)
    ph = (_vp * n)(*[p.h for p in pyrs])
    pr = (_vp * n)(*[_ptr(r) for r in rgbas])
    pg = None if grays is None else (_vp * n)(*[_ptr(g) for g in grays])
    check(lib.alva_pyramid_build_from_rgba_batch(ctx.h, ph, pr, rgbas[0].stride(0), pg, 0 if grays is None else grays[0].stride(0), n))


def relpose_draw_samples(n_points: int, count: int, do_random: bool = False, seed: int = 12345):
    import numpy as np
    s = np.zeros((count, 8), np.int32)
    check(lib.alva_relpose_draw_samples(n_points, count, int(do_random), seed, s.ctypes.data))
    return s


def kernel_times(fn, reps: int):
    """Run fn() reps times with per-kernel HIP-event timing on; returns {kernel: (launches, average microseconds)}."""
    check(lib.alva_prof_enable(1))
    try:
        for _ in range(reps):
            fn()
    finally:
        check(lib.alva_prof_enable(0))
    buf = C.create_string_buffer(1 << 16)
    check(lib.alva_prof_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, us = line.split("\t")
        out[name] = (int(calls), float(us))
    return out


def frontend_run_many(fes, steps, warmup, frames, pts, bearings, uv, wpts, K):
    """fes: list of Frontend; frames: list (per stream) of [ring,H,W,4] u8 cuda tensors; pts/bearings/uv/wpts: per-stream tensors.
    Returns (seconds, accepted poses)."""
    n = len(fes)
    ring = frames[0].shape[0]
    arr = lambda ptrs: (_vp * len(ptrs))(*ptrs)
    a_fe = arr([f.h.value for f in fes])
    a_fr = arr([frames[s][k].data_ptr() for s in range(n) for k in range(ring)])
    a_pts, a_bv, a_uv, a_wp = (arr([t[s].data_ptr() for s in range(n)]) for t in (pts, bearings, uv, wpts))
    secs, acc = C.c_double(0), C.c_int(0)
    check(lib.alva_frontend_run_many(a_fe, n, steps, warmup, a_fr, ring, frames[0].stride(1), a_pts, pts[0].shape[0], a_bv, a_uv, a_wp,
                                     bearings[0].shape[0], K[0], K[1], K[2], K[3], C.byref(secs), C.byref(acc)))
    return secs.value, acc.value


import numpy as np  # noqa: E402  (MedoidStore builds its operation records with numpy)


class MedoidStore:
    """alva_medoid_store (include/alvaar_hip.h, f1): the map points' descriptor tables on the device, edited by replaying operation logs.
    ops: list of (mp_slot, op, kf, desc32 | None, rehash_to) in program order (op 0 add, 1 remove, 2 clear, 3 reset)."""
    OP = np.dtype([("op", "<i4"), ("kf", "<i4"), ("rehash_to", "<i4"), ("next", "<i4"), ("desc", "u1", 32), ("pad", "<i4", 4)])
    TABLE_SLOT = np.dtype([("key", "<i4"), ("next", "<i4"), ("dist", "<f4"), ("pad", "<i4"), ("desc", "u1", 32)])

    def __init__(self, ctx: "Context"):
        lib.alva_medoid_store_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.alva_medoid_store_destroy.argtypes = [C.c_void_p]
        lib.alva_medoid_store_destroy.restype = None
        lib.alva_medoid_replay.argtypes = [C.c_void_p, _i, C.c_void_p, _i, C.c_void_p, C.c_void_p, _i]
        lib.alva_medoid_export.argtypes = [C.c_void_p, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.alva_medoid_dump.argtypes = [C.c_void_p, _i, C.c_void_p, C.c_size_t]
        lib.alva_medoid_table_bytes.restype = C.c_size_t
        lib.alva_medoid_op_bytes.restype = C.c_size_t
        assert lib.alva_medoid_op_bytes() == self.OP.itemsize
        self.ctx = ctx
        self.h = C.c_void_p()
        check(lib.alva_medoid_store_create(ctx.h, C.byref(self.h)))

    def replay(self, ops, slots: int):
        rec = np.zeros(len(ops), self.OP)
        first, last, touched = {}, {}, []
        for i, (slot, op, kf, desc, rehash_to) in enumerate(ops):
            rec[i]["op"], rec[i]["kf"], rec[i]["rehash_to"], rec[i]["next"] = op, kf, rehash_to, -1
            if desc is not None:
                rec[i]["desc"] = desc
            if slot in last:
                rec[last[slot]]["next"] = i
            else:
                first[slot] = i
                touched.append(slot)
            last[slot] = i
        mp = np.array(touched, np.int32)
        fo = np.array([first[s] for s in touched], np.int32)
        check(lib.alva_medoid_replay(self.h, len(ops), rec.ctypes.data, len(mp), mp.ctypes.data, fo.ctypes.data, int(slots)))

    def export(self, slots):
        s = np.ascontiguousarray(slots, np.int32)
        desc, valid, info = np.zeros((len(s), 32), np.uint8), np.zeros(len(s), np.uint8), np.zeros((len(s), 3), np.int32)
        check(lib.alva_medoid_export(self.h, len(s), s.ctypes.data, desc.ctypes.data, valid.ctypes.data, info.ctypes.data))
        return desc, valid, info

    def dump(self, slot: int):
        """the raw table of one map point -> (medoid bytes, has medoid, [(key, dist)] in the container's iteration order, bucket count)"""
        n = lib.alva_medoid_table_bytes()
        raw = np.zeros(n, np.uint8)
        check(lib.alva_medoid_dump(self.h, int(slot), raw.ctypes.data, n))
        hdr = raw[:32].view(np.int32)   # head, free, count, nbkt, used, medoid_valid, overflow, medoid_kf
        cap = 48                        # alva_medoid::CAP (csrc/slam/medoid_table.hpp): 64-byte header | CAP slots | 59 bucket heads + pad
        assert n == 64 + cap * self.TABLE_SLOT.itemsize + 60 * 4, "medoid table layout changed"
        slots = raw[64:64 + cap * self.TABLE_SLOT.itemsize].view(self.TABLE_SLOT)
        out, s = [], int(hdr[0])
        while s != -1 and len(out) <= cap:
            out.append((int(slots[s]["key"]), float(slots[s]["dist"])))
            s = int(slots[s]["next"])
        return raw[32:64].copy(), bool(hdr[5]), out, int(hdr[3]), bool(hdr[6])

    def close(self):
        if self.h:
            lib.alva_medoid_store_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close
