#!/usr/bin/env python3
"""bench_detail.py -- the SECONDARY measurement lines (everything that is not the driver's headline): written to bench_detail.json by
`bench.py` (full run) or by running this file; never printed on stdout.

  system_720p         configs[4]'s geometry (1280x720, cell 15) through the same System surface
  system_streams      S independent alva::System sessions on one GPU, one host thread each
  system_group        S sessions through alva_system_group (fibers; lock-step launches shared across sessions)
  local_ba_batch      64 local-BA problems through alva_local_ba_batch (the batched Schur variant)
  two_view_init       the map-initialisation call (five-point RANSAC + refinement)
  batched_preprocess  gray + LK pyramid of 64 cameras in five launches
  track_mono_batch / frame_step_batch   a RIG of lock-step cameras through alva_track_batch_step
  config_1280x720     configs[2]: cv::ORB detectAndCompute(4000) + brute-force Hamming
  stage_list_driver / stage_us          round 1's headline (fixed correspondences) and its per-stage times
  cpu_stages          per-stage milliseconds of the reference's own L1 functions (oracle/_ref) at 1 thread / 8 callers
"""
from __future__ import annotations

import json
import time

import numpy as np
import torch

from bench_common import ROOT, W, H, NKP, HBM_PEAK_GBS, RING, STREAM_FRAMES, SYSTEM_CELL, stream_index, SystemJob, log


def bench_ba(*a, **k):
    import bench
    return bench.bench_ba(*a, **k)


def roofline_ba(*a, **k):
    import bench
    return bench.roofline_ba(*a, **k)


def make_keypoints(n: int, seed: int) -> np.ndarray:
    rng = np.random.RandomState(seed)
    # one point per 12-px grid cell (+jitter), inside the 31-px descriptor border
    gx, gy = np.meshgrid(np.arange(W // 12), np.arange(H // 12))
    pts = np.stack([gx.ravel() * 12 + 6, gy.ravel() * 12 + 6], 1).astype(np.float32)[:n]
    pts += rng.uniform(-2, 2, pts.shape).astype(np.float32)
    return np.clip(pts, [32, 32], [W - 33, H - 33]).astype(np.float32)


class FrameJob:
    """Everything one stream needs, resident on one GPU."""

    def __init__(self, device: int, seed: int, own_stream: bool = False):
        import alvaar_amd
        from alvaar_amd import synth
        self.dev = torch.device("cuda", device)
        self.tstream = torch.cuda.Stream(device) if own_stream else None
        if own_stream:
            torch.cuda.set_stream(self.tstream)   # per-thread current stream: torch allocations/copies follow it
        self.ctx = alvaar_amd.Context(device, stream=self.tstream.cuda_stream if own_stream else None)
        frames = synth.stream_rgba(W, H, RING, seed=seed, noise=True)
        self.frames = torch.from_numpy(frames).to(self.dev)
        self.pyr = [alvaar_amd.Pyramid(self.ctx, W, H, 9, 3) for _ in range(2)]
        self.gray = torch.empty((H, W), dtype=torch.uint8, device=self.dev)
        self.pts = torch.from_numpy(make_keypoints(NKP, seed)).to(self.dev)
        pb = synth.make_pnp_problem(NKP, seed, outlier_frac=0.1, pose_noise=0.01)
        self.bv = torch.from_numpy(pb["bv"]).to(self.dev)
        self.wpt = torch.from_numpy(pb["wpt"]).to(self.dev)
        self.uv = torch.from_numpy(pb["uv"]).to(self.dev)
        self.K = pb["K"]
        self.pose0 = pb["pose_init"]
        self.k = 0
        self.orb = alvaar_amd.Orb(self.ctx, W, H, 2000)
        # second lane (own non-blocking HIP stream) for the detector: it only needs the gray image, not the tracker's output
        self.lane_b = alvaar_amd.Context(device, own_stream=True)
        cap = 4 * 2000 + 1024
        self.kp_buf = [torch.zeros((cap, 6), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.desc_buf = [torch.zeros((cap, 32), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.match = None
        self.fe = alvaar_amd.Frontend(device, W, H, NKP, 2000)   # native per-frame driver (same stages, host side in C++)
        self.maxq = 0.001
        self._det = torch.zeros((NKP, 2), dtype=torch.float32, device=self.dev)
        # prime: frame 0 pyramid + descriptors
        self.pyr[0].build_from_rgba(self.frames[0], self.gray)
        self.prev_desc, _ = self.ctx.describe(self.gray, self.pts)
        torch.cuda.synchronize(self.dev)

    def det_buf(self, det):
        """fixed-size keypoint buffer for the describe/match stages (detections, padded with the grid points)"""
        n = min(det.shape[0], NKP)
        self._det[:n] = det[:n]
        if n < NKP:
            self._det[n:] = self.pts[n:]
        return self._det

    def step_native(self, lookahead: bool = True):
        """The frame through alva_frontend_track_ahead: the same stage calls as step_overlapped(), issued from C++.  With
        look-ahead the NEXT frame of the resident ring has its gray image + pyramid built on a third stream meanwhile (every
        step still builds exactly one pyramid)."""
        self.k += 1
        nxt = self.frames[(self.k + 1) % RING] if lookahead else None
        st, pose, nkp = self.fe.track(self.frames[self.k % RING], self.pts, self.bv, self.uv, self.wpt, self.K, rgba_next=nxt)
        return st == 2

    def step_overlapped(self):
        """Same work as step(): ORB + matching run on lane B while fb-KLT + pose run on lane A (one frame, two HIP streams)."""
        ctx, lb = self.ctx, self.lane_b
        self.k += 1
        cur, prev = self.pyr[self.k % 2], self.pyr[(self.k - 1) % 2]
        cur.build_from_rgba(self.frames[self.k % RING], self.gray)                    # a2 + a3   (lane A)
        lb.wait_for(ctx)
        tracked, status = ctx.fbklt_track(prev, cur, self.pts, self.pts, 3)           # a4        (lane A)
        ctx.compute_pose_enqueue(self.bv, self.uv, self.wpt, self.K)                  # a8 + a9   (lane A, no host wait)
        self.orb.enqueue(self.gray, self.kp_buf[self.k % 2], self.desc_buf[self.k % 2], ctx=lb)   # a5' + a6 (lane B, no host wait)
        st, pose, m1, m2 = ctx.compute_pose_collect()                                  # host result (lane A)
        kp, desc = self.orb.collect()                                                  # count -> host (lane B)
        self.match = lb.bf_match_hamming(desc, self.prev_desc)                         # a7        (lane B)
        self.prev_desc = desc   # (double-buffered; lane B orders the next frame's detector after this match)
        return st == 2

    def step(self, grid_detector: bool = False):
        ctx = self.ctx
        self.k += 1
        cur, prev = self.pyr[self.k % 2], self.pyr[(self.k - 1) % 2]
        cur.build_from_rgba(self.frames[self.k % RING], self.gray)                    # a2 + a3
        tracked, status = ctx.fbklt_track(prev, cur, self.pts, self.pts, 3)           # a4
        if grid_detector:
            det, self.maxq = ctx.detect_grid(self.gray, 12, max_quality=self.maxq)     # a5 (count -> host)
            desc, valid = ctx.describe(self.gray, self.det_buf(det))                   # a6
        else:
            kp, desc = self.orb.detect_and_compute(self.gray)                          # a5' + a6 (count -> host)
        idx, dist = ctx.bf_match_hamming(desc, self.prev_desc)                         # a7
        # a8 + a9 as VisualFrontend::computePose chains them (P3P -> drop outliers -> PnP), one host sync
        st, pose, m1, m2 = ctx.compute_pose(self.bv, self.uv, self.wpt, self.K)
        self.prev_desc = desc
        return st == 2

    # ---- per-stage HIP-event timing (not part of the timed region) ----
    def stage_times(self, reps: int = 20):
        ctx = self.ctx
        cur, prev = self.pyr[0], self.pyr[1]
        prev.build_from_rgba(self.frames[1], self.gray)
        tracked, _ = ctx.fbklt_track(prev, cur, self.pts, self.pts, 3)
        desc, _ = ctx.describe(self.gray, tracked)
        stages = {
            "orb_detect_and_compute": lambda: self.orb.detect_and_compute(self.gray),
            "detect_grid": lambda: ctx.detect_grid(self.gray, 12, max_quality=0.001),
            "gray+pyramid": lambda: cur.build_from_rgba(self.frames[2], self.gray),
            "fbklt": lambda: ctx.fbklt_track(prev, cur, self.pts, self.pts, 3),
            "describe(blur7+brief)": lambda: ctx.describe(self.gray, tracked),
            "bf_hamming": lambda: ctx.bf_match_hamming(desc, self.prev_desc),
            "p3p_lmeds": lambda: ctx.p3p_lmeds(self.bv, self.wpt, 100, 3.0, self.K[0], self.K[1]),
            "pnp_refine": lambda: ctx.pnp_refine(self.uv, self.wpt, self.pose0, self.K),
            "compute_pose(p3p->pnp)": lambda: ctx.compute_pose(self.bv, self.uv, self.wpt, self.K),
        }
        out = {}
        for name, fn in stages.items():
            fn()
            torch.cuda.synchronize(self.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(self.dev)
            out[name] = e0.elapsed_time(e1) / reps * 1e3  # us
        return out


def bench_system_streams(device: int, n_streams: int, steps: int = 300):
    """S independent camera sessions on ONE GPU: S alva::System objects (each its own HIP stream, pyramids, map), one host thread each,
    all fed the same resident frames.  A single session leaves the GPU idle most of the time (every kernel of its chain is latency-
    bound) and its host-side map layer runs on one core; sessions are independent, so they overlap.  Aggregate frames/s."""
    import threading
    jobs = [SystemJob(device, 7, host_copy=False) if i == 0 else None for i in range(n_streams)]
    for i in range(1, n_streams):   # share the resident frames (read-only); every session has its own System
        j = SystemJob.__new__(SystemJob)
        j.__dict__.update(jobs[0].__dict__)
        from alvaar_amd.system import AlvaAR
        j.ar = AlvaAR(W, H, device=device, cell_size=SYSTEM_CELL, random_sampling=False)
        j.k = -1
        j.status_hist = [0, 0, 0, 0]
        jobs[i] = j
    for j in jobs:   # past the initialisation, into the steady state (30-keyframe window full), like the headline
        j.warm_to_steady_state()
    start = threading.Barrier(n_streams + 1)
    done = []

    def run(j):
        start.wait()
        for _ in range(steps):
            j.step()
        done.append(time.perf_counter())
    th = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = max(done) - t0
    tracked = sum(j.status_hist[1] for j in jobs)
    for j in jobs:
        j.ar.close()
    return {"sessions": n_streams, "frames_per_s": n_streams * steps / dt, "frames_per_s_per_session": steps / dt, "steps_per_session": steps,
            "tracked_frac": tracked / max(sum(sum(j.status_hist) for j in jobs), 1),
            "note": "S independent alva::System sessions on one GPU, one host thread each (Python threads; the C call releases the GIL), frames resident in HBM"}


def _cpu_throttled():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines()) if k in ("nr_throttled", "throttled_usec", "usage_usec")}
    except Exception:
        return {}


def bench_system_group(device: int, n_sessions: int, n_threads: int, steps: int = 200, n_streams: int = 0, lockstep: bool = True, lanes: int = 2,
                       stagger: int = 0):
    """S independent alva::System sessions on ONE GPU through alva_system_group: W host threads, the sessions as fibers -- a session's
    waits for the GPU run the thread's other sessions, so the threads execute map-layer work only.  Session i runs on worker i % W and
    belongs to lane (i + i // W) % lanes: with lock-step launches (include/alvaar_system.h) the seven launches of a lane's tracking frames
    are issued once per kind for all of its sessions, on the lane's stream, while the workers do the host half of the other lanes'
    sessions; every session keeps a stream of its own for its keyframe stages (n_streams = 0; k > 0: session i on shared stream
    (i % W) % k).  stagger = 0: all sessions replay the same resident stream in lock-step (keyframes coincide on one group step: the
    worst case); stagger = d: session i is d * i frames ahead in the stream, so the sessions' keyframes spread over the keyframe period
    as those of independent cameras do.  Aggregate frames/s in the steady state."""
    from alvaar_amd.system import AlvaAR, SystemGroup
    base = SystemJob(device, 7, host_copy=False)
    group = SystemGroup([], n_threads)
    group.set_lockstep(lockstep)
    group.set_lanes(lanes)
    if n_streams > 0:
        base.ar.close()
        sessions = [AlvaAR(W, H, device=device, cell_size=SYSTEM_CELL, random_sampling=False, hip_stream=group.stream((i % n_threads) % n_streams, device))
                    for i in range(n_sessions)]
        base.ar = sessions[0]
    else:
        sessions = [base.ar] + [AlvaAR(W, H, device=device, cell_size=SYSTEM_CELL, random_sampling=False) for _ in range(n_sessions - 1)]
    group.set_sessions(sessions)
    k = 0

    def step():
        nonlocal k
        st = group.step_device([base.ptrs[stream_index(k + stagger * i)] for i in range(n_sessions)], 33.0 * k)
        k += 1
        return st
    while int(base.ar.state()[11]) < 34 and k < 2500:   # steady state: the 30-keyframe window full
        step()
    l0, c0 = group.launch_stats()
    group.time_stats()
    thr0 = _cpu_throttled()
    kf0 = int(base.ar.state()[11])
    t0 = time.perf_counter()
    tracked = 0
    step_ms, kf_seen = [], kf0
    for _ in range(steps):
        ts = time.perf_counter()
        tracked += int((step() == 1).sum())
        kf_now = int(base.ar.state()[11])
        step_ms.append(((time.perf_counter() - ts) * 1e3, kf_now != kf_seen))
        kf_seen = kf_now
    dt = time.perf_counter() - t0
    trk = sorted(m for m, is_kf in step_ms if not is_kf)
    kfs = [m for m, is_kf in step_ms if is_kf]
    l1, c1 = group.launch_stats()
    t_run, t_work, n_slices, n_work = group.time_stats()
    thr1 = _cpu_throttled()
    kf = int(base.ar.state()[11]) - kf0
    for s in sessions:
        s.close()
    group.close()
    return {"sessions": n_sessions, "host_threads": n_threads, "session_streams": n_streams or n_sessions, "lockstep": lockstep,
            "lanes": lanes if lockstep else 0,
            "frames_per_s": n_sessions * steps / dt, "ms_per_group_step": dt / steps * 1e3, "keyframes_per_session": kf,
            "tracked_frac": tracked / (n_sessions * steps), "stagger_frames": stagger,
            "ms_tracking_step_median": round(trk[len(trk) // 2], 3) if trk else None, "ms_keyframe_step_mean": round(sum(kfs) / len(kfs), 3) if kfs else None,
            "keyframe_steps": len(kfs),
            "worker_busy_frac": round(t_work / max(t_run, 1e-9), 3), "host_work_us_per_frame": round(1e6 * t_work / (n_sessions * steps), 1),
            "worker_slices_per_frame": round(n_slices / (n_sessions * steps), 1),
            "cgroup_cpu": {k_: thr1.get(k_, 0) - thr0.get(k_, 0) for k_ in thr1}, "process_cpu_cores_used": round((thr1.get("usage_usec", 0) - thr0.get("usage_usec", 0)) / (dt * 1e6), 2) if thr1 else None,
            "chain_launches_issued_per_group_step": round((l1 - l0) / steps, 2), "session_launches_carried_per_group_step": round((c1 - c0) / steps, 2),
            "note": "alva_system_group: sessions are fibers on the worker threads (a wait for the GPU switches to the thread's next session); "
                    "frames resident in HBM, every session its own map / pyramids / kernels' work; lockstep: one launch per kernel kind for the "
                    "sessions of a lane (lane.hpp)"}


def bench_multi_stream(device: int, n_streams: int, steps: int, warmup: int = 5):
    """S independent camera streams on ONE GPU, each with its own alva_frontend (two HIP streams) and its own host thread
    inside the library (alva_frontend_run_many).  Every stage of a single stream is latency-bound at these sizes, so
    concurrent streams fill the idle CUs."""
    import alvaar_amd
    from alvaar_amd import capi, synth
    dev = torch.device("cuda", device)
    fes, frames, pts, bv, uv, wp = [], [], [], [], [], []
    K = None
    for s in range(n_streams):
        fes.append(alvaar_amd.Frontend(device, W, H, NKP, 2000))
        frames.append(torch.from_numpy(synth.stream_rgba(W, H, RING, seed=7 + s, noise=True)).to(dev))
        pts.append(torch.from_numpy(make_keypoints(NKP, 7 + s)).to(dev))
        pb = synth.make_pnp_problem(NKP, 7 + s, outlier_frac=0.1, pose_noise=0.01)
        bv.append(torch.from_numpy(pb["bv"]).to(dev))
        uv.append(torch.from_numpy(pb["uv"]).to(dev))
        wp.append(torch.from_numpy(pb["wpt"]).to(dev))
        K = pb["K"]
    torch.cuda.synchronize()
    wall, accepted = capi.frontend_run_many(fes, steps, warmup, frames, pts, bv, uv, wp, K)
    for f in fes:
        f.close()
    return {"streams": n_streams, "frames_per_s": n_streams * steps / wall, "ms_per_frame_per_stream": wall / steps * 1e3,
            "poses_accepted": accepted, "frames": n_streams * steps}


def bench_720p(device: int, reps: int = 50, valu_peak_tops: float | None = None):
    """BASELINE configs[2]: 1280x720, ORB extract 4000 kp/frame + brute-force Hamming match (secondary line)."""
    import alvaar_amd
    from alvaar_amd import synth, capi
    w, h = 1280, 720
    ctx = alvaar_amd.Context(device)
    frames = torch.from_numpy(synth.stream_rgba(w, h, 2, seed=11, noise=True)).to(f"cuda:{device}")
    gray = [ctx.rgba2gray(frames[k]) for k in range(2)]
    orb = alvaar_amd.Orb(ctx, w, h, 4000)
    cap = 4 * 4000 + 1024
    bufs = [(torch.zeros((cap, 6), dtype=torch.float32, device=gray[0].device), torch.zeros((cap, 32), dtype=torch.uint8, device=gray[0].device))
            for _ in range(2)]
    orb.enqueue(gray[0], *bufs[0])
    _, prev = orb.collect()

    def step(k=[0]):
        k[0] += 1
        orb.enqueue(gray[k[0] & 1], *bufs[k[0] & 1])
        kp, desc = orb.collect()
        step.match = ctx.bf_match_hamming(desc, step.prev)
        step.prev = desc
        step.n = desc.shape[0]
    step.prev = prev
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kt = capi.kernel_times(step, 20)
    P2 = w * h
    algb = {"k_fast_nms": 3.27 * P2, "k_blur7_multi": 2 * 3.27 * P2, "k_pyramid": (1 + 3.27) * P2, "k_bf_partial": 32 * 2 * step.n + 8 * step.n * ((step.n + 63) // 64)}
    ham = None
    if "k_bf_partial" in kt and valu_peak_tops:
        ops = 24.0 * step.n * step.n           # per pair 8 xor + 8 popcount + 8 add on 32-bit words
        us = kt["k_bf_partial"][1]
        ham = {"kernel": "k_bf_partial", "ops": ops, "avg_us": round(us, 2), "achieved_Tops": round(ops / (us * 1e-6) / 1e12, 2),
               "peak_Tops_measured": round(valu_peak_tops, 1), "valu_frac": ops / (us * 1e-6) / 1e12 / valu_peak_tops,
               "note": "integer VALU bound, not HBM (288 KB of descriptors); queries live in registers, 64 train rows per LDS tile, no cross-lane reduction"}
    return {"workload": "configs[2]: 1280x720, cv::ORB detectAndCompute(4000, 1.2, 8) + BFMatcher(HAMMING) vs the previous frame",
            "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "keypoints": int(step.n), "hamming_valu": ham,
            "kernels": {k: {"avg_us": round(v[1], 2), "launches_per_frame": round(v[0] / 20, 2),
                            **({"GBps": round(algb[k] / (v[1] * 1e-6) / 1e9, 1)} if k in algb else {})}
                        for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8]}}


def bench_ba_batch(ctx, pb, peaks, problems: int = 64, reps: int = 3):
    """SURVEY.md 8(d) "BA Schur reduce ... report a batched variant (>= 64 problems)": `problems` independent 20 KF x 3000 pts local-BA
    problems through alva_local_ba_batch (one set of launches per LM iteration for all of them; every problem bit-identical to its own
    alva_local_ba).  The problems are the SURVEY instance with independently perturbed inverse depths and observations."""
    from alvaar_amd import capi
    rng = np.random.RandomState(5)
    pbs = []
    for b in range(problems):
        q = dict(pb)
        q["inv_depth"] = pb["inv_depth"] * (1.0 + 1e-3 * rng.randn(len(pb["inv_depth"])))
        q["obs_uv"] = pb["obs_uv"] + 0.05 * rng.randn(*np.asarray(pb["obs_uv"]).shape)
        pbs.append(q)
    ctx.local_ba_batch(pbs, 5, 0.0)   # warm (scratch allocation)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = ctx.local_ba_batch(pbs, 5, 0.0)
    dt = (time.perf_counter() - t0) / reps
    kt = capi.kernel_times(lambda: ctx.local_ba_batch(pbs, 5, 0.0), 1)
    nobs = len(pb["obs_kf"])
    iters = [int(r["info"][0]) - 1 for r in res]
    work = nobs * sum(iters)
    nfree = int((np.asarray(pb["kf_const"]) == 0).sum())
    m = ((6 * nfree + 1 + 15) // 16) * 16
    out = dict(problems=problems, residual_blocks_per_problem=nobs, lm_iterations=iters[:4] + ["..."], ms_per_batch=dt * 1e3,
               residual_block_iters_per_s=work / dt, kernel_us_per_batch=round(sum(c * u for c, u in kt.values()), 1), kernels={})
    for name, (calls, us) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8]:
        e = {"launches": calls, "avg_us": round(us, 1)}
        if name.startswith("k_gemm"):
            fl = 2.0 * m * m * len(pb["anchor_kf"]) * problems
            e.update(bound="mfma_f64", flops_per_launch=int(fl), achieved_TFLOPs=round(fl / (us * 1e-6) / 1e12, 2), peak_TFLOPs_measured=round(peaks[0], 1),
                     frac_of_measured=fl / (us * 1e-6) / 1e12 / peaks[0])
        if name.startswith("k_point"):
            by = (60 + 112) * nobs * problems
            e.update(bound="hbm", alg_bytes_per_launch=int(by), achieved_GBps=round(by / (us * 1e-6) / 1e9, 1), frac=by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        out["kernels"][name] = e
    out["note"] = ("whole alva_local_ba_batch calls (host structure build of every problem, one upload, the LM loop with one scalar read-back per iteration, "
                   "results back); launches above are per batch and cover all problems")
    return out


def bench_batched_preprocess(device: int, cameras: int = 64, reps: int = 20):
    """Secondary line for the roofline discussion: gray + LK pyramid of `cameras` 640x480 frames in FIVE launches
    (alva_pyramid_build_from_rgba_batch).  One frame per launch is launch-latency-bound (roofline.frac ~ 0.005); this shows what
    the same kernels reach when a launch carries enough bytes.  Algorithmic bytes per camera: 5 P (RGBA -> gray) + 6.64 P
    (pyramid + Scharr), SURVEY.md 8(d)."""
    import alvaar_amd
    from alvaar_amd import capi, synth
    dev = torch.device("cuda", device)
    ctx = alvaar_amd.Context(device, own_stream=True)
    base = torch.from_numpy(synth.stream_rgba(W, H, 4, seed=5, noise=True)).to(dev)
    frames = [base[c % 4].clone() for c in range(cameras)]
    grays = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(cameras)]
    pyrs = [alvaar_amd.Pyramid(ctx, W, H, 9, 3) for _ in range(cameras)]
    capi.build_pyramids_batch(ctx, pyrs, frames, grays)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.build_pyramids_batch(ctx, pyrs, frames, grays)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    kt = capi.kernel_times(lambda: capi.build_pyramids_batch(ctx, pyrs, frames, grays), 5)
    ctx.sync()
    kernel_us = sum(v[0] / 5 * v[1] for v in kt.values())
    alg = cameras * (5 + 6.64) * W * H
    for p in pyrs:
        p.close()
    return dict(cameras=cameras, launches=5, ms_per_batch=dt * 1e3, frames_per_s=cameras / dt, kernel_us_per_batch=kernel_us,
                alg_bytes_per_batch=int(alg), achieved_GBps=alg / (kernel_us * 1e-6) / 1e9, hbm_frac=alg / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                kernels={k: {"avg_us": round(v[1], 2), "launches_per_batch": round(v[0] / 5, 2)} for k, v in kt.items()},
                note="event-timed kernels of alva_pyramid_build_from_rgba_batch; achieved = algorithmic bytes / sum of kernel times")


def bench_track_mono_batch(device: int, cameras: int = 64, reps: int = 10, seed: int = 7, detector: bool = False, orb_features: int = 2000):
    """Secondary lines: `cameras` lock-step cameras through alva_track_batch_step.  detector=False ("track_mono_batch"):
    VisualFrontend::trackMono (preprocessImage -> kltTracking -> computePose; the detector belongs to the keyframe branch).
    detector=True ("frame_step_batch"): the headline's full stage list per camera -- the above plus cv::ORB detectAndCompute(2000) and
    the Hamming match against the camera's previous descriptors -- i.e. B times the work of one alva_frontend_track.
    Through alva_track_batch_step -- 10 launches and one synchronisation per lane for ALL cameras.
    Every camera has its own frame ring (4 distinct synthetic streams, cycled), 2120 keypoints and 2120 correspondences.
    Algorithmic HBM bytes per camera frame: 4P RGBA in + 7.64P pyramid/Scharr (no separate gray copy) + the KLT gathers, which stay
    in L2 and are not counted => 11.64 P."""
    import alvaar_amd
    from alvaar_amd import capi, synth
    dev = torch.device("cuda", device)
    nsrc = min(cameras, 4)
    rings = [torch.from_numpy(synth.stream_rgba(W, H, RING, seed=seed + s, noise=True)).to(dev) for s in range(nsrc)]
    pts, bv, uv, wp = [], [], [], []
    for s in range(nsrc):
        pb = synth.make_pnp_problem(NKP, seed + s, outlier_frac=0.1, pose_noise=0.01)
        pts.append(torch.from_numpy(make_keypoints(NKP, seed + s)).to(dev))
        bv.append(torch.from_numpy(pb["bv"]).to(dev))
        uv.append(torch.from_numpy(pb["uv"]).to(dev))
        wp.append(torch.from_numpy(pb["wpt"]).to(dev))
        K = pb["K"]
    # every camera owns its frames (no two cameras read the same HBM lines)
    frames = [rings[c % nsrc].clone() for c in range(cameras)]
    tb = alvaar_amd.TrackBatch(device, W, H, cameras, NKP, NKP)
    if detector:
        tb.enable_detector(orb_features)   # + cv::ORB detectAndCompute(2000) and the Hamming match per camera: the headline's full stage list
    tb.bind([pts[c % nsrc] for c in range(cameras)], [bv[c % nsrc] for c in range(cameras)], [uv[c % nsrc] for c in range(cameras)],
            [wp[c % nsrc] for c in range(cameras)])
    tables = [tb.frame_table([f[r] for f in frames]) for r in range(RING)]   # the resident frames' pointer tables, built once
    k = 0

    def step():
        nonlocal k
        k += 1
        return tb.step_table(tables[k % RING], K)
    for _ in range(3):
        st, _ = step()
    torch.cuda.synchronize(dev)
    ok = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        st, _ = step()
        ok += int((st == 2).sum())
    dt = (time.perf_counter() - t0) / reps
    kt = capi.kernel_times(step, 3)
    kernel_us = sum(v[0] / 3 * v[1] for v in kt.values())
    alg = cameras * (11.64 + (1 + 2 * 3.27 + 2 * 3.27 if detector else 0)) * W * H   # + gray copy, ORB pyramid w+r, blur r+w (L8 = 3.27 P)
    steps_done, fallbacks = tb.stats()
    tb.close()
    traffic = None   # HBM bytes per step from the PMC counters of the 64-camera step with the detector lane (two --pmc passes, tools/frame_step_pmc.py)
    tfile = ROOT / "profiles" / "r2_pmc_traffic_frame_step64.json"
    if detector and cameras == 64 and tfile.exists():
        per_step = {"k_pyr_stage_batch": 4, "k_resize_b": 7}
        traffic = int(sum(v["hbm_bytes_per_launch"] * per_step.get(k, 1) for k, v in json.loads(tfile.read_text())["kernels"].items()
                          if "rocclr" not in k))
    return dict(cameras=cameras, detector=detector, launches_per_step=24 if detector else 10, ms_per_step=dt * 1e3, frames_per_s=cameras / dt, poses_accepted_frac=ok / (reps * cameras),
                single_camera_fallbacks=fallbacks, kernel_us_per_step=kernel_us, alg_bytes_per_step=int(alg), hbm_traffic_bytes_per_step_pmc=traffic,
                achieved_GBps=alg / dt / 1e9, hbm_frac=alg / dt / 1e9 / HBM_PEAK_GBS,
                kernels={n: {"avg_us": round(v[1], 2), "launches_per_step": round(v[0] / 3, 2)} for n, v in kt.items()},
                note="whole alva_track_batch_step calls (pointer tables, argument copy, 10 launches in stream order (+ 14 of the detector lane on a second stream), one synchronisation per stream, pose decode); "
                     "achieved = algorithmic image bytes / wall time of the step, not / kernel time")


def bench_two_view_init(ctx, reps: int = 10):
    """§8f-2 secondary line: the map-initialisation call (compute5ptEssentialMatrix) on 2000 correspondences, 25 % mismatches."""
    import torch
    from alvaar_amd import synth, capi
    p = synth.make_relpose_problem(2000, 8, 0.25)
    b1, b2 = torch.from_numpy(p["bv1"]).cuda(), torch.from_numpy(p["bv2"]).cuda()
    ctx.compute_5pt_essential(b1, b2)  # warm
    t0 = time.perf_counter()
    for _ in range(reps):
        ok, R, t, mask, info = ctx.compute_5pt_essential(b1, b2)
    dt = (time.perf_counter() - t0) / reps
    kt = capi.kernel_times(lambda: ctx.compute_5pt_essential(b1, b2), 5)
    return dict(correspondences=2000, ok=bool(ok), ransac_iterations=int(info.iterations), inliers=int(info.n_inliers),
                lm_iterations=int(info.lm_iterations), ms_per_call=dt * 1e3, calls_per_s=1.0 / dt,
                kernels={k: {"avg_us": round(v[1], 2), "launches_per_call": round(v[0] / 5, 2)} for k, v in kt.items()},
                rotation_error_vs_truth=float(np.abs(R - p["R12"]).max()),
                note="whole alva_compute_5pt_essential call: host sample draw, 112 five-point hypotheses, adaptive-loop replay, "
                     "on-device Levenberg-Marquardt refinement, one stream synchronisation")


def cpu_stage_table(width: int, height: int, cell: int, orb_features: int, seed: int, budget_s: float = 4.0):
    """SURVEY.md 8(d) "CPU baseline timing (2)": per-stage milliseconds of the reference's own L1 functions / vendored OpenCV, OpenGV and
    Ceres calls (oracle/_ref: FeatureExtractor::detectFeaturePoints feature_extractor.cpp:11-158, describeFeaturePoints :160-214,
    FeatureTracker::fbKltTracking feature_tracker.cpp:5-111, MultiViewGeometry::p3pRansac / ceresPnP multi_view_geometry.cpp:24-223,
    cv::cvtColor, cv::buildOpticalFlowPyramid, cv::BFMatcher, cv::ORB::detectAndCompute) on this box's host cores, same synthetic
    frames as the GPU path.  "ms_1_thread" = median over the repetitions on one core.  The reference build has NO intra-call
    threading (wasm, single-threaded; OpenCV without a parallel backend, Ceres NO_THREADS -- as shipped), so "8 threads" means 8
    independent callers: "ms_8_callers" is the wall time per call when 8 host threads each run the stage on their own data."""
    import threading
    import oracles
    from alvaar_amd import synth
    R = oracles.Ref
    canvas = synth.texture_canvas(width, height, seed)
    rgba = [synth.gray_to_rgba(synth.frame_gray(canvas, k, width, height, noise_seed=11)) for k in (0, 1, 5)]
    gray = [R.rgba2gray(f) for f in rgba]
    pts, _ = R.detect_grid(gray[0], cell)
    n = len(pts)
    d0, _ = R.describe(gray[0], pts)
    d5, _ = R.describe(gray[2], pts)
    pb = synth.make_pnp_problem(n, seed, outlier_frac=0.1, pose_noise=0.01)
    stages = {
        "cvtColor(RGBA2GRAY)": lambda: R.rgba2gray(rgba[1]),
        "buildOpticalFlowPyramid(9x9, 3)": lambda: R.build_pyramid(gray[1]),
        "detectFeaturePoints": lambda: R.detect_grid(gray[1], cell),
        "describeFeaturePoints": lambda: R.describe(gray[1], pts),
        "fbKltTracking(3 levels)": lambda: R.fbklt(gray[0], gray[1], pts, pts, 3),
        "BFMatcher(HAMMING) NxN": lambda: R.bf_match(d0, d5),
        "p3pRansac(100 it)": lambda: R.p3p_lmeds(pb["bv"], pb["wpt"], fx=pb["K"][0], fy=pb["K"][1]),
        "ceresPnP": lambda: R.pnp_refine(pb["uv"], pb["wpt"], pb["pose_init"], pb["K"]),
        f"cv::ORB::detectAndCompute({orb_features})": lambda: R.orb(gray[1], orb_features),
    }
    out = {}
    per = budget_s / len(stages)
    for name, fn in stages.items():
        t0 = time.perf_counter()
        fn()
        first = time.perf_counter() - t0
        reps = int(min(20, max(3, 0.5 * per / max(first, 1e-6))))
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ms1 = float(np.median(ts)) * 1e3
        reps8 = max(2, reps // 3)
        go = threading.Barrier(9)

        def run():
            go.wait()
            for _ in range(reps8):
                fn()
        th = [threading.Thread(target=run) for _ in range(8)]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        ms8 = (time.perf_counter() - t0) / (8 * reps8) * 1e3
        out[name] = {"ms_1_thread": round(ms1, 3), "ms_8_callers": round(ms8, 3), "reps": reps}
    return {"geometry": f"{width}x{height}, cell {cell}", "keypoints": n, "stages": out}


def run_system_line(local: int, seed: int, width: int, height: int, cell: int, steps: int):
    """A secondary System line (default-on for configs[4]'s geometry): steady-state warm-up, then >= 0.5 s of the resident-frame loop."""
    job = SystemJob(local, seed, host_copy=False, width=width, height=height, cell=cell)
    extra, period = job.warm_to_steady_state()
    torch.cuda.synchronize()
    kf0 = int(job.ar.state()[11])
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(steps):
            job.step()
        n += steps
        if time.perf_counter() - t0 > 0.5:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = job.ar.state()
    job.ar.timing(); job.ar.timing_keyframe()
    for _ in range(200):
        job.step()
    kfd = int(job.ar.state()[11])
    sec, kfsec = job.ar.timing(), job.ar.timing_keyframe()
    nk = max(kfd - int(st[11]), 1)
    out = {"workload": f"{width}x{height} RGBA stream, cell {cell}, alva_system_find_camera_pose_device, frames resident in HBM",
           "frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "steps": n, "keyframes_in_region": int(st[11]) - kf0,
           "untimed_frames_to_steady_state": extra, "keyframe_period_frames": period,
           "keypoints_per_frame": int(st[2]), "keypoints_3d": int(st[4]), "keyframes_in_map": int(st[6]), "map_points": int(st[7]),
           "ms_per_keyframe": round(1e3 * (sec["keyframe_create"] + sec["mapping"]) / nk, 3),
           "tracking_frame_us": round(1e6 * sum(v for k_, v in sec.items() if k_ not in ("keyframe_create", "mapping")) / 200, 1)}
    job.ar.close()
    del job
    torch.cuda.empty_cache()
    return out



def stage_list_driver(local: int, seed: int, steps: int):
    """round 1's headline (fixed correspondences, three HIP streams): an upper bound of stage throughput, not the reference's dataflow"""
    job = FrameJob(local, seed=seed)

    def timed(fn, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    dt_drv = timed(job.step_native, 10)
    dt_nola = timed(lambda: job.step_native(lookahead=False), 3)
    dt_serial = timed(job.step, 3)
    return {"frames_per_s": steps / dt_drv, "ms_per_step": dt_drv / steps * 1e3, "no_lookahead_frames_per_s": steps / dt_nola,
            "one_hip_stream_frames_per_s": steps / dt_serial,
            "note": "the configs[1] stage list (gray, pyramid, fb-KLT 3 levels, cv::ORB detectAndCompute 2000, BF Hamming, P3P -> PnP) through "
                    "alva_frontend_track_ahead on three HIP streams with FIXED pose correspondences"}, job.stage_times()


def run_secondary(local: int, seed: int, steps: int, bctx, ba_pb, peaks, is720: bool = False, with_cpu: bool = True,
                  group_sessions=(8, 16, 32, 64), part: str = "all") -> dict:
    """Every secondary line, each guarded: a failing line is recorded as {"error": ...} and never costs the others.
    part = "line": only the three figures bench.py's compact line carries (32 sessions, the 1280x720 System stream, configs[2]);
    part = "rest": everything else (the group sweep keeps its 32-session entry: a second sample of the same figure)."""
    out = {}

    def guarded(name, fn):
        log(f"secondary: {name}")
        try:
            out[name] = fn()
        except Exception as e:   # noqa: BLE001 -- a secondary line must not take the record down
            out[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if part == "line":
        guarded("system_group32", lambda: bench_system_group(local, 32, 16, lanes=4))
        if not is720:
            guarded("system_720p", lambda: run_system_line(local, seed, 1280, 720, 15, steps))
        guarded("config_1280x720", lambda: bench_720p(local, valu_peak_tops=peaks[1]))
        return out
    # 16 worker threads (the GPU boxes give the container 16 CPUs), 4 lanes; and the same 32 sessions without lock-step launches
    guarded("system_group", lambda: [bench_system_group(local, s_, 16, lanes=4) for s_ in group_sessions] +
            [bench_system_group(local, 32, 16, lockstep=False), bench_system_group(local, 32, 8, lanes=2)])
    guarded("system_streams", lambda: [bench_system_streams(local, c_) for c_ in (4, 8)])
    if not is720 and part == "all":
        guarded("system_720p", lambda: run_system_line(local, seed, 1280, 720, 15, steps))
    guarded("local_ba_batch", lambda: bench_ba_batch(bctx, ba_pb, peaks))
    guarded("two_view_init", lambda: bench_two_view_init(bctx))
    guarded("batched_preprocess", lambda: bench_batched_preprocess(local))
    guarded("track_mono_batch", lambda: [bench_track_mono_batch(local, c_) for c_ in (16, 64)])
    guarded("frame_step_batch", lambda: [bench_track_mono_batch(local, c_, detector=True) for c_ in (16, 64)])
    if part == "all":
        guarded("config_1280x720", lambda: bench_720p(local, valu_peak_tops=peaks[1]))

    def sld():
        a, b = stage_list_driver(local, seed, steps)
        out["stage_us"] = b
        return a
    guarded("stage_list_driver", sld)
    if with_cpu:
        import sys
        sys.path.insert(0, str(ROOT / "tests"))
        import oracles
        if oracles.ref_available():
            guarded("cpu_stages", lambda: {"stages_640x480": cpu_stage_table(W, H, SYSTEM_CELL, 2000, seed, budget_s=3.0),
                                           "stages_1280x720": cpu_stage_table(1280, 720, 15, 4000, seed, budget_s=5.0)})
    return out


def main():
    import argparse
    import alvaar_amd
    from alvaar_amd import synth
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", type=str, default="", help="comma-separated line names (e.g. system_group)")
    ap.add_argument("--sessions", type=str, default="8,16,32,64")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--stagger", type=int, default=0)
    ap.add_argument("--no-lockstep", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    bctx = alvaar_amd.Context(0)
    if args.only:
        res = {}
        for name in args.only.split(","):
            if name == "system_group":
                res[name] = [bench_system_group(0, int(s_), args.threads, lockstep=not args.no_lockstep, n_streams=args.streams, lanes=args.lanes, stagger=args.stagger)
                             for s_ in args.sessions.split(",")]
            elif name == "system_streams":
                res[name] = [bench_system_streams(0, c_) for c_ in (4, 8)]
            elif name == "system_720p":
                res[name] = run_system_line(0, 7, 1280, 720, 15, args.steps)
            elif name == "config_1280x720":
                res[name] = bench_720p(0, valu_peak_tops=bench.measured_peaks(bctx)[1])
            else:
                raise SystemExit(f"unknown line {name}")
    else:
        peaks = bench.measured_peaks(bctx)
        res = run_secondary(0, 7, args.steps, bctx, synth.make_ba_problem(20, 3000, 42), peaks, with_cpu=not args.no_cpu,
                            group_sessions=tuple(int(s_) for s_ in args.sessions.split(",")))
    p = ROOT / "gpurun_out"
    target = (p if p.is_dir() else ROOT) / "bench_detail_secondary.json"
    target.write_text(json.dumps(res, indent=1))
    log(f"wrote {target}")
    print(json.dumps({k: (v if not isinstance(v, dict) or len(json.dumps(v)) < 600 else "see file") for k, v in res.items()})[:3500])


if __name__ == "__main__":
    main()
