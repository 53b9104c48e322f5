#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X, one JSON line (see the driver contract).

HEADLINE (`value`): frames/s of the reference's OWN dataflow through the drop-in surface -- System::findCameraPose
(src/slam/src/system.cpp:106-175) = alva_system_find_camera_pose_device -- on a synthetic 640x480 stream with ~2000 keypoints per frame
(cell size 12 => 2120 cells; BASELINE.json configs[1]), every frame ALREADY resident in HBM, in the STEADY STATE of a long session.  A
"step" is ONE frame through the whole per-frame state machine, every stage consuming what the previous one produced:
    RGBA -> gray -> LK pyramid (+Scharr)                                                        (a2, a3)
    motion-model priors -> forward-backward KLT (1 level from the priors, the full pyramid for the rest and for the retries), one
        workgroup per slot of the frame container, with undistortion / bearings                (a4; visual_frontend.cpp:103-243)
    compaction of the 3-D survivors on the device                                              (:275-298)
    P3P-LMedS (100 hypotheses) -> drop its outliers -> robust PnP (5 LM iterations) on the tracker's OWN survivors (a8, a9; :245-417)
    host bookkeeping of the frame (keypoint updates / removals, motion model, keyframe decision)
and on the frames the reference's keyframe policy selects (every 18th here): grid Shi-Tomasi detection + ORB description (a5, a6),
triangulation, covisibility, guided Hamming matching to the local map + map-point merges (a7 / f1), local bundle adjustment with outlier
sweep, write-back and culling (a10-a13).  The stream is the (2, 1) px / frame crop of a textured canvas, 200 frames, played forwards
and backwards.  UNTIMED top-up before the window: the session runs until the 30-keyframe window of the reference's mapper is full
(keyframe 34, ~600 frames) and on to the middle of a keyframe period, so that the K timed steps hold round(K / period) keyframes --
round 2's window sat right after initialisation (2-3 cheap keyframes in the map) and overstated the sustained rate 1.5x.
`value` = frames/s over all ranks (one independent stream per GPU, no collective on the data path); "value_window", "sustained" (the
same loop for >= 0.5 s), the keyframes inside each, and "system_surface" (the same loop fed from HOST memory exactly as src/system.js
does: one copy into the wrapper's registered frame buffer, read in place over PCIe) are reported side by side.

Beside it: "bounds" = the three end-to-end bounds of a frame (HBM, PCIe, launch latency) + the measured dependent kernel
chain; "klt" = the tracker's own figures (keypoint-levels/s, L2 hit rate); "roofline" = the contract's object for the dominant kernel
(bound: latency -- see its note); "local_ba" / "roofline_ba" = the second half of BASELINE.json's metric (20 KF x 3000 pts, 5 LM
iterations); "system_720p" = configs[4]'s geometry through the same surface; "map_merge" = the optional shared-map merge on the
process group's backend (RCCL: one all_gather_into_tensor + fuse; initialised for ONE rank too); "system_streams" = 4 / 8 independent
sessions on the one GPU; "cpu_baseline" = the compiled reference (oracle/_ref) on this box's host cores: System frames/s at cell 12 /
cell 40 (shipped) / 1280x720, 1 core and 8 threads, per-stage milliseconds of its L1 functions, one Ceres local-BA solve.

Secondary lines for a RIG of lock-step cameras (alva_track_batch_*): "track_mono_batch", "frame_step_batch", "batched_preprocess";
"config_1280x720" = configs[2]; "stage_list_driver" = round 1's headline (fixed correspondences: an upper bound of stage throughput,
not the reference's dataflow).  --quick skips the secondary lines, --no-cpu-baseline the reference.  stdout carries exactly ONE line
(the JSON); everything else -- progress, the reference's and RCCL's own prints -- goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The HIP runtime multiplexes all streams of a priority class onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams
# that share a queue serialise; the multi-camera measurement drives up to 16 x 3 streams.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

W, H, NKP = 640, 480, 2120          # C640: cell 12 -> 53 x 40 = 2120 keypoints
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
RING = 8                            # synthetic frames resident in HBM


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



STREAM_FRAMES = 200                  # frames of the synthetic stream resident in HBM (245 MB)
SYSTEM_CELL = 12                     # 53 x 40 = 2120 cells => ~2000 keypoints per frame


def stream_index(k: int) -> int:
    """frame k of the endless stream: the 200-frame crop sequence forwards, then backwards, ..."""
    period = 2 * (STREAM_FRAMES - 1)
    r = k % period
    return r if r < STREAM_FRAMES else period - r


class SystemJob:
    """The drop-in surface on one GPU: alva::System at cell 12 over a synthetic stream resident in HBM (and the same stream in host
    memory for the PCIe-fed variant).  width / height / cell default to configs[1]; --config 720p-streams runs configs[4]'s geometry
    (1280x720, cell 15 => 4080 cells) instead."""

    def __init__(self, device: int, seed: int, host_copy: bool = True, width: int = W, height: int = H, cell: int | None = None):
        from alvaar_amd import synth
        from alvaar_amd.system import AlvaAR
        self.dev = torch.device("cuda", device)
        cell = SYSTEM_CELL if cell is None else cell
        canvas = synth.texture_canvas(width, height, seed)
        host = np.stack([synth.gray_to_rgba(synth.frame_gray(canvas, k, width, height, noise_seed=11)) for k in range(STREAM_FRAMES)])
        self.frames = torch.from_numpy(host).to(self.dev)
        self.host_frames = host if host_copy else None
        self.ar = AlvaAR(width, height, device=device, cell_size=cell, random_sampling=False)
        self.k = -1
        self.status_hist = [0, 0, 0, 0]
        self.ptrs = [int(self.frames[i].data_ptr()) for i in range(STREAM_FRAMES)]

    def step(self):
        self.k += 1
        st = self.ar.find_camera_pose_device(self.ptrs[stream_index(self.k)], 33.0 * self.k)
        self.status_hist[st] += 1
        return st == 1

    def step_ahead(self):
        """the same step with the NEXT frame named (alva_system_hint_next_frame_device): its gray image / pyramid are built beside this
        frame's pose solve"""
        self.k += 1
        st = self.ar.find_camera_pose_device(self.ptrs[stream_index(self.k)], 33.0 * self.k, self.ptrs[stream_index(self.k + 1)])
        self.status_hist[st] += 1
        return st == 1

    def step_host(self):
        self.k += 1
        # src/system.js:175 memImg.write(frame.data): AlvaAR.findCameraPose copies the caller's frame into its ONE registered frame buffer
        pose, st = self.ar.findCameraPose(self.host_frames[stream_index(self.k)], 33.0 * self.k)
        self.status_hist[st] += 1
        return st == 1

    def warm_to_steady_state(self, max_frames: int = 2500, then_untimed: int = 0):
        """Untimed: run until the map is in the regime a long session lives in -- the 30-keyframe window full (mapper.cpp:24-28 removes
        keyframe k - 30 from keyframe 31 on) -- then on to the middle of a keyframe period, so that a K-step window holds round(K / period)
        keyframes: the nearest whole number to their natural share.  Returns (frames run, keyframe period in frames)."""
        n0 = self.k
        kf_frames = []
        last = int(self.ar.state()[11])
        while self.k - n0 < max_frames:
            self.step()
            nk = int(self.ar.state()[11])
            if nk != last:
                kf_frames.append(self.k)
                last = nk
            if nk >= 34 and len(kf_frames) >= 8:
                break
        period = float(np.median(np.diff(kf_frames[-8:]))) if len(kf_frames) >= 3 else 0.0
        if period > 2:
            # the caller runs `then_untimed` more untimed steps (--warmup) before its window: aim so that the WINDOW starts mid-period
            target = int(period // 2 - then_untimed) % int(period)
            while (self.k - kf_frames[-1]) != target and self.k - n0 < max_frames + 64:
                self.step()
                nk = int(self.ar.state()[11])
                if nk != last:
                    kf_frames.append(self.k)
                    last = nk
        return self.k - n0, period


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


def bench_system_group(device: int, n_sessions: int, n_threads: int, steps: int = 200, n_streams: int = 0):
    """S independent alva::System sessions on ONE GPU through alva_system_group: W host threads, the sessions as fibers -- a session's
    waits for the GPU run the thread's other sessions, so the threads execute map-layer work only.  All sessions replay the same resident
    stream in lock-step (keyframes coincide: the worst case for the host).  Aggregate frames/s in the steady state."""
    from alvaar_amd.system import AlvaAR, SystemGroup
    base = SystemJob(device, 7, host_copy=False)
    group = SystemGroup([], n_threads)
    if n_streams > 0:   # sessions share n_streams HIP streams (session i -> worker i % n_threads -> stream (i % n_threads) % n_streams)
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
        ptr = base.ptrs[stream_index(k)]
        st = group.step_device([ptr] * n_sessions, 33.0 * k)
        k += 1
        return st
    while int(base.ar.state()[11]) < 34 and k < 2500:   # steady state: the 30-keyframe window full
        step()
    t0 = time.perf_counter()
    tracked = 0
    for _ in range(steps):
        tracked += int((step() == 1).sum())
    dt = time.perf_counter() - t0
    group.close()
    for s in sessions:
        s.close()
    return {"sessions": n_sessions, "host_threads": n_threads, "hip_streams": n_streams or n_sessions, "frames_per_s": n_sessions * steps / dt, "ms_per_group_step": dt / steps * 1e3,
            "tracked_frac": tracked / (n_sessions * steps),
            "note": "alva_system_group: sessions are fibers on the worker threads (a wait for the GPU switches to the thread's next session); "
                    "frames resident in HBM, every session its own map / streams / kernels"}


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
    algb = {"k_fast_nms": 3.27 * P2, "k_blur7_batch": 2 * 3.27 * P2, "k_bf_partial": 32 * 2 * step.n + 8 * step.n * ((step.n + 63) // 64)}
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


def bench_ba(ctx, reps: int = 3):
    from alvaar_amd import synth
    pb = synth.make_ba_problem(20, 3000, 42)
    ctx.local_ba(pb, 5, 0.0)  # warm (scratch allocation)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ctx.local_ba(pb, 5, 0.0)
    dt = (time.perf_counter() - t0) / reps
    nobs = len(pb["obs_kf"])
    iters = int(r["info"][0]) - 1
    return dict(residual_blocks=nobs, lm_iterations=iters, ms_per_solve=dt * 1e3,
                residual_block_iters_per_s=nobs * iters / dt, final_cost=float(r["info"][2]),
                note="whole alva_local_ba call incl. host structure build, H2D of the problem and D2H of results; the synthetic problem of "
                     "SURVEY.md 8(d) converges by function tolerance 0 after 4 accepted steps (5 iterations allowed)"), pb


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


def measured_peaks(ctx):
    """FP64 MFMA TFLOP/s and integer VALU Tops/s of this GPU (alva_microbench_peaks): the guide lists neither"""
    import ctypes as C
    from alvaar_amd.capi import lib, check
    lib.alva_microbench_peaks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    a, b = C.c_double(0), C.c_double(0)
    best = [0.0, 0.0]
    for _ in range(3):
        check(lib.alva_microbench_peaks(ctx.h, C.byref(a), C.byref(b)))
        best = [max(best[0], a.value), max(best[1], b.value)]
    return best


def roofline_ba(ctx, pb, peaks):
    """Roofline of the local-BA kernels on the 20 KF x 3000 pts problem: the per-point Jacobian kernel and the pair reduction against
    HBM, the Schur-complement GEMM against the MEASURED FP64-MFMA ceiling (event-timed kernels over whole alva_local_ba calls)."""
    from alvaar_amd import capi
    reps = 3
    kt = capi.kernel_times(lambda: ctx.local_ba(pb, 5, 0.0), reps)
    nobs, npt = len(pb["obs_kf"]), len(pb["anchor_kf"])
    nfree = int((np.asarray(pb["kf_const"]) == 0).sum())
    m = ((6 * nfree + 1 + 15) // 16) * 16                      # reduced camera system + rhs column, padded to 16 x 16 MFMA tiles
    out = {"problem": {"residual_blocks": nobs, "points": npt, "free_keyframes": nfree}, "kernels": {}}
    alg = {
        # per residual block and Jacobian evaluation: 60 B in (2 observations + indices), 112 B stored (J_obs 2x6 + residual)
        "k_point<true, true>": ("hbm", (60 + 112) * nobs),
        "k_pairs": ("hbm", 112 * nobs + 27 * 8 * 400),
        "k_gemm": ("mfma", 2.0 * m * m * npt),                    # G = Z'Z: [m x points] x [points x m], FP64
    }
    for k, (bound, work) in alg.items():
        hit = [n for n in kt if n.startswith(k.split("<")[0])]
        if not hit:
            continue
        calls, us = kt[hit[0]]
        if bound == "hbm":
            a = work / (us * 1e-6) / 1e9
            out["kernels"][hit[0]] = {"bound": "hbm", "avg_us": round(us, 2), "launches_per_solve": calls / reps, "alg_bytes": int(work),
                                      "achieved_GBps": round(a, 1), "peak_GBps": HBM_PEAK_GBS, "frac": a / HBM_PEAK_GBS}
        else:
            a = work / (us * 1e-6) / 1e12
            out["kernels"][hit[0]] = {"bound": "mfma_f64", "avg_us": round(us, 2), "launches_per_solve": calls / reps, "flops": int(work),
                                      "achieved_TFLOPs": round(a, 2), "peak_TFLOPs_measured": round(peaks[0], 1),
                                      "peak_TFLOPs_spec": 78.6, "frac_of_measured": a / peaks[0]}
    tot = sum(c / reps * u for c, u in kt.values())
    out["kernel_us_per_solve"] = round(tot, 1)
    out["largest"] = {n: round(c / reps * u, 1) for n, (c, u) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]}
    out["note"] = ("one 20 KF x 3000 pts problem cannot fill the chip: the reduced camera system is 109 x 109 and k_solve is a single-workgroup "
                   "dependent chain; MFMA utilisation is reported against the measured v_mfma_f64_16x16x4_f64 ceiling (alva_microbench_peaks)")
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


def cpu_baseline(seed: int, frames_1: int = 150, frames_8: int = 60):
    """The reference itself on the host cores of this box (SURVEY.md 8(d) "CPU baseline timing" (1)-(3)), same stream, explicit
    timestamps, fixed-seed sampling, Ceres' wall-clock caps frozen (they would silently skip work):
      (1) System::findCameraPose frames/s on ONE core (the reference is single-threaded: wasm, NO_THREADS Ceres) at cell 12 (the metric's
          ~2000 keypoints; this is `value`), at the SHIPPED cell 40 (system.cpp:15), at 1280x720 / cell 15 (configs[4]); and 8 independent
          reference Systems on 8 host threads (streams are independent: that is how the reference would use 8 cores);
      (2) per-stage milliseconds, cpu_stage_table();  (3) one local-BA solve through Ceres with the reference's cost functions.
    Bounded sample: the first frames of the stream (incl. initialisation and the first keyframes); ~15 s of CPU work in total."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracles
    from alvaar_amd import synth
    if not oracles.ref_available():
        return cpu_baseline_port(seed)
    import sysdiff
    import threading
    canvas = synth.texture_canvas(W, H, seed)
    frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, W, H, noise_seed=11)) for k in range(max(frames_1, frames_8, 200))]

    def run(n, out, slot, w=W, h=H, cell=SYSTEM_CELL, fr=frames):
        ref = sysdiff.RefSystem(w, h, cell)
        t0 = time.perf_counter()
        st = [ref.step(fr[k], 33.0 * k)[0] for k in range(n)]
        out[slot] = (time.perf_counter() - t0, st, int(ref.state()[2]), len(ref.keyframe_ids()))
        ref.close()
    one = [None]
    run(frames_1, one, 0)
    dt1, st1, nkp, nkf = one[0]
    res = [None] * 8
    th = [threading.Thread(target=run, args=(frames_8, res, i)) for i in range(8)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt8 = time.perf_counter() - t0
    c40 = [None]
    run(200, c40, 0, cell=40)
    canvas720 = synth.texture_canvas(1280, 720, seed)
    frames720 = [synth.gray_to_rgba(synth.frame_gray(canvas720, k, 1280, 720, noise_seed=11)) for k in range(40)]
    c720 = [None]
    run(40, c720, 0, w=1280, h=720, cell=15, fr=frames720)
    pbba = synth.make_ba_problem(20, 3000, 42)
    t1 = time.perf_counter()
    r = oracles.Ref.local_ba(pbba, 5, 0.0)
    dtb = time.perf_counter() - t1
    desc = lambda tup, n: f"{tup[1].count(3)} initialising, {tup[1].count(1)} tracked, {tup[3]} keyframes, {tup[2]} keypoints at the end, {n} frames"
    return {"value": frames_1 / dt1, "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's System::findCameraPose (oracle/_ref) on the first {frames_1} frames of the same stream, cell {SYSTEM_CELL}: "
                      f"{st1.count(3)} initialising, {st1.count(1)} tracked, {nkf} keyframes with local BA, {nkp} keypoints at the end; + 1 local-BA solve (20 KF x 3000 pts)",
            "eight_threads": {"value": 8 * frames_8 / dt8, "unit": "frames/s", "cores": 8,
                              "sample": f"8 independent reference Systems on 8 host threads, {frames_8} frames each (the reference is single-threaded; independent streams are its only parallelism)"},
            "system_cell40_shipped": {"value": 200 / c40[0][0], "unit": "frames/s", "cores": 1, "sample": "640x480, cell 40 (system.cpp:15): " + desc(c40[0], 200)},
            "system_1280x720_cell15": {"value": 40 / c720[0][0], "unit": "frames/s", "cores": 1, "sample": "configs[4] geometry: " + desc(c720[0], 40)},
            "stages_640x480": cpu_stage_table(W, H, SYSTEM_CELL, 2000, seed, budget_s=3.0),
            "stages_1280x720": cpu_stage_table(1280, 720, 15, 4000, seed, budget_s=5.0),
            "local_ba_residual_block_iters_per_s": len(pbba["obs_kf"]) * (int(r["info"][0]) - 1) / dtb,
            "local_ba_ms": dtb * 1e3}


def cpu_baseline_port(seed: int, budget_s: float = 12.0):
    """Fallback when the compiled reference is absent: the stage-wise C restatement on a bounded sample of the stage list."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracles
    from scipy.spatial.transform import Rotation
    from alvaar_amd import synth
    O = oracles.Orc
    frames = synth.stream_rgba(W, H, 4, seed=seed, noise=True)
    pts = make_keypoints(NKP, seed)
    pb = synth.make_pnp_problem(NKP, seed, outlier_frac=0.1, pose_noise=0.01)
    prev = O.rgba2gray(frames[0])
    n, t0 = 0, time.perf_counter()
    while True:
        k = 1 + n % 3
        g = O.rgba2gray(frames[k])
        O.fbklt(prev, g, pts, pts, 3)
        ok, R, t, outl = O.p3p_lmeds(pb["bv"], pb["wpt"], fx=pb["K"][0], fy=pb["K"][1])
        keep = np.setdiff1d(np.arange(NKP), outl)
        q = Rotation.from_matrix(R).as_quat()
        O.pnp_refine(pb["uv"][keep], pb["wpt"][keep], np.concatenate([t, q]), pb["K"])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 40:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} tracking frames (gray, 2 LK pyramids, fb-KLT 3 levels, P3P-LMedS, PnP) through the C restatement; no keyframes"}


_T0 = time.perf_counter()


def log(msg: str):
    """progress on stderr (stdout carries the one JSON line)"""
    print(f"[bench {time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def launch_latency(ctx):
    import ctypes as C
    from alvaar_amd.capi import lib, check
    lib.alva_microbench_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    a, b = C.c_double(0), C.c_double(0)
    best = [1e9, 1e9]
    for _ in range(3):
        check(lib.alva_microbench_launch(ctx.h, 200, C.byref(a), C.byref(b)))
        best = [min(best[0], a.value), min(best[1], b.value)]
    return best


def pmc_reference(kernel: str, sources: list[str]):
    """HBM traffic and L2 hit rate of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc cannot run inside this
    process).  The file is stamped with the commit and the sha256 of the kernel's source files at capture time: if the sources have
    changed since, the numbers are reported as stale (null) instead of silently carried over."""
    import hashlib
    f = ROOT / "profiles" / "r3_pmc_track_klt.json"
    if not f.exists():
        return None, None, {"file": None, "note": "no PMC capture committed for this kernel"}
    j = json.loads(f.read_text())
    now = {src: hashlib.sha256((ROOT / src).read_bytes()).hexdigest()[:16] for src in sources}
    stale = any(j.get("source_sha16", {}).get(src) != h for src, h in now.items())
    k = j.get("kernels", {}).get(kernel, {})
    stamp = {"file": str(f.relative_to(ROOT)), "captured_at_commit": j.get("commit"), "source_sha16_at_capture": j.get("source_sha16"), "stale": stale}
    if stale:
        return None, None, stamp
    return k.get("hbm_bytes_per_launch"), k.get("l2_hit_rate"), stamp


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["640x480", "720p-streams"], default="640x480",
                    help="640x480 = BASELINE configs[1] (the metric's configuration, default); 720p-streams = configs[4]: one independent "
                         "1280x720 stream (cell 15 => 4080 cells) per GPU, what an 8-GPU run of that config executes on every rank")
    ap.add_argument("--multi-stream", action="store_true",
                    help="also run the superseded 4- and 16-host-thread measurement (independent alva_frontend objects); off by default: its "
                         "concurrent launches of the same kernels would inflate their averages in a rocprofv3 profile of this command")
    ap.add_argument("--no-multi-stream", action="store_true", help="accepted for compatibility (the default now)")
    ap.add_argument("--quick", action="store_true", help="headline, roofline and CPU baseline only (skips the secondary rig / batch lines)")
    ap.add_argument("--system-streams", type=str, default="",
                    help="comma-separated session counts: time S independent alva::System sessions on rank 0's GPU (reported under "
                         "system_streams; not part of value); default 4,8 in a full run")
    ap.add_argument("--streams-per-gpu", type=int, default=0,
                    help="also time S concurrent independent streams on rank 0's GPU (reported under multi_stream; not part of value)")
    ap.add_argument("--merge-every", type=int, default=4, help="shared-map merge (RCCL all_gather + fuse) every this many keyframes in the merge line")
    args = ap.parse_args()
    # stdout carries exactly one line, the JSON: the compiled reference (cpu_baseline) and RCCL print to the C-level stdout, so fd 1 is
    # pointed at stderr for the duration of the run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from alvaar_amd import multi
    import torch.distributed as td
    shard = multi.shard_from_env()
    rank, world, local = shard.rank, shard.world, shard.local_rank
    torch.cuda.set_device(local)
    # "nccl" IS RCCL on ROCm.  The data path has no collective (independent streams); the group carries the barrier + timing reduction
    # and the optional shared-map merge -- initialised for ONE rank too, so that the merge line below runs on RCCL in every run.
    try:
        dist = multi.init_process_group(shard, "nccl", force=True)
        dist_err = None
    except Exception as e:   # a single-GPU box without a usable RCCL still measures the data path
        if world > 1:
            raise
        dist, dist_err = False, repr(e)

    is720 = args.config == "720p-streams"
    Wc, Hc, cellc = (1280, 720, 15) if is720 else (W, H, SYSTEM_CELL)
    sysjob = SystemJob(local, seed=shard.stream_seed, width=Wc, height=Hc, cell=cellc)
    job = None if is720 else FrameJob(local, seed=shard.stream_seed)

    def timed(fn, warmup, steps):
        """W untimed + exactly K timed steps, barrier + device sync on both sides, MAX over ranks"""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            el = multi.max_over_ranks(el, torch.device("cuda", local))
            td.barrier()
        torch.cuda.synchronize()
        return el

    # ---- headline: the System surface, frames resident in HBM, STEADY STATE.  The untimed top-up runs the session until the 30-keyframe
    # window is full (keyframe 34; ~600 frames) and on to the middle of a keyframe period, so the K timed steps hold round(K / period)
    # keyframes -- a keyframe costs several tracking frames, and a window right after initialisation (2-3 keyframes in the map, cheap
    # keyframes) overstated the rate a session sustains by 1.5x (round 2's verdict).  --warmup W steps run on top, as the contract says.
    log("warming the session into steady state")
    extra, period = sysjob.warm_to_steady_state(then_untimed=args.warmup)
    log(f"steady state after {extra} frames, keyframe period {period}")
    kf_before = int(sysjob.ar.state()[11])
    dt = timed(sysjob.step, args.warmup, args.steps)
    kf_in_window = int(sysjob.ar.state()[11]) - kf_before
    hist_timed = list(sysjob.status_hist)
    log(f"headline window: {args.steps / dt:.0f} frames/s per rank")
    # the same loop for at least 0.5 s
    long_steps = max(args.steps, int(0.6 * args.steps / max(dt, 1e-9)) + 1)
    kf_before = int(sysjob.ar.state()[11])
    dt_long = timed(sysjob.step, 0, long_steps)
    kf_long = int(sysjob.ar.state()[11]) - kf_before
    ar = sysjob.ar
    ar.timing()
    ar.timing_keyframe()
    ar.klt_work()
    kf0 = int(ar.state()[11])
    n_sec = 400
    for _ in range(n_sec):
        sysjob.step()
    sections, kf_detail, n_kf_sec = ar.timing(), ar.timing_keyframe(), int(ar.state()[11]) - kf0
    sys_state = ar.state()
    sys_counters = ar.counters()
    log(f"sustained: {long_steps / dt_long:.0f} frames/s per rank")
    # the same loop with look-ahead hints (the caller names the next frame: its images are built behind this frame's pose solve)
    kf_before = int(ar.state()[11])
    dt_ahead = timed(sysjob.step_ahead, 5, long_steps)
    kf_ahead = int(ar.state()[11]) - kf_before
    log(f"with next-frame hints: {long_steps / dt_ahead:.0f} frames/s per rank")
    # PCIe-fed variant: host RGBA in through AlvaAR.findCameraPose (memImg.write + the registered buffer read in place), same length
    dt_host = timed(sysjob.step_host, 5, long_steps)
    ar.timing()
    for _ in range(100):
        sysjob.step_host()
    sections_host = ar.timing()
    t0 = time.perf_counter()
    for i in range(50):
        np.copyto(ar.mem_img, sysjob.host_frames[i])
    copy_us = (time.perf_counter() - t0) / 50 * 1e6
    log(f"host-fed: {long_steps / dt_host:.0f} frames/s per rank")
    # ---- the optional shared-map merge on the process group's backend (RCCL): pack -> ONE all_gather_into_tensor -> fuse on the GPU
    merge = None
    try:
        import alvaar_amd
        mctx = alvaar_amd.Context(local)
        rounds = []
        for _ in range(3):
            for _ in range(int(max(period, 8) * args.merge_every)):
                sysjob.step()
            if dist:
                td.barrier()
            rounds.append(multi.map_merge_round(ar, mctx, shard))
        last = rounds[-1]
        merge = {"every_keyframes": args.merge_every, "backend": last["backend"], "world": world, "records_this_rank": last["records_this_rank"],
                 "records_gathered": last["records_gathered"], "fused": last["fused"], "bytes_gathered_per_rank": last["bytes_gathered"],
                 "pack_us": round(min(r["pack_us"] for r in rounds), 1), "all_gather_us": round(min(r["all_gather_us"] for r in rounds), 1),
                 "fuse_us": round(min(r["fuse_us"] for r in rounds), 1),
                 "note": "north_star's optional shared-map merge: this rank's 3-D map points (id, xyz, descriptor medoid; 64 B records) -> one "
                         "all_gather_into_tensor on the process group (RCCL over xGMI when N > 1; one rank fuses nothing: the rule only fuses "
                         "across streams) -> alva_fuse_map_points; NOT part of `value` (the data path has no collective); pack_us is host-side "
                         "(debug export of the map + numpy packing)"}
    except Exception as e:
        merge = {"error": repr(e), "process_group_error": dist_err}
    log(f"map merge: {merge}")
    dt_drv = dt_nola = dt_serial = None
    if job is not None:
        # ---- round 1's headline as a secondary line (fixed correspondences, three HIP streams)
        dt_drv = timed(job.step_native, min(args.warmup, 10), args.steps)
        dt_nola = timed(lambda: job.step_native(lookahead=False), 3, args.steps)
        dt_serial = timed(job.step, 3, args.steps)
    if rank == 0:
        import alvaar_amd
        from alvaar_amd import capi
        fps = world * args.steps / dt
        bctx = job.ctx if job is not None else alvaar_amd.Context(local)
        stage_us = job.stage_times() if job is not None else None
        log("stage list driver done; local BA")
        ba, ba_pb = bench_ba(bctx)
        peaks = measured_peaks(bctx)
        lat_dep, lat_rt = launch_latency(bctx)
        P = Wc * Hc
        # ---- roofline: per-KERNEL durations from HIP events recorded on each launch stream, over a further pass of the headline loop
        # (the events cost a few us per launch, so they stay out of the pass that gives `value`)
        PROF_STEPS = 200
        log("kernel times of the headline loop")
        ar.klt_work()
        kt = capi.kernel_times(sysjob.step, PROF_STEPS)
        klt_levels, klt_slots = ar.klt_work()
        nkp = int(sys_state[2])
        n3d = int(sys_state[4])
        # ALGORITHMIC bytes per launch (SURVEY.md §8d per-unit figures x the units one launch processes)
        alg = {
            "k_level0<true>": 4 * P + 2 * P,                        # RGBA in; gray copy + padded level 0 out
            "k_pyr_stage": (P + 4 * P + P / 4) * (1 + 1 / 4 + 1 / 16 + 1 / 64) / 4,   # per launch (4 launches): level in, Scharr out, next level out
            # fb-KLT, one launch per frame over every slot: the four levels of BOTH pyramids once (gray u8 + Ix,Iy i16 = 5 B/px per level)
            # + the slot table in (33 B per slot) + per-slot results out (1 + 8 + 8 + 24 B, device memory)
            "k_track_klt": 2 * 5 * P * (1 + 1 / 4 + 1 / 16 + 1 / 64) + 33 * nkp + 41 * nkp,
            "k_track_stage_in": 2 * 33 * nkp,                       # slot table: pinned host -> device
            "k_track_compact": 42 * nkp + 41 * nkp + 64 * n3d,      # per-slot results in; the same to pinned host + correspondences of the pose solve out
        }
        per_frame = {k: v[0] / PROF_STEPS * v[1] for k, v in kt.items()}
        kernels = {k: {"launches_per_frame": round(v[0] / PROF_STEPS, 3), "avg_us": round(v[1], 2),
                       **({"alg_bytes": int(alg[k]), "GBps": round(alg[k] / (v[1] * 1e-6) / 1e9, 1)} if k in alg else {})}
                   for k, v in sorted(kt.items(), key=lambda kv: -per_frame[kv[0]])[:28]}
        dom = max(per_frame, key=per_frame.get)
        hbm_dom = max((k for k in per_frame if k in alg), key=per_frame.get)
        achieved = alg[hbm_dom] / (kt[hbm_dom][1] * 1e-6) / 1e9
        traffic, l2_hit, pmc_stamp = pmc_reference(hbm_dom, ["alvaar_amd/csrc/klt.hip", "alvaar_amd/csrc/stages_hip.hip", "alvaar_amd/csrc/track_slots.hpp"])
        klt_us_total = kt["k_track_klt"][0] * kt["k_track_klt"][1] if "k_track_klt" in kt else None
        # ---- SURVEY.md 8(d), last table row: the three end-to-end bounds of one frame next to the achieved number
        chain = ["k_level0<true>", "k_pyr_stage", "k_track_stage_in", "k_track_klt", "k_track_compact", "k_p3p", "k_pnp"]
        chain_launches = sum(round(kt[k][0] / PROF_STEPS) for k in chain if k in kt)
        chain_kernel_us = sum(per_frame.get(k, 0.0) for k in chain)
        launches_per_frame = sum(v[0] for v in kt.values()) / PROF_STEPS
        bounds = {
            "hbm_frames_per_s": HBM_PEAK_GBS * 1e9 / (18.3 * P),
            "pcie_gen5_host_fed_frames_per_s": 63e9 / (4 * P),
            "launch_latency_frames_per_s": 1e6 / (chain_launches * lat_dep + 2 * lat_rt),
            "dependent_kernel_chain_frames_per_s": 1e6 / max(chain_kernel_us, 1e-9),
            "achieved_frames_per_s": fps / world, "achieved_sustained_frames_per_s": long_steps / dt_long,
            "inputs": {"irreducible_hbm_bytes_per_frame": int(18.3 * P), "rgba_bytes_per_frame": 4 * P, "pcie_GBps": 63.0,
                       "us_per_dependent_empty_launch": round(lat_dep, 2), "us_launch_plus_sync_round_trip": round(lat_rt, 2),
                       "launches_on_the_tracking_chain": chain_launches, "host_waits_per_tracking_frame": 2,
                       "launches_per_frame_all": round(launches_per_frame, 2), "tracking_chain_kernel_us": round(chain_kernel_us, 1)},
            "note": "SURVEY.md 8(d): HBM roof = 8 TB/s over the irreducible 18.3 P bytes of a frame; PCIe Gen5 x16 upload of the RGBA frame for a "
                    "host-fed stream; launch latency = the tracking frame's dependent launches (gray, 4 pyramid stages, stage-in, fb-KLT, "
                    "compaction, P3P, PnP) at the measured empty-launch rate + its two host waits at the measured launch+sync round trip "
                    "(alva_microbench_launch); dependent_kernel_chain = the same chain's measured kernel durations with zero gaps -- the bound "
                    "this single stream actually runs against"}
        us = lambda d, n: {a: round(1e6 * b / max(n, 1), 1) for a, b in d.items()}
        full = not args.quick and world == 1
        log("assembling the line")
        out = {
            "metric": "frames/sec @640x480 2000kp; local-BA residuals/sec (20KFx3k pts)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i16 image stages, f32 KLT, f64 pose+BA", "data": "synthetic",
            "config": {"workload": ("configs[4]: 1280x720 RGBA stream per GPU, ~4000 keypoints per frame (cell 15), " if is720 else
                                    "configs[1]: 640x480 RGBA stream, ~2000 keypoints per frame (cell 12), ") + "the reference's System::findCameraPose dataflow "
                                   "(two-pass fb-KLT from motion-model priors -> P3P-LMedS -> PnP on the tracker's survivors; keyframes: grid detector + ORB "
                                   "description, triangulation, guided Hamming matching to the local map, local BA) through alva_system_find_camera_pose_device, "
                                   "STEADY STATE (30-keyframe window full)",
                       "frames_resident_in_hbm": True, "stream": f"{STREAM_FRAMES} frames, (2, 1) px per frame, forwards / backwards",
                       "keypoints_per_frame": nkp, "keypoints_3d": n3d, "keyframes_in_map": int(sys_state[6]), "map_points": int(sys_state[7]),
                       "keyframes_created_so_far": int(sys_state[11]),
                       "status_histogram_reset_init_tracked": {"1_tracked": hist_timed[1], "2_reset": hist_timed[2], "3_initialising": hist_timed[3]},
                       "untimed_frames_to_steady_state": extra, "keyframe_period_frames": period,
                       "keyframes_in_timed_window": kf_in_window, "natural_keyframes_per_window": (args.steps / period) if period else None,
                       "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
                       "parallelism": f"{world} independent camera streams, one per GPU, no collective on the data path"},
            "value_window": {"frames_per_s": fps, "steps": args.steps, "seconds": dt, "keyframes": kf_in_window},
            "sustained": {"frames_per_s": world * long_steps / dt_long, "steps": long_steps, "seconds": dt_long, "keyframes": kf_long,
                          "value_over_sustained": fps / (world * long_steps / dt_long),
                          "note": "the same timed loop continued for at least 0.5 s"},
            "system_lookahead": {"frames_per_s": world * long_steps / dt_ahead, "ms_per_step": dt_ahead / long_steps * 1e3, "steps": long_steps,
                                 "keyframes": kf_ahead,
                                 "note": "the sustained loop with alva_system_hint_next_frame_device before every call (the caller names the frame "
                                         "of its next call; gray + LK pyramid of that frame are enqueued behind this frame's pose kernels and run "
                                         "while the host does its bookkeeping; results identical, tests/test_gpu_system.py).  NOT `value`: the reference's "
                                         "findCameraPose is handed one frame per call"},
            "system_surface": {"frames_per_s": world * long_steps / dt_host, "ms_per_step": dt_host / long_steps * 1e3, "steps": long_steps,
                               "caller_copy_us": round(copy_us, 1),
                               "per_frame_us_host_fed": us(sections_host, 100),
                               "note": "the same loop fed from HOST memory exactly as src/system.js does: memImg.write(frame) = one 1.2 MB copy into the wrapper's "
                                       "ONE frame buffer (caller_copy_us, numpy), which is registered (alva_system_register_frame_buffer) and read in place over "
                                       "PCIe by the gray / pyramid kernel -- no staging copy, no copy command; 'upload+pyramid' in per_frame_us_host_fed is the "
                                       "enqueue, the PCIe read itself overlaps the slot gathering"},
            "bounds": bounds,
            "frame_sections_us": {"per_frame": us(sections, n_sec), "frames": n_sec, "keyframes": n_kf_sec,
                                  "per_keyframe_detail": us(kf_detail, n_kf_sec),
                                  "ms_per_keyframe": round(1e3 * (sections["keyframe_create"] + sections["mapping"]) / max(n_kf_sec, 1), 3),
                                  "local_ba_solves_total": sys_counters["ba_solves"],
                                  "note": "host wall-clock per section of the frame loop (alva_system_debug_timing); keyframe sections averaged over ALL frames in per_frame"},
            "klt": {"keypoint_levels_per_s": (klt_levels / (klt_us_total * 1e-6)) if klt_us_total else None,
                    "keypoint_levels_per_frame": klt_levels / PROF_STEPS, "slots_per_frame": klt_slots / PROF_STEPS,
                    "kernel_us": kt.get("k_track_klt", (0, None))[1], "l2_hit_rate": l2_hit, "pmc": pmc_stamp,
                    "note": "SURVEY.md 8(d) fb-KLT row: the tracker is an L2-gather / dependent-iteration-latency kernel, so its figures are "
                            "keypoint-levels per second of kernel time (LK passes over one pyramid level: forward levels + the backward pass, "
                            "counted from the per-slot result codes) and the L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS), PMC pass under profiles/)"},
            "map_merge": merge,
            "local_ba": ba,
            "roofline_ba": roofline_ba(bctx, ba_pb, peaks),
            "measured_peaks": {"mfma_f64_TFLOPs": peaks[0], "valu_int32_Tops": peaks[1],
                               "note": "alva_microbench_peaks: independent v_mfma_f64_16x16x4_f64 chains / xor-popcount-add chains on every SIMD"},
            "roofline": {"bound": "latency", "roofline_axis": "hbm", "kernel": hbm_dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stamp": pmc_stamp,
                         "avg_us": kt[hbm_dom][1], "alg_bytes_per_launch": int(alg[hbm_dom]),
                         "largest_kernel_by_time": dom,
                         "note": "the contract's roofline object for the dominant kernel: algorithmic bytes / HIP-event kernel time against the 8 TB/s HBM peak. "
                                 "It is NOT what limits this kernel: traffic == algorithmic bytes (nothing re-read) and the frame is 1.2 MB; the "
                                 "kernel runs as long as its slowest keypoint's dependent LK iterations (bound: latency) -- its own figures are under 'klt', "
                                 "the frame's bounds under 'bounds'; rocprofv3 summary in profiles/"},
            "kernels": kernels,
        }
        if job is not None:
            out["stage_list_driver"] = {"frames_per_s": world * args.steps / dt_drv, "ms_per_step": dt_drv / args.steps * 1e3,
                                        "no_lookahead_frames_per_s": world * args.steps / dt_nola, "one_hip_stream_frames_per_s": world * args.steps / dt_serial,
                                        "note": "round 1's headline: the configs[1] stage list (gray, pyramid, fb-KLT 3 levels, cv::ORB detectAndCompute 2000, BF Hamming, "
                                                "P3P -> PnP) through alva_frontend_track_ahead on three HIP streams with FIXED pose correspondences; an upper bound of "
                                                "stage throughput, not the reference's dataflow"}
            out["stage_us"] = stage_us
        if full:
            log("secondary lines")
            if not is720:
                out["system_720p"] = run_system_line(local, shard.stream_seed, 1280, 720, 15, args.steps)   # configs[4]'s geometry, default-on
            out["local_ba_batch"] = bench_ba_batch(bctx, ba_pb, peaks)
            out["two_view_init"] = bench_two_view_init(bctx)
            out["batched_preprocess"] = bench_batched_preprocess(local)
            out["track_mono_batch"] = [bench_track_mono_batch(local, c_) for c_ in (16, 64)]
            out["frame_step_batch"] = [bench_track_mono_batch(local, c_, detector=True) for c_ in (16, 64)]
            out["config_1280x720"] = bench_720p(local, valu_peak_tops=peaks[1])
        if args.streams_per_gpu > 1:
            out["multi_stream"] = [bench_multi_stream(local, args.streams_per_gpu, max(20, args.steps // 2))]
        elif world == 1 and args.multi_stream:
            # secondary: several independent cameras on the one GPU (native host threads); shows the head-room a single stream leaves
            out["multi_stream"] = [bench_multi_stream(local, s_, 60) for s_ in (4, 16)]
        if world == 1 and (args.system_streams or full):
            # S independent alva::System sessions on the ONE GPU, a host thread each: what a single latency-bound stream leaves idle
            counts = [int(v) for v in args.system_streams.split(",")] if args.system_streams else [4, 8]
            out["system_streams"] = [bench_system_streams(local, c_) for c_ in counts]
            out["system_group"] = [bench_system_group(local, s_, 8) for s_ in (8, 16, 32)]
        if not args.no_cpu_baseline and world == 1:   # the contract: reference CPU path timed on rank 0 at N = 1 only
            log("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(shard.stream_seed)
            log("done")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
