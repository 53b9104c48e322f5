"""Sustained run of the System surface on the bench stream (frames resident in HBM): frames/s and the host-side section timers over the
last WINDOW frames.  env: CELL (12), FRAMES (1000), WINDOW (400)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import bench_detail as bench  # noqa: E402

n, win = int(os.environ.get("FRAMES", "1000")), int(os.environ.get("WINDOW", "400"))
bench.SYSTEM_CELL = int(os.environ.get("CELL", "12"))
job = bench.SystemJob(0, 7, host_copy=False)
ar = job.ar
for k in range(n - win):
    job.step()
ar.timing(); ar.timing_keyframe(); ar.timing_fine()
kf0 = int(ar.state()[11])
t0 = time.perf_counter()
for k in range(win):
    job.step()
dt = time.perf_counter() - t0
sec, kfd, nkf = ar.timing(), ar.timing_keyframe(), int(ar.state()[11]) - kf0
s = ar.state()
print(f"{win / dt:.0f} frames/s over the last {win} of {n} frames; {nkf} keyframes; keypoints {s[2]} ({s[4]} 3-D), keyframes in map {s[6]}, map points {s[7]}, status {job.status_hist}")
print("  us per frame:", {a: round(1e6 * b / win, 1) for a, b in sec.items()})
print("  us per keyframe:", {a: round(1e6 * b / max(nkf, 1), 1) for a, b in kfd.items()})
if os.environ.get("FINE"):
    fine = ar.timing_fine()
    print("  fine, per keyframe (us | counts):", {a: round((1e6 if not a.startswith("#") else 1) * b / max(nkf, 1), 1) for a, b in fine.items()})
