"""Sweep of alva_track_batch_step over the number of lock-step cameras (bench.bench_track_mono_batch)."""
import json
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
import bench_detail as bench  # noqa: E402

det = os.environ.get("DETECTOR") == "1"
feat = 2000
if os.environ.get("SIZE") == "720":   # BASELINE configs[2] geometry: 1280x720, 4080 tracked keypoints, ORB 4000
    bench.W, bench.H, bench.NKP, feat = 1280, 720, 4080, 4000
for b in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 128]:
    r = bench.bench_track_mono_batch(0, b, reps=6 if b >= 64 else 20, detector=det, orb_features=feat)
    print(json.dumps({k: r[k] for k in ("cameras", "ms_per_step", "frames_per_s", "poses_accepted_frac", "single_camera_fallbacks",
                                        "kernel_us_per_step", "achieved_GBps")}), {n: v["avg_us"] for n, v in r["kernels"].items()}, flush=True)
