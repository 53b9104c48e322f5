"""k_p3p_batch time against the number of correspondences per camera (64 cameras, one lane)."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["ALVA_TRACK_BATCH_ONE_LANE"] = "1"
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
import bench_detail as bench  # noqa: E402
for n in (265, 530, 1060, 2120, 4240):
    bench.NKP = n
    r = bench.bench_track_mono_batch(0, 64, reps=3)
    print(n, {k: v["avg_us"] for k, v in r["kernels"].items() if "p3p" in k or "pnp" in k or "klt" in k}, flush=True)
