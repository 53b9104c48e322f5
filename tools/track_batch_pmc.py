"""Four alva_track_batch_step calls on 64 cameras: the command profiled with rocprofv3 --pmc for the SQ counters of the batch kernels."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
import bench_detail as bench  # noqa: E402
r = bench.bench_track_mono_batch(0, int(sys.argv[1]) if len(sys.argv) > 1 else 64, reps=1)
print(r["ms_per_step"])
