"""A few alva_track_batch_step_detect calls on 64 cameras (detector lane on, one HIP stream): the command profiled with rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE (two separate passes) for the HBM traffic of the batched step."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["ALVA_TRACK_BATCH_ONE_LANE"] = "1"
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
import bench_detail as bench  # noqa: E402
r = bench.bench_track_mono_batch(0, 64, reps=1, detector=True)
print(r["ms_per_step"], r["alg_bytes_per_step"])
