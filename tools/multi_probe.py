"""Aggregate frames/s of S independent streams on one GPU (native host threads): python tools/multi_probe.py [S ...]"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "16"))
sys.path.insert(0, ".")
import torch  # noqa: E402,F401
import bench_detail as bench  # noqa: E402
for s in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8, 16, 32):
    r = bench.bench_multi_stream(0, s, 100, 10)
    print(os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("ALVA_FE_PRIORITIES"), r["streams"], round(r["frames_per_s"]), flush=True)
