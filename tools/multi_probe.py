"""Aggregate frames/s of S independent streams on one GPU (native host threads)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
for s in (1, 2, 4, 8, 16, 32):
    r = bench.bench_multi_stream(0, s, 100, 10)
    print(r, flush=True)
