"""alva_pyramid_build_from_rgba_batch on 64 cameras, a few times -- the command profiled with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
(two separate passes) for the HBM traffic of the batched image kernels."""
import sys
sys.path.insert(0, ".")
import torch
import bench_detail as bench
r = bench.bench_batched_preprocess(0, 64, reps=5)
print({k: r[k] for k in ("cameras", "ms_per_batch", "achieved_GBps", "alg_bytes_per_batch")})
