import sys, json
sys.path.insert(0, ".")
import bench
for s_, w_, q_ in ((8, 8, 4), (16, 8, 4), (16, 8, 2), (32, 8, 4), (16, 4, 4), (8, 4, 4), (16, 8, 8)):
    r = bench.bench_system_group(0, s_, w_, steps=150, n_streams=q_)
    print(s_, w_, q_, round(r["frames_per_s"]), round(r["ms_per_group_step"], 2), r["tracked_frac"], flush=True)
