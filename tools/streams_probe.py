import sys, json
sys.path.insert(0, ".")
import bench_detail as bench  # noqa: E402
for n in (4, 8):
    r = bench.bench_system_streams(0, n, steps=300)
    print(n, "sessions:", round(r["frames_per_s"]), "frames/s")
r = bench.bench_system_group(0, 8, 8, steps=200)
print("group 8/8:", round(r["frames_per_s"]))
