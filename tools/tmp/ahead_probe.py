import sys, time
sys.path.insert(0, ".")
import bench, torch
job = bench.SystemJob(0, 7, host_copy=False)
for _ in range(700): job.step()
job.ar.timing()
for name, fn in (("plain", job.step), ("hints", job.step_ahead), ("plain", job.step), ("hints", job.step_ahead)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(540): fn()
    torch.cuda.synchronize(); print(name, round(540 / (time.perf_counter() - t0)), "frames/s", {a: round(1e6 * b / 540, 1) for a, b in job.ar.timing().items()})
