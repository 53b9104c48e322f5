"""The driver's window (bench.py --steps 20 --warmup 5): host-side section timers over exactly those 20 steps."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch  # noqa: F401
import bench_detail as bench  # noqa: E402
job = bench.SystemJob(0, 7, host_copy=False)
ar = job.ar
extra = 0
while job.status_hist[1] == 0 and extra < 60:
    job.step(); extra += 1
for _ in range(5):
    job.step()
ar.timing(); ar.timing_keyframe()
kf0 = int(ar.state()[11])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    job.step()
dt = time.perf_counter() - t0
sec, kfd, nkf = ar.timing(), ar.timing_keyframe(), int(ar.state()[11]) - kf0
print(f"{20 / dt:.0f} frames/s, {dt * 1e3:.2f} ms for 20 steps, untimed frames before {extra + 5}, keyframes in window {nkf}, state {list(ar.state())}")
print("  us per frame:", {a: round(1e6 * b / 20, 1) for a, b in sec.items()})
print("  us per keyframe:", {a: round(1e6 * b / max(nkf, 1), 1) for a, b in kfd.items()})
