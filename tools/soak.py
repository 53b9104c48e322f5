"""Soak: 30 000 frames through the native driver (look-ahead on), then 600 frames through the System surface from a cold start with
a reset in the middle; prints the rate of each third of the run and the process RSS, to catch leaks or slow drifts."""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import psutil  # noqa: E402
import torch  # noqa: E402
import bench_detail as bench  # noqa: E402
from alvaar_amd import synth  # noqa: E402
from alvaar_amd.system import AlvaAR  # noqa: E402

proc = psutil.Process()
job = bench.FrameJob(0, 7)
for part in range(3):
    t0 = time.perf_counter()
    ok = 0
    for _ in range(10000):
        ok += job.step_native()
    torch.cuda.synchronize()
    print(f"driver part {part}: {10000 / (time.perf_counter() - t0):.0f} frames/s, poses accepted {ok}, RSS {proc.memory_info().rss >> 20} MiB, "
          f"GPU mem {torch.cuda.mem_get_info()[0] >> 20} MiB free", flush=True)
w, h = 640, 480
ar = AlvaAR.Initialize(w, h)
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(100)]
st = []
t0 = time.perf_counter()
for rep in range(6):
    if rep == 3:
        ar.reset()
    for k in range(100):
        pose, status = ar.findCameraPose(frames[k if rep % 2 == 0 else 99 - k])
        st.append(status)
print(f"system: {600 / (time.perf_counter() - t0):.0f} frames/s, status histogram {np.bincount(st, minlength=4).tolist()}, RSS {proc.memory_info().rss >> 20} MiB")
