"""Host-side timeline of alva_frontend_track_ahead (ALVA_FE_TIMING=1): python tools/fe_timeline.py [steps] [lookahead 0/1]"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["ALVA_FE_TIMING"] = "1"
sys.path.insert(0, ".")
import time  # noqa: E402
import torch  # noqa: E402
import bench_detail as bench  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
la = int(sys.argv[2]) if len(sys.argv) > 2 else 1
job = bench.FrameJob(0, 7)
for _ in range(20):
    job.step_native(lookahead=bool(la))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    job.step_native(lookahead=bool(la))
torch.cuda.synchronize()
print("lookahead", la, "frames/s", steps / (time.perf_counter() - t0))
