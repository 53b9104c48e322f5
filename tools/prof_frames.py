"""The bench's frame loop only (for rocprofv3): python tools/prof_frames.py [steps] [native|python|serial]."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench_detail as bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "native"
job = bench.FrameJob(0, 7)
fn = {"native": job.step_native, "python": job.step_overlapped, "serial": job.step}[mode]
for _ in range(steps):
    fn()
torch.cuda.synchronize()
