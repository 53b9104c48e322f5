"""A group of S sessions on W worker threads (alva_system_group, lock-step launches): run to the steady state, then `steps` group steps.
Under `rocprofv3 --kernel-trace` (tools/group_trace.sh) the kernel trace of the LAST window is folded by tools/group_trace_fold.py into:
GPU busy fraction (union of kernel intervals), mean concurrency, per-kernel count / mean duration.  Prints wall-clock frames/s and the
window's start / end in the GPU's timestamp domain is not needed: the fold takes the last `steps` tracker launches.
usage: python tools/group_trace.py [S] [W] [steps] [lockstep 0|1]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench_detail as bd  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Wt = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 72
lock = (int(sys.argv[4]) != 0) if len(sys.argv) > 4 else True
lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
r = bd.bench_system_group(0, S, Wt, steps=steps, lockstep=lock, lanes=lanes)
print(json.dumps({k: v for k, v in r.items() if k != "note"}))
