"""compact view of the system_group lines bench_detail.py just wrote: python tools/show_group.py [tag]"""
import json
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[1]
p = root / "gpurun_out" / "bench_detail_secondary.json"
if not p.exists():
    p = root / "bench_detail_secondary.json"
tag = sys.argv[1] if len(sys.argv) > 1 else ""
keep = ("sessions", "host_threads", "lanes", "frames_per_s", "ms_tracking_step_median", "ms_keyframe_step_mean", "worker_busy_frac",
        "host_work_us_per_frame", "chain_launches_issued_per_group_step")
for e in json.loads(p.read_text()).get("system_group", []):
    print(tag, {k: (round(v, 1) if isinstance(v, float) and v > 100 else v) for k, v in e.items() if k in keep})
