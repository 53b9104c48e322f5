"""S independent alva::System sessions on one GPU (bench.bench_system_streams) for S in argv (default 1 2 4 8 16)."""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
import bench_detail as bench  # noqa: E402
for s in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    r = bench.bench_system_streams(0, s, 300)
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k != "note"}), flush=True)
