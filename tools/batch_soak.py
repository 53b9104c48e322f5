"""Soak of the batched step: N steps of 16 cameras with the detector lane; poses must stay accepted, keypoint counts stable, no
single-camera fallbacks, no device-memory growth."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import alvaar_amd  # noqa: E402
import bench_detail as bench  # noqa: E402
from alvaar_amd import synth  # noqa: E402

steps, B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 16
dev = torch.device("cuda", 0)
rings = [torch.from_numpy(synth.stream_rgba(bench.W, bench.H, bench.RING, seed=3 + s, noise=True)).to(dev) for s in range(4)]
pbs = [synth.make_pnp_problem(bench.NKP, 3 + s, outlier_frac=0.1, pose_noise=0.01) for s in range(4)]
pts = [torch.from_numpy(bench.make_keypoints(bench.NKP, 3 + s)).to(dev) for s in range(4)]
bv, uv, wp = ([torch.from_numpy(p[k]).to(dev) for p in pbs] for k in ("bv", "uv", "wpt"))
tb = alvaar_amd.TrackBatch(0, bench.W, bench.H, B, bench.NKP, bench.NKP)
tb.enable_detector(2000)
tb.bind([pts[c % 4] for c in range(B)], [bv[c % 4] for c in range(B)], [uv[c % 4] for c in range(B)], [wp[c % 4] for c in range(B)])
frames = [rings[c % 4].clone() for c in range(B)]
tables = [tb.frame_table([f[r] for f in frames]) for r in range(bench.RING)]
K = pbs[0]["K"]
for k in range(20):
    tb.step_table(tables[k % bench.RING], K)
free0 = torch.cuda.mem_get_info(0)[0]
ref_nkp = {}
bad = 0
for k in range(20, 20 + steps):
    st, _ = tb.step_table(tables[k % bench.RING], K)
    bad += int((st != 2).sum())
    key = k % bench.RING
    if key in ref_nkp:
        assert np.array_equal(ref_nkp[key], tb.nkp), (k, ref_nkp[key], tb.nkp)   # same frames -> same keypoint counts, every time
    else:
        ref_nkp[key] = tb.nkp.copy()
free1 = torch.cuda.mem_get_info(0)[0]
print("steps", steps, "cameras", B, "rejected poses", bad, "fallbacks", tb.stats()[1], "device memory delta (MB)", (free0 - free1) / 1e6,
      "keypoints", int(tb.nkp.min()), int(tb.nkp.max()))
assert bad == 0 and tb.stats()[1] == 0 and abs(free0 - free1) < 64e6
