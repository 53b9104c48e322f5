"""Cost of ONE LK iteration for a wave that is alone on its SIMD (the regime of the tracker launch's last slots): k_klt on n keypoints of
two independent noise images (nothing converges: every level runs to the cap), time against the iteration cap.
python tools/klt_lone_probe.py   (GPU box)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import alvaar_amd

ctx = alvaar_amd.Context(0)
w, h = 640, 480
rng = np.random.RandomState(5)
a = rng.randint(0, 256, (h, w)).astype(np.uint8)
b = rng.randint(0, 256, (h, w)).astype(np.uint8)
pa, pb = alvaar_amd.Pyramid(ctx, w, h, 9, 3), alvaar_amd.Pyramid(ctx, w, h, 9, 3)
pa.build_from_gray(torch.from_numpy(a).cuda())
pb.build_from_gray(torch.from_numpy(b).cuda())
for n in (1, 64, 1024):
    pts = torch.from_numpy(rng.uniform(100, [w - 100, h - 100], (n, 2)).astype(np.float32)).cuda()
    res = {}
    for mi in (10, 40, 100):
        for _ in range(3):
            ctx.fbklt_track(pa, pb, pts, pts.clone(), 3, max_iters=mi)
        ctx.sync()
        reps = 40
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fbklt_track(pa, pb, pts, pts.clone(), 3, max_iters=mi)
        ctx.sync()
        res[mi] = (time.perf_counter() - t0) / reps * 1e6
    # forward 4 levels (+ the backward one when the forward result passes its gate: rare on noise)
    per_it = (res[100] - res[10]) / 90.0
    print(f"n={n:5d}: cap 10 {res[10]:7.1f} us, cap 40 {res[40]:7.1f} us, cap 100 {res[100]:7.1f} us -> {per_it:.3f} us per cap step = {per_it / 4:.3f} us per iteration if 4 levels run to the cap")
