"""Per-kernel event timings of cv::ORB::detectAndCompute on the GPU: python tools/orb_kernels.py [width height nfeatures]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import alvaar_amd
from alvaar_amd import capi, synth

w, h, nf = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 2000)
ctx = alvaar_amd.Context(0)
g = torch.from_numpy(synth.frame_gray(synth.texture_canvas(w, h, 7), 1, w, h, noise_seed=7)).cuda()
orb = alvaar_amd.Orb(ctx, w, h, nf)
for _ in range(5):
    orb.detect_and_compute(g)
import time
kp_buf = torch.zeros((4 * nf + 1024, 6), dtype=torch.float32, device="cuda")
desc_buf = torch.zeros((4 * nf + 1024, 32), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    orb.enqueue(g, kp_buf, desc_buf)
    orb.collect()
print("wall per detectAndCompute (enqueue + collect, caller's buffers): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
kt = capi.kernel_times(lambda: orb.detect_and_compute(g), 50)
tot = 0
for k, (calls, us) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    print(f"{k:24s} launches/call {calls / 50:5.2f}  avg {us:7.2f} us")
    tot += calls / 50 * us
print("sum of kernel times per call %.1f us" % tot)
