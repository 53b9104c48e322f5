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
kt = capi.kernel_times(lambda: orb.detect_and_compute(g), 50)
tot = 0
for k, (calls, us) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    print(f"{k:24s} launches/call {calls / 50:5.2f}  avg {us:7.2f} us")
    tot += calls / 50 * us
print("sum of kernel times per call %.1f us" % tot)
