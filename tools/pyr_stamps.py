"""Phase stamps of the fused ORB pyramid launch (ALVA_KSTAMPS=1): when its workgroups start and what their phases take.
python tools/pyr_stamps.py [width height]   (GPU box)"""
import ctypes as C
import os
import sys

os.environ["ALVA_KSTAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
import torch
import alvaar_amd
from alvaar_amd import synth
from alvaar_amd.capi import lib

w, h = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (1280, 720)
ctx = alvaar_amd.Context(0)
g = torch.from_numpy(synth.frame_gray(synth.texture_canvas(w, h, 7), 1, w, h, noise_seed=7)).cuda()
orb = alvaar_amd.Orb(ctx, w, h, 4000)
lib.alva_debug_kstamps.argtypes = [C.c_void_p]
buf = np.zeros(4096, np.uint64)
for _ in range(5):
    orb.detect_and_compute(g)
assert lib.alva_debug_kstamps(buf.ctypes.data) == 0
names = ["table", "footprint", "taps+L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7"]
for rep in range(3):
    orb.detect_and_compute(g)
    assert lib.alva_debug_kstamps(buf.ctypes.data) == 0
    s = buf.reshape(256, 16).astype(np.int64)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    start = (s[:, 0] - t0) / 100.0
    last = np.max(np.where(s > 0, s, 0), axis=1)
    nph = int((s[0] > 0).sum())
    print(f"rep {rep}: {len(s)} stamped workgroups; start after the first (us): p50 {np.percentile(start, 50):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f};"
          f" life (us): mean {((last - s[:, 0]) / 100.0).mean():.2f} max {((last - s[:, 0]) / 100.0).max():.2f}; end of the last after the first start {(last.max() - t0) / 100.0:.2f}")
    d = np.diff(s[:, :nph], axis=1) / 100.0
    print("   phases (us, mean):", {names[i]: round(float(d[:, i].mean()), 2) for i in range(min(nph - 1, len(names)))})
