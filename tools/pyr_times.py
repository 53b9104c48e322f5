"""Kernel times (in-library HIP events) of alva_pyramid_build_from_rgba, 640x480 and 1280x720; run once as is (k_pyr_all) and once with
ALVA_PYRAMID_TWO_LAUNCHES=1 (k_level0 + k_pyr_rest).  python tools/pyr_times.py   (GPU box)"""
import sys, time
sys.path.insert(0, ".")
import torch
import alvaar_amd
from alvaar_amd import capi, synth

ctx = alvaar_amd.Context(0)
for w, h in ((640, 480), (1280, 720)):
    rgba = torch.from_numpy(synth.gray_to_rgba(synth.frame_gray(synth.texture_canvas(w, h, seed=9), 2, w, h, noise_seed=4), seed=8)).cuda()
    pyr = alvaar_amd.Pyramid(ctx, w, h, 9, 3)
    g = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    f = lambda: pyr.build_from_rgba(rgba, g)
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000 * 1e6
    kt = capi.kernel_times(f, 500)
    torch.cuda.synchronize()
    print(f"{w}x{h}: {dt:.2f} us per build back to back;", {k: round(v[1], 2) for k, v in kt.items()})
