"""Kernel times of the §8(f) entry points (HIP events through alva_prof_*)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import alvaar_amd
from alvaar_amd import synth, capi

ctx = alvaar_amd.Context(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (w, h) in ((640, 480), (1280, 720)):
    g = d(synth.frame_gray(synth.texture_canvas(w, h, 1), 2, w, h, noise_seed=1))
    fn = lambda: ctx.clahe(g)
    fn(); torch.cuda.synchronize()
    kt = capi.kernel_times(fn, 50)
    P = w * h
    print(f"clahe {w}x{h}:", {k: round(v[1], 2) for k, v in kt.items()}, f"apply: {3 * P / (kt['k_clahe_apply'][1] * 1e-6) / 1e9:.0f} GB/s of 3P")
pb = synth.make_triangulation_problem(2000, 4, 3)
T = torch.zeros((4, 36), dtype=torch.float64, device="cuda"); T[:, 0] = T[:, 4] = T[:, 8] = 1; T[:, 12] = T[:, 16] = T[:, 20] = 1; T[:, 24] = T[:, 28] = T[:, 32] = 1; T[:, 9] = 0.5
args = (T, d(pb["group"]), d(pb["bvl"]), d(pb["bvr"]), d(pb["unpxl"]), d(pb["unpxr"]), pb["K"])
fn = lambda: ctx.triangulate(*args)
fn(); torch.cuda.synchronize()
print("triangulate 2000 pts:", {k: round(v[1], 2) for k, v in capi.kernel_times(fn, 50).items()})
px = torch.rand((2000, 2), device="cuda") * 400 + 50
fn = lambda: ctx.undistort_points(px, (520., 515., 318.5, 241.25), (-0.28, 0.07, 2e-4, -3e-4))
fn(); torch.cuda.synchronize()
print("undistort 2000 pts:", {k: round(v[1], 2) for k, v in capi.kernel_times(fn, 50).items()})
