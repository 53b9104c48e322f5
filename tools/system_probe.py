"""Per-frame wall time of the System surface (host-fed RGBA frames, the reference's shipped configuration)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from alvaar_amd import synth
from alvaar_amd.system import AlvaAR

w, h, Z = 640, 480, 4.0
ar = AlvaAR.Initialize(w, h)
K = ar.intrinsics
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(40)]
ar.findCameraPose(frames[0])
ids, px, is3d = ar.keypoints()
X = np.stack([(px[:, 0] - K["cx"]) / K["fx"] * Z, (px[:, 1] - K["cy"]) / K["fy"] * Z, np.full(len(px), Z)], 1)
ar.set_map_points(ids, X)
for k in range(1, 8):
    ar.findCameraPose(frames[k])
t0 = time.perf_counter()
ok = 0
for k in range(8, 40):
    pose, st = ar.findCameraPose(frames[k])
    ok += st == 1
dt = (time.perf_counter() - t0) / 32
print(f"System::findCameraPose: {dt * 1e3:.3f} ms/frame ({1 / dt:.0f} frames/s), {ok}/32 poses, {len(ids)} keypoints (cell 40), host-fed 1.2 MB RGBA per frame")
