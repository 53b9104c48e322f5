"""Per-frame wall time of the System surface (host-fed RGBA frames in, pose out) from a cold start: tracking frames and keyframe
frames separately.  CELL=40 is the reference's shipped configuration (192 keypoints), CELL=12 the 2000-keypoint workload."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from alvaar_amd import synth
from alvaar_amd.system import AlvaAR

w, h = 640, 480
cell = int(os.environ.get("CELL", "40"))
n = int(os.environ.get("FRAMES", "150"))
ar = AlvaAR(w, h, cell_size=cell, random_sampling=False)
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h)) for k in range(n)]
track, kf, init = [], [], []
prev_kf = -1
sect_track, sect_kf = {}, {}
ar.timing()
fixed = np.empty_like(frames[0])  # one frame buffer reused like the JS wrapper's memImg (page-locked by the library on its second use)
for k in range(n):
    t0 = time.perf_counter()
    np.copyto(fixed, frames[k])   # src/system.js:175 memImg.write(frame.data) -- part of the host-fed cost
    pose, st = ar.findCameraPose(fixed, 33.0 * k)
    dt = time.perf_counter() - t0
    s = ar.state()
    tm = ar.timing()
    if st != 1:
        init.append(dt)
    elif s[11] != prev_kf:
        kf.append(dt)
        for a, b in tm.items():
            sect_kf[a] = sect_kf.get(a, 0.0) + b
    else:
        track.append(dt)
        for a, b in tm.items():
            sect_track[a] = sect_track.get(a, 0.0) + b
    prev_kf = s[11]
s = ar.state()
ms = lambda v: (1e3 * float(np.median(v)), 1e3 * float(np.mean(v))) if v else (0.0, 0.0)
print(f"cell {cell}: {s[2]} keypoints ({s[4]} 3-D), {s[6]} keyframes, {ar.counters()}")
print(f"  tracking frames  {len(track):4d}: median {ms(track)[0]:.3f} ms  mean {ms(track)[1]:.3f} ms  -> {1e3 / ms(track)[1]:.0f} frames/s")
print(f"  keyframe frames  {len(kf):4d}: median {ms(kf)[0]:.3f} ms  mean {ms(kf)[1]:.3f} ms")
print(f"  initialising     {len(init):4d}: median {ms(init)[0]:.3f} ms")
tot = sum(track) + sum(kf)
print(f"  whole stream after initialisation: {(len(track) + len(kf)) / tot:.0f} frames/s (host-fed 1.2 MB RGBA per frame)")
print("  sections, us per tracking frame:", {a: round(1e6 * b / max(len(track), 1), 1) for a, b in sect_track.items()})
print("  sections, us per keyframe frame:", {a: round(1e6 * b / max(len(kf), 1), 1) for a, b in sect_kf.items()})
print("  keyframe detail, us per keyframe:", {a: round(1e6 * b / max(len(kf), 1), 1) for a, b in ar.timing_keyframe().items()})
