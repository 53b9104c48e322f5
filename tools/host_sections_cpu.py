"""Host-side map-layer sections per keyframe, measured WITHOUT a GPU: the product's slam/*.cpp over the reference's L1 stages
(oracle/_ref: syscpu_*, test infrastructure) on the bench stream.  The stage calls are slow there (CPU OpenCV / Ceres), the host-only
sections are what this prints.  env: CELL (12), FRAMES (700), WINDOW (300), W/H"""
import os, sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import ctypes as C
import sysdiff
from alvaar_amd import synth

n, win, cell = int(os.environ.get("FRAMES", "700")), int(os.environ.get("WINDOW", "300")), int(os.environ.get("CELL", "12"))
w, h = int(os.environ.get("W", "640")), int(os.environ.get("H", "480"))
NF = 200
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(NF)]
period = 2 * (NF - 1)
idx = lambda k: (k % period) if (k % period) < NF else period - (k % period)
s = sysdiff.CpuSystem(w, h, cell)
L = s.L
L.syscpu_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.syscpu_timing_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
sec, kf, fine = np.zeros(8), np.zeros(16), np.zeros(32)
names8 = ("upload+pyramid", "gather", "track_step", "track_apply", "pose_wait", "pose_apply+kf_check", "keyframe_create", "mapping")
names16 = ("prepare", "describe_tracked", "detect", "describe_new", "insert+copy", "triangulate", "covisibility", "local_map_matching",
           "optimize", "(match stage)", "(BA stage)", "(BA build)", "(BA solves+sweep)", "(BA write-back)", "(BA culling)", "(descriptor medoids)")
t0 = time.time()
for k in range(n):
    if k == n - win:
        L.syscpu_timing(s.h, sec.ctypes.data, kf.ctypes.data, 1)
        L.syscpu_timing_fine(s.h, fine.ctypes.data, 1)
        kf0 = int(s.state()[11])
    s.step(frames[idx(k)], 33.0 * k)
L.syscpu_timing(s.h, sec.ctypes.data, kf.ctypes.data, 1)
nkf = int(s.state()[11]) - kf0
st = s.state()
print(f"{n} frames in {time.time() - t0:.1f} s; last {win}: {nkf} keyframes; keypoints {st[2]} ({st[4]} 3-D), keyframes in map {st[6]}, map points {st[7]}, local map {st[13]}")
print("  us per frame (host sections incl. CPU stages):", {a: round(1e6 * b / win, 1) for a, b in zip(names8, sec)})
d = {a: 1e6 * b / max(nkf, 1) for a, b in zip(names16, kf)}
print("  us per keyframe:", {a: round(b, 1) for a, b in d.items()})
host = {"prepare": d["prepare"], "medoids": d["(descriptor medoids)"],
        "insert+copy (net)": d["insert+copy"] - d["describe_tracked"] - d["detect"] - d["describe_new"] - d["(descriptor medoids)"],
        "covisibility": d["covisibility"], "matching (net of stage)": d["local_map_matching"] - d["(match stage)"], "BA build": d["(BA build)"],
        "BA sweep (net of stage)": d["(BA solves+sweep)"] - d["(BA stage)"], "BA write-back": d["(BA write-back)"],
        "keyframe filter": d["optimize"] - d["(BA build)"] - d["(BA solves+sweep)"] - d["(BA write-back)"] - d["(BA culling)"]}
print("  HOST-ONLY us per keyframe:", {a: round(b, 1) for a, b in host.items()}, "sum", round(sum(host.values()), 1))
L.syscpu_timing_fine(s.h, fine.ctypes.data, 1)
from alvaar_amd.system import AlvaAR
print("  fine, per keyframe (us | counts):", {a: round((1e6 if not a.startswith("#") else 1) * b / max(nkf, 1), 1) for a, b in zip(AlvaAR.FINE_NAMES, fine) if a})
