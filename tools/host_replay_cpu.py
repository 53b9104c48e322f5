"""The HOST's share of a keyframe, measured WITHOUT a GPU and without the CPU stages' cache footprint: the product's slam/*.cpp over a
TAPE of stage results (oracle/sys_cpu.cpp MemoStages, test infrastructure).  Pass 1 runs the bench stream over the reference's L1 stages
and records what every stage call returned; pass 2 is a fresh map layer over the same frames that gets those results back at memcpy cost
-- what the map layer sees from a device that answers at once.  Its section timers then hold host work only, with the map layer's own
working set in the caches (tools/host_sections_cpu.py times the same sections with OpenCV / Ceres evicting everything in between: 3 - 4 x
higher).  The replayed run's state is compared with the recorded run's at the end.
env: CELL (12), FRAMES (700), WINDOW (300), REPS (3: replays, the best per section is printed too), W/H"""
import os, sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import ctypes as C
import sysdiff
from alvaar_amd import synth
from alvaar_amd.system import AlvaAR

n, win, cell = int(os.environ.get("FRAMES", "700")), int(os.environ.get("WINDOW", "300")), int(os.environ.get("CELL", "12"))
w, h, reps = int(os.environ.get("W", "640")), int(os.environ.get("H", "480")), int(os.environ.get("REPS", "3"))
NF = 200
canvas = synth.texture_canvas(w, h, 7)
frames = [synth.gray_to_rgba(synth.frame_gray(canvas, k, w, h, noise_seed=11)) for k in range(NF)]
period = 2 * (NF - 1)
idx = lambda k: (k % period) if (k % period) < NF else period - (k % period)
names8 = ("upload+pyramid", "gather", "track_step", "track_apply", "pose_wait", "pose_apply+kf_check", "keyframe_create", "mapping")
names16 = ("prepare", "describe_tracked", "detect", "describe_new", "insert+copy", "triangulate", "covisibility", "local_map_matching",
           "optimize", "(match stage)", "(BA stage)", "(BA build)", "(BA solves+sweep)", "(BA write-back)", "(BA culling)", "(descriptor medoids)")


def run(tape, replay):
    s = sysdiff.CpuSystem(w, h, cell)
    L = s.L
    L.syscpu_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.syscpu_timing_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.syscpu_attach_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.syscpu_attach_tape(s.h, tape, 1 if replay else 0)
    sec, kf, fine = np.zeros(8), np.zeros(16), np.zeros(32)
    t0 = time.perf_counter()
    for k in range(n):
        if k == n - win:
            L.syscpu_timing(s.h, sec.ctypes.data, kf.ctypes.data, 1)
            L.syscpu_timing_fine(s.h, fine.ctypes.data, 1)
            kf0 = int(s.state()[11])
            tw = time.perf_counter()
        s.step(frames[idx(k)], 33.0 * k)
    wall = time.perf_counter() - tw
    L.syscpu_timing(s.h, sec.ctypes.data, kf.ctypes.data, 1)
    L.syscpu_timing_fine(s.h, fine.ctypes.data, 1)
    nkf = int(s.state()[11]) - kf0
    st = [int(v) for v in s.state()]
    out = dict(state=st, nkf=nkf, wall=wall, total=time.perf_counter() - t0,
               frame={a: 1e6 * b / win for a, b in zip(names8, sec)},
               kf={a: 1e6 * b / max(nkf, 1) for a, b in zip(names16, kf)},
               fine={a: (1e6 if not a.startswith("#") else 1) * b / max(nkf, 1) for a, b in zip(AlvaAR.FINE_NAMES, fine) if a})
    del s
    return out


L0 = sysdiff.CpuSystem(w, h, cell).L
L0.syscpu_tape_new.restype = C.c_void_p
L0.syscpu_tape_bytes.restype = C.c_longlong
L0.syscpu_tape_bytes.argtypes = [C.c_void_p]
tape = C.c_void_p(L0.syscpu_tape_new())
rec = run(tape, False)
print(f"recorded {n} frames in {rec['total']:.1f} s ({L0.syscpu_tape_bytes(tape) / 1e6:.0f} MB of stage results); last {win}: {rec['nkf']} keyframes; "
      f"keypoints {rec['state'][2]} ({rec['state'][4]} 3-D), keyframes in map {rec['state'][6]}, map points {rec['state'][7]}")
runs = [run(tape, True) for _ in range(reps)]
for r in runs:
    assert r["state"] == rec["state"], ("the replayed run ended in another state", r["state"], rec["state"])
best = lambda f, k: min(r[f][k] for r in runs)
med = lambda f, k: float(np.median([r[f][k] for r in runs]))
print(f"replayed {reps} x: {min(r['wall'] for r in runs) * 1e6 / win:.1f} us per frame of wall clock over the window (host only)")
print("  us per frame:", {a: round(med("frame", a), 1) for a in names8})
d = {a: med("kf", a) for a in names16}
print("  us per keyframe (median of the replays):", {a: round(b, 1) for a, b in d.items()})
host = {"prepare": d["prepare"], "create: stages' host side": d["describe_tracked"] + d["detect"] + d["describe_new"],
        "insert+copy (net)": d["insert+copy"] - d["describe_tracked"] - d["detect"] - d["describe_new"],
        "triangulate": d["triangulate"], "covisibility": d["covisibility"], "matching": d["local_map_matching"], "BA build": d["(BA build)"],
        "BA solves+sweep": d["(BA solves+sweep)"], "BA write-back": d["(BA write-back)"],
        "keyframe filter": d["optimize"] - d["(BA build)"] - d["(BA solves+sweep)"] - d["(BA write-back)"] - d["(BA culling)"]}
print("  HOST-ONLY us per keyframe:", {a: round(b, 1) for a, b in host.items()}, "sum", round(sum(host.values()), 1))
print("  fine, per keyframe (us | counts):", {a: round(med("fine", a), 1) for a in runs[0]["fine"]})
