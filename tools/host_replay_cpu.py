"""The HOST's share of a keyframe, measured WITHOUT a GPU and without the CPU stages' cache footprint: the product's slam/*.cpp over a
TAPE of stage results (oracle/sys_cpu.cpp MemoStages, test infrastructure).  Pass 1 runs the bench stream over the reference's L1 stages
and records what every stage call returned; pass 2 is a fresh map layer over the same frames that gets those results back at memcpy cost
-- what the map layer sees from a device that answers at once.  Its section timers then hold host work only, with the map layer's own
working set in the caches (tools/host_sections_cpu.py times the same sections with OpenCV / Ceres evicting everything in between: 3 - 4 x
higher).  The replayed run's state is compared with the recorded run's at the end.
env: CELL (12), FRAMES (700), WINDOW (300), REPS (3 replays), W/H; BASE_LIB=<another build of libalva_ref.so>: its replays alternate with the
in-tree build's over the same tape and the per-section minima of both are printed side by side (this machine's clock drifts by 10 % between
invocations; only runs inside one invocation compare)"""
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


import oracles
_libs = {}


def run(tape, replay, lib=None):
    if lib:   # the map layer of another build over the same tape (its own copy of every symbol: ctypes opens RTLD_LOCAL)
        if lib not in _libs:
            _libs[lib] = C.CDLL(lib)
        keep, oracles._ref = oracles._ref, _libs[lib]
        try:
            s = sysdiff.CpuSystem(w, h, cell)
        finally:
            oracles._ref = keep
    else:
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
base_lib = os.environ.get("BASE_LIB")
runs, base_runs = [], []
for _ in range(reps):
    if base_lib:
        base_runs.append(run(tape, True, base_lib))
    runs.append(run(tape, True))
for r in runs + base_runs:
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
if base_runs:
    bmin = lambda f, k: min(r[f][k] for r in base_runs)
    print(f"  A/B (minimum of {reps} alternating replays each; base = {base_lib}):")
    print(f"    wall per frame: base {min(r['wall'] for r in base_runs) * 1e6 / win:.1f}  in-tree {min(r['wall'] for r in runs) * 1e6 / win:.1f}")
    for f, keys in (("kf", names16), ("fine", [a for a in runs[0]["fine"] if not a.startswith("#")]), ("frame", names8)):
        for a in keys:
            b0, b1 = bmin(f, a), best(f, a)
            if max(b0, b1) >= 5:
                print(f"    {a:34s} {b0:8.1f} -> {b1:8.1f}  {b1 - b0:+7.1f}")
