"""Soak of the headline loop with the fused pose launch: SECONDS (default 20) of alva_system_find_camera_pose_device in the steady state;
prints the rate per third and how often the queued launch was answered in time / gave up and was replaced (alva_debug_pose_all_stats).
python tools/pose_all_soak.py [seconds]   (GPU box)"""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import bench_common as bc
from alvaar_amd.capi import lib

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
job = bc.SystemJob(0, 7, host_copy=False)
job.warm_to_steady_state()
have = hasattr(lib, "alva_debug_pose_all_stats")   # (an older build loaded through ALVA_LIB for an A/B has no such counters)
st = (C.c_long * 3)()
if have:
    lib.alva_debug_pose_all_stats.argtypes = [C.c_void_p]
    lib.alva_debug_pose_all_stats(st)
base = list(st)
for part in range(3):
    t0 = time.perf_counter()
    n = 0
    worst = 0.0
    while time.perf_counter() - t0 < secs / 3:
        t1 = time.perf_counter()
        job.step()
        worst = max(worst, time.perf_counter() - t1)
        n += 1
    dt = time.perf_counter() - t0
    if have:
        lib.alva_debug_pose_all_stats(st)
    print(f"part {part}: {n / dt:.0f} frames/s over {n} frames, slowest frame {worst * 1e3:.2f} ms; fused pose launches queued {st[0] - base[0]}, answered with go {st[1] - base[1]}, "
          f"gave up -> separate launches {st[2] - base[2]}; status histogram {job.status_hist}; state {list(job.ar.state())[6:8]} keyframes / map points", flush=True)
