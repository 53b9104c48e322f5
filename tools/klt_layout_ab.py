"""A/B of the single session's tracker layouts (ALVA_TRACK_KLT_LANES = 64 | 32 | 16 | 5, read once per process): sustained frames/s and the
tracker kernel's event-timed average over the steady-state bench stream.  python tools/klt_layout_ab.py   (spawns one process per layout)"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import time
    import bench_common as bc
    from alvaar_amd import capi
    job = bc.SystemJob(0, 7, host_copy=False)
    for _ in range(700):
        job.step()
    t0 = time.perf_counter()
    for _ in range(720):
        job.step()
    dt = time.perf_counter() - t0
    kt = capi.kernel_times(job.step, 200)
    k = [(n, v) for n, v in kt.items() if n.startswith("k_track_klt") or "k_track_klt_w" in n]
    print(f"lanes {sys.argv[1]:>2}: {720 / dt:7.0f} frames/s sustained;", ", ".join(f"{n} {v[1]:.1f} us x{v[0] / 200:.2f}" for n, v in k), flush=True)
else:
    for lanes in ("64", "32", "16", "5", "64", "32"):
        env = dict(os.environ, ALVA_TRACK_KLT_LANES=lanes)
        subprocess.run([sys.executable, __file__, lanes], env=env, check=False)
