"""Concurrency on one GPU: the empty-launch rate and the tracking step of S sessions on S host threads at once (what caps system_streams / system_group)"""
import sys, time, threading
sys.path.insert(0, ".")
import bench_detail as bench
import numpy as np
def run(S):
    jobs = [bench.SystemJob(0, 7, host_copy=False)]
    from alvaar_amd.system import AlvaAR
    for i in range(1, S):
        j = bench.SystemJob.__new__(bench.SystemJob); j.__dict__.update(jobs[0].__dict__)
        j.ar = AlvaAR(bench.W, bench.H, device=0, cell_size=bench.SYSTEM_CELL, random_sampling=False); j.k = -1; j.status_hist = [0,0,0,0]
        jobs.append(j)
    for j in jobs:
        for _ in range(650): j.step()
    jobs[0].ar.timing(); jobs[0].ar.timing_keyframe()
    bar = threading.Barrier(S)
    def work(j):
        bar.wait()
        for _ in range(300): j.step()
    th = [threading.Thread(target=work, args=(j,)) for j in jobs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    sec = jobs[0].ar.timing(); kf = jobs[0].ar.timing_keyframe()
    print(S, "sessions:", round(S * 300 / dt), "f/s; session 0 us/frame:", {a: round(1e6 * b / 300, 1) for a, b in sec.items()})
    print("   per keyframe (17 kf):", {a: round(1e6 * b / 17) for a, b in kf.items() if b > 50e-6 * 17})
    for j in jobs: j.ar.close()
for S in (1, 4, 8):
    run(S)
