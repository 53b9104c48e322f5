"""Timing probe: fb-KLT kernel time as a function of the iteration cap (tail analysis)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench_detail as bench

job = bench.FrameJob(0, seed=7)
ctx = job.ctx
cur, prev = job.pyr[0], job.pyr[1]
prev.build_from_rgba(job.frames[1], job.gray)
cur.build_from_rgba(job.frames[2], job.gray)
for mi in (30, 20, 10, 5, 2, 1):
    for lv in (3, 0):
        ctx.fbklt_track(prev, cur, job.pts, job.pts, lv, max_iters=mi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.fbklt_track(prev, cur, job.pts, job.pts, lv, max_iters=mi)
        e1.record()
        torch.cuda.synchronize()
        print(f"max_iters={mi:2d} levels={lv}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
