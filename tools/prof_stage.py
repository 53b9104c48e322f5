"""Run one hot-path stage repeatedly (for `rocprofv3 --kernel-trace --stats -- python tools/prof_stage.py <stage> [reps]`)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench_detail as bench  # noqa: E402


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    job = bench.FrameJob(0, 7)
    ctx = job.ctx
    cur, prev = job.pyr[0], job.pyr[1]
    prev.build_from_rgba(job.frames[1], job.gray)
    cur.build_from_rgba(job.frames[2], job.gray)
    tracked, _ = ctx.fbklt_track(prev, cur, job.pts, job.pts, 3)
    desc, _ = ctx.describe(job.gray, tracked)
    fns = {
        "detect": lambda: ctx.detect_grid(job.gray, 12, max_quality=0.001),
        "detect40": lambda: ctx.detect_grid(job.gray, 40, max_quality=0.001),
        "pyramid": lambda: cur.build_from_rgba(job.frames[2], job.gray),
        "klt": lambda: ctx.fbklt_track(prev, cur, job.pts, job.pts, 3),
        "describe": lambda: ctx.describe(job.gray, tracked),
        "match": lambda: ctx.bf_match_hamming(desc, job.prev_desc),
        "p3p": lambda: ctx.p3p_lmeds(job.bv, job.wpt, 100, 3.0, job.K[0], job.K[1]),
        "pnp": lambda: ctx.pnp_refine(job.uv, job.wpt, job.pose0, job.K),
    }
    if stage == "ba":
        from alvaar_amd import synth
        pb = synth.make_ba_problem(20, 3000, 42)
        for _ in range(reps):
            ctx.local_ba(pb, 5, 0.0)
    elif stage == "orb":
        import alvaar_amd
        orb = alvaar_amd.Orb(ctx, bench.W, bench.H, 2000)
        for _ in range(reps):
            orb.detect_and_compute(job.gray)
    elif stage == "all":
        for _ in range(reps):
            job.step()
    else:
        for _ in range(reps):
            fns[stage]()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
