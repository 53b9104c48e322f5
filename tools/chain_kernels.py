"""Event-timed averages of the tracking chain's kernels over one period of the bench stream in the steady state, and the sustained rate:
python tools/chain_kernels.py   (GPU box; A/B: tools/ab_run.sh python tools/chain_kernels.py)"""
import sys, time
sys.path.insert(0, ".")
import bench_common as bc
from alvaar_amd import capi

job = bc.SystemJob(0, 7, host_copy=False)
for _ in range(700):
    job.step()
t0 = time.perf_counter()
for _ in range(796):
    job.step()
fps = 796 / (time.perf_counter() - t0)
kt = capi.kernel_times(job.step, 398)
chain = ("k_track_stage_in", "k_track_klt", "k_track_compact", "k_p3p_s", "k_pnp", "k_level0<true>", "k_pyr_rest")
print(f"{fps:.0f} frames/s |", "  ".join(f"{k} {kt[k][1]:.2f}" for k in chain if k in kt))
