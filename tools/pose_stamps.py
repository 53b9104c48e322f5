"""In-kernel phase stamps of k_p3p / k_pnp on the bench stream (ALVA_KSTAMPS=1 is set here).  env: FRAMES (400), SAMPLE (40)"""
import os, sys
os.environ["ALVA_KSTAMPS"] = "1"
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
import bench_detail as bench  # noqa: E402
from alvaar_amd import system as S

lib = S.lib
lib.alva_debug_kstamps.argtypes = [C.c_void_p]
n, sample = int(os.environ.get("FRAMES", "400")), int(os.environ.get("SAMPLE", "40"))
bench.SYSTEM_CELL = int(os.environ.get("CELL", "12"))
job = bench.SystemJob(0, 7, host_copy=False)
buf = np.zeros(4096, np.uint64)
for k in range(n - sample):
    job.step()
lib.alva_debug_kstamps(buf.ctypes.data)
P, Q = [], []
for k in range(sample):
    job.step()
    assert lib.alva_debug_kstamps(buf.ctypes.data) == 0
    b = buf.astype(np.int64)
    wg = b[:2048].reshape(256, 8)
    live = wg[:, 0] > 0
    if not live.any():
        continue
    w = wg[live] * 10.0 / 1000.0   # 100 MHz ticks -> us
    t0 = w[:, 0].min()
    last = np.argmax(w[:, 6])
    d = lambda a, b_: np.median(w[:, a] - w[:, b_])
    P.append([live.sum(), w[:, 0].max() - t0, d(1, 0), d(2, 1), d(3, 2), d(4, 3), (w[:, 4].max() - t0), w[last, 5] - w[last, 4], w[last, 6] - w[last, 5],
              w[last, 6] - t0, (w[:, 1] - w[:, 0]).max(), (w[:, 3] - w[:, 2]).max()])
    ns = int(b[2047])
    st = b[2048:2048 + ns] * 10.0 / 1000.0
    if ns >= 4:
        ev = st[1:-1]                      # eval begin/end pairs
        durs = ev[1::2] - ev[0::2]
        gaps = ev[2::2] - ev[1:-1:2]
        Q.append([ns, st[-1] - st[0], len(durs), durs.mean(), gaps.mean() if len(gaps) else 0, st[1] - st[0], st[-1] - st[-2], st[0] - w[last, 6]])
P, Q = np.array(P), np.array(Q)
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print("k_p3p per frame (median over", len(P), "frames), us:")
for name, v in zip(("workgroups", "start skew", "hypothesis (med)", "score (med)", "radix select (med)", "penalty+fence+arrive (med)", "last arrival since start",
                    "last WG: selection", "last WG: inliers", "kernel span", "hypothesis (max)", "radix select (max)"), np.median(P, 0)):
    print(f"  {name:32s} {v:8.2f}")
print("k_pnp per frame (median over", len(Q), "frames), us:")
for name, v in zip(("stamps", "span", "evals", "eval mean", "between evals mean", "setup before first eval", "after last eval", "gap p3p end -> pnp start"), np.median(Q, 0)):
    print(f"  {name:32s} {v:8.2f}")
print("evals histogram:", np.bincount(Q[:, 2].astype(int)))
