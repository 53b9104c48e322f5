"""Per-slot wall time of the tracker launch (ALVA_KLT_STAMPS=1) over the steady-state bench stream: what the slowest slots of a frame
are, and how much of the launch they decide.  python tools/klt_slot_stamps.py   (GPU box)"""
import ctypes as C
import os
import sys

os.environ["ALVA_KLT_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
import bench_common as bc
from alvaar_amd.capi import lib

job = bc.SystemJob(0, 7, host_copy=False)
for _ in range(700):
    job.step()
lib.alva_debug_klt_stamps.argtypes = [C.c_void_p]
buf = np.zeros(16384, np.uint64)
rows = []
for f in range(40):
    job.step()
    assert lib.alva_debug_klt_stamps(buf.ctypes.data) == 0
    n = int(job.ar.state()[2])
    b = buf[:n + 200]
    b = b[b != 0]
    us = (b & np.uint64(0xffffffff)).astype(np.float64) / 100.0
    code = ((b >> np.uint64(32)) & np.uint64(15)).astype(int)
    prior = ((b >> np.uint64(36)) & np.uint64(1)).astype(int)
    retried = ((b >> np.uint64(37)) & np.uint64(1)).astype(int)
    rows.append((us, code, prior, retried))
us = np.concatenate([r[0] for r in rows]); code = np.concatenate([r[1] for r in rows]); prior = np.concatenate([r[2] for r in rows]); ret = np.concatenate([r[3] for r in rows])
print(f"slots per frame {len(us) / len(rows):.0f}; slot time us: mean {us.mean():.1f} p50 {np.percentile(us, 50):.1f} p90 {np.percentile(us, 90):.1f} p99 {np.percentile(us, 99):.1f} p99.9 {np.percentile(us, 99.9):.1f} max {us.max():.1f}")
print("per-frame max (us):", np.round([r[0].max() for r in rows], 1))
for name, m in (("from projection, ok (code 1)", code == 1), ("full pyramid ok (code 2)", code == 2), ("retried ok (code 3)", code == 3), ("lost after prior+retry", (code == 0) & (ret == 1)), ("lost on full pyramid", (code == 0) & (ret == 0))):
    if m.any():
        print(f"  {name:32s} share {m.mean():.4f}  mean {us[m].mean():6.1f}  p99 {np.percentile(us[m], 99):6.1f}  max {us[m].max():6.1f}")
top = [np.sort(r[0])[::-1][:8] for r in rows[:6]]
print("eight slowest slots of six frames:", [list(np.round(t, 1)) for t in top])
thr = [np.mean([(r[0] > x).sum() for r in rows]) for x in (20, 30, 40, 50)]
print("slots per frame slower than 20 / 30 / 40 / 50 us:", np.round(thr, 1))
