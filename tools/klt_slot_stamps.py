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
    keep = b != 0
    b = b[keep]
    cnt = buf[8192:8192 + n + 200][keep]
    us = (b & np.uint64(0xffffffff)).astype(np.float64) / 100.0
    code = ((b >> np.uint64(32)) & np.uint64(15)).astype(int)
    prior = ((b >> np.uint64(36)) & np.uint64(1)).astype(int)
    retried = ((b >> np.uint64(37)) & np.uint64(1)).astype(int)
    why = ((b >> np.uint64(40)) & np.uint64(15)).astype(int)
    why2 = ((b >> np.uint64(44)) & np.uint64(15)).astype(int)
    rows.append((us, code, prior, retried, why, why2, cnt))
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

why = np.concatenate([r[4] for r in rows]); why2 = np.concatenate([r[5] for r in rows])
names = {0: "ok", 1: "status 0 (level-0 template / min-eig / out of bounds)", 2: "err > threshold", 3: "not in border", 4: "backward status 0", 5: "fb distance"}
lost = code == 0
print("why the lost slots are lost (first attempt -> retry), count, mean us, max us:")
import collections
c = collections.Counter(zip(ret[lost], why[lost], why2[lost]))
for (r_, a, b_), n in sorted(c.items(), key=lambda kv: -kv[1]):
    m = lost & (ret == r_) & (why == a) & (why2 == b_)
    print(f"  retried {r_}: {names[a]} -> {names[b_] if r_ else '-'}: {n}  mean {us[m].mean():.1f}  max {us[m].max():.1f}")

cnt = np.concatenate([r[6] for r in rows])
if cnt.any():   # the counting build (hipcc -DALVA_KLT_COUNT): LK iterations, window moves, tile restages, levels per slot
    it = (cnt & np.uint64(0xffff)).astype(float); mv = ((cnt >> np.uint64(16)) & np.uint64(0xffff)).astype(float)
    rs = ((cnt >> np.uint64(32)) & np.uint64(0xffff)).astype(float); lv = ((cnt >> np.uint64(48)) & np.uint64(0xff)).astype(float)
    f_tpl = ((cnt >> np.uint64(56)) & np.uint64(3)).astype(int); f_eig = ((cnt >> np.uint64(58)) & np.uint64(3)).astype(int); f_oob = ((cnt >> np.uint64(60)) & np.uint64(3)).astype(int)
    s0 = lost & (why == 1)
    print(f"lost with status 0 in the first attempt: {s0.sum()}; level-0 failures seen in those slots (either attempt): template out of range {int((f_tpl[s0] > 0).sum())}, min-eig / det {int((f_eig[s0] > 0).sum())}, window out of bounds during the iteration {int((f_oob[s0] > 0).sum())}")
    print(f"per slot: iterations {it.mean():.1f}, integer-origin moves {mv.mean():.1f}, tile restages {rs.mean():.2f}, levels {lv.mean():.2f}")
    slow = us > 30
    print(f"slots > 30 us: iterations {it[slow].mean():.1f}, moves {mv[slow].mean():.1f}, restages {rs[slow].mean():.1f}, levels {lv[slow].mean():.1f}, us {us[slow].mean():.1f}")
    A = np.stack([it, mv, rs, lv, np.ones_like(it)], 1)
    coef, *_ = np.linalg.lstsq(A, us, rcond=None)
    print("least squares us = a*iterations + b*moves + c*restages + d*levels + e:", np.round(coef, 3))
    A2 = A[slow]; coef2, *_ = np.linalg.lstsq(A2, us[slow], rcond=None)
    print("same over the slots > 30 us:", np.round(coef2, 3))
