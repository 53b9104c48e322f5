"""Fold a rocprofv3 kernel trace of tools/group_trace.py: the window of the last N launches of the tracker kernel (k_track_klt /
k_track_klt_multi).  python tools/group_trace_fold.py <kernel_trace.csv> <tracker launches in the window>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nwin = int(sys.argv[2])


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
trk = [i for i, k in enumerate(ks) if k[2].startswith("k_track_klt") and "retry" not in k[2]]
t0 = ks[trk[-nwin]][0]
win = [k for k in ks if k[0] >= t0]
t1 = max(k[1] for k in win)
# union of intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in win)
print(f"window {1e-6 * (t1 - t0):.2f} ms, {len(win)} launches, {len(set(k[3] for k in win))} queues; GPU busy (union) {busy / (t1 - t0):.3f}, "
      f"mean concurrency while busy {tot / busy:.2f}, launches/s {len(win) / ((t1 - t0) * 1e-9):.0f}")
acc = defaultdict(lambda: [0, 0])
for s, e, n, _ in win:
    acc[n][0] += 1
    acc[n][1] += e - s
print(f"{'kernel':34s} {'calls':>7s} {'mean us':>9s} {'share of kernel time':>8s}")
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:34s} {c:7d} {d / c / 1e3:9.1f} {d / tot:8.3f}")
if len(sys.argv) > 3:   # timeline of the last <ms> milliseconds of the window
    span = float(sys.argv[3]) * 1e6
    tl = [k for k in win if k[0] >= t1 - span]
    base = tl[0][0]
    print(f"--- timeline of the last {sys.argv[3]} ms ({len(tl)} launches)")
    for s, e, n, q in tl:
        print(f"  q{q:>3} {n:34s} start {(s - base) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}")
