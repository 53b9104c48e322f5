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
if len(sys.argv) > 3 and float(sys.argv[3]) > 0:   # timeline of the last <ms> milliseconds of the window
    span = float(sys.argv[3]) * 1e6
    tl = [k for k in win if k[0] >= t1 - span]
    base = tl[0][0]
    print(f"--- timeline of the last {sys.argv[3]} ms ({len(tl)} launches)")
    for s, e, n, q in tl:
        print(f"  q{q:>3} {n:34s} start {(s - base) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}")
if len(sys.argv) > 4:   # coarse histogram of the whole window: per bin (ms) the busy fraction (union), launches, mean concurrency
    binw = float(sys.argv[4]) * 1e6
    nb = int((t1 - t0) / binw) + 1
    busy_b, sum_b, cnt_b = [0.0] * nb, [0.0] * nb, [0] * nb
    ev = []
    for s, e, n, q in win:
        cnt_b[int((s - t0) / binw)] += 1
        b0, b1 = int((s - t0) / binw), int((e - t0) / binw)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
            sum_b[b] += max(0.0, hi - lo)
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last = 0, t0
    for t, d in ev:
        if depth > 0:
            a, bb = last, t
            b0, b1 = int((a - t0) / binw), int((bb - t0) / binw)
            for b in range(b0, min(b1, nb - 1) + 1):
                lo, hi = max(a, t0 + b * binw), min(bb, t0 + (b + 1) * binw)
                busy_b[b] += max(0.0, hi - lo)
        depth += d
        last = t
    print(f"--- per {sys.argv[4]} ms bin: busy fraction | launches | mean kernels in flight")
    print(" ".join(f"{busy_b[b] / binw:.2f}|{cnt_b[b]}|{sum_b[b] / binw:.1f}" for b in range(nb)))
