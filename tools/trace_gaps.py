"""Per-frame timeline from a rocprofv3 kernel trace of tools/prof_frames.py: for the last frames, start/end of every kernel relative
to the frame's first kernel, per stream.  python tools/trace_gaps.py <kernel_trace.csv> [frames]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0],
       r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ks.sort()
# frame boundary = every k_klt launch (one per frame)
starts = [i for i, k in enumerate(ks) if k[2].startswith("k_klt")]
for f in range(len(starts) - nshow - 1, len(starts) - 1):
    a, b = starts[f], starts[f + 1]
    # include the kernels launched just before k_klt that belong to this frame (pyramid) -- show window [klt start - 60us, next klt start)
    t0 = ks[a][0]
    print(f"--- frame {f}: next frame's k_klt starts at +{(ks[b][0] - t0) / 1e3:.1f} us")
    for s, e, name, q in ks:
        if t0 - 60000 <= s < ks[b][0]:
            print(f"  q{q:>3} {name:28s} start {(s - t0) / 1e3:8.1f}  end {(e - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:6.1f}")
