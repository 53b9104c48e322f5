"""Per-frame GPU timeline of the System surface from a rocprofv3 kernel trace of tools/system_sustained.py: for tracking frames (no
keyframe kernels) near the end, start / end / duration of every kernel relative to the frame's k_level0, and the idle gaps between them.
usage (GPU box): rocprofv3 --kernel-trace -d /tmp/tr -o tr --output-format csv -- python tools/system_sustained.py; python tools/system_trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("alva_slam::", "").replace("void ", "").split("(")[0].split("<")[0]) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2] == "k_level0"]
shown, acc, nacc = 0, {}, 0
for f in range(len(starts) - 200, len(starts) - 1):
    a, b = starts[f], starts[f + 1]
    names = [k[2] for k in ks[a:b]]
    if names != ["k_level0", "k_pyr_stage", "k_pyr_stage", "k_pyr_stage", "k_pyr_stage", "k_track_stage_in", "k_track_klt", "k_track_compact", "k_p3p", "k_pnp"]:
        continue
    t0 = ks[a][0]
    prev_end = None
    for s, e, n in ks[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        acc.setdefault(n + ".dur", 0.0); acc.setdefault(n + ".gap_before", 0.0)
        acc[n + ".dur"] += (e - s) / 1e3 / (4 if n == "k_pyr_stage" else 1)
        acc[n + ".gap_before"] += gap / (4 if n == "k_pyr_stage" else 1)
        if shown < 1:
            print(f"  {n:20s} start {(s - t0) / 1e3:8.1f}  end {(e - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:6.1f}  gap before {gap:6.1f}")
        prev_end = e
    acc["frame_period"] = acc.get("frame_period", 0.0) + (ks[b][0] - t0) / 1e3
    acc["chain_end"] = acc.get("chain_end", 0.0) + (ks[b - 1][1] - t0) / 1e3
    shown += 1
    nacc += 1
print(f"averages over {nacc} tracking frames (us):")
for k, v in acc.items():
    print(f"  {k:32s} {v / max(nacc, 1):8.2f}")
