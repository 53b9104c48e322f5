#!/usr/bin/env bash
# kernel timeline of the steady-state tracking frame (one session): rocprofv3 --kernel-trace over tools/system_sustained.py, last frames
set -uo pipefail
repo=$(pwd); out=/tmp/fg; rm -rf $out; mkdir -p $out "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( cd "$repo" && FRAMES=760 WINDOW=60 rocprofv3 --kernel-trace -d $out -o fg --output-format csv -- python tools/system_sustained.py ) > "$repo/gpurun_out/fg.log" 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("alva_slam::", "").split("(")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
klt = [i for i, k in enumerate(ks) if k[2] == "k_track_klt"]
for f in klt[-6:-3]:
    t0 = ks[f][0]
    print("--- frame")
    prev_end = None
    for s, e, n in ks[f - 3:f + 5]:
        gap = "" if prev_end is None else f" gap {(s - prev_end) / 1e3:6.1f}"
        print(f"  {n:24s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}{gap}")
        prev_end = e
PY
