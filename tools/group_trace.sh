#!/usr/bin/env bash
# usage (GPU box, repo root): tools/group_trace.sh <tag> <S> <W> <steps> <lockstep>
set -uo pipefail
tag=$1; S=$2; W=$3; steps=$4; lock=$5; lanes=${6:-2}
repo=$(pwd)
out=/tmp/gt_$tag; rm -rf "$out"; mkdir -p "$out" "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( cd "$repo" && rocprofv3 --kernel-trace -d "$out" -o gt --output-format csv -- python tools/group_trace.py $S $W $steps $lock $lanes ) > "$repo/gpurun_out/gt_${tag}.log" 2>&1
tail -1 "$repo/gpurun_out/gt_${tag}.log"
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
# tracker launches in the window: lock-step = W per group step (one per worker), otherwise S
if [ "$lock" = "1" ]; then n=$((lanes * steps * ((S / lanes + 3) / 4))); else n=$((S * steps)); fi
python3 "$repo/tools/group_trace_fold.py" "$f" $n ${TL:-0} ${BIN:-} | tee "$repo/gpurun_out/gt_${tag}_fold.txt"
