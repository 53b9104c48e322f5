#!/usr/bin/env bash
# SQ counters of a command, folded per kernel: tools/pmc_sq.sh <name> <command...>   (GPU box, repo root; counters only, no trace domains)
set -uo pipefail
name=$1; shift
repo=$(pwd)
out=/tmp/pmc_$name
rm -rf "$out"; mkdir -p "$out" "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( cd "$repo" && rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY} -d "$out" -o "$name" --output-format csv -- "$@" ) > "$repo/gpurun_out/${name}_pmc.log" 2>&1
f=$(find "$out" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python3 "$repo/tools/pmc_fold.py" "$f" ${FILTER:-} > "$repo/gpurun_out/${name}_pmc_sq.txt"; cat "$repo/gpurun_out/${name}_pmc_sq.txt" | head -${LINES_OUT:-60}; else echo "no counters"; tail -20 "$repo/gpurun_out/${name}_pmc.log"; fi
