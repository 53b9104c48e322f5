#!/usr/bin/env bash
# HBM traffic per kernel of a command: two separate rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE), folded by tools/pmc_summary.py
# usage: tools/pmc_traffic.sh <name> <command...>   (GPU box, repo root) -> gpurun_out/<name>_pmc_traffic.json
set -uo pipefail
name=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=/tmp/pmc_${name}_$c; rm -rf "$out"; mkdir -p "$out"
  ( cd "$repo" && rocprofv3 --kernel-trace --pmc $c -d "$out" -o "$name" --output-format csv -- "$@" ) > "$repo/gpurun_out/${name}_pmc_$c.log" 2>&1
done
f=$(find /tmp/pmc_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then PMC_COMMAND="$*" python3 "$repo/tools/pmc_summary.py" "$f" "$w" "$repo/gpurun_out/${name}_pmc_traffic.json" | head -${LINES_OUT:-16}; else echo "counter pass failed"; tail -5 "$repo"/gpurun_out/${name}_pmc_*.log; fi
