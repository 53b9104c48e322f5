#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of a command; copies the kernel-stats CSV to gpurun_out/<name>_kernel_stats.csv
# usage: tools/prof_kernels.sh <name> <command...>      (run on the GPU box from the repo root)
set -uo pipefail
name=$1; shift
repo=$(pwd)
out=/tmp/prof_$name
rm -rf "$out"; mkdir -p "$out" "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( cd "$repo" && rocprofv3 --kernel-trace --stats -d "$out" -o "$name" --output-format csv -- "$@" ) > "$repo/gpurun_out/${name}_prof.log" 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$repo/gpurun_out/${name}_kernel_stats.csv"; head -30 "$f"; else echo "no kernel stats produced"; tail -20 "$repo/gpurun_out/${name}_prof.log"; fi
