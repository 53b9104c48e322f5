#!/usr/bin/env bash
# HBM traffic + L2 hit rate of the tracking step's kernels: three separate rocprofv3 counter passes (FETCH_SIZE | WRITE_SIZE |
# TCC_HIT_sum TCC_MISS_sum; MI355X_MICROARCH.md "HBM" / "L2" / "PMC slots") over tools/system_sustained.py, folded by
# tools/pmc_klt_fold.py into gpurun_out/r6_pmc_track_klt.json, stamped with the commit ($ALVA_COMMIT, passed in by the caller: the GPU
# box has no .git) and the sha256 of the kernel's sources.  bench.py reports the file's numbers only while the sources are unchanged.
# usage (GPU box, repo root): ALVA_COMMIT=<sha> tools/pmc_klt.sh
set -uo pipefail
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export FRAMES=${FRAMES:-700} WINDOW=${WINDOW:-100}
# counter collection makes every launch synchronous: the fused pose launch (which waits for the host's answer WHILE it runs) would only
# ever time out into its fallback; the tracker's counters do not need it
export ALVA_NO_POSE_ALL=1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=${c%% *}
  out=/tmp/pmc_klt_$tag; rm -rf "$out"; mkdir -p "$out"
  ( cd "$repo" && rocprofv3 --kernel-trace --pmc $c -d "$out" -o klt --output-format csv -- python tools/system_sustained.py ) > "$repo/gpurun_out/pmc_klt_$tag.log" 2>&1
done
python3 "$repo/tools/pmc_klt_fold.py" /tmp/pmc_klt_FETCH_SIZE /tmp/pmc_klt_WRITE_SIZE /tmp/pmc_klt_TCC_HIT_sum "$repo/gpurun_out/r6_pmc_track_klt.json"
