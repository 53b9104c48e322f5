# consolidated GPU run: tests, PMC passes (stamped), bench line (reads the fresh stamp), rocprofv3 kernel stats of the same command,
# ORB (configs[2]) kernel times + HBM traffic.   usage (GPU box, repo root): ALVA_COMMIT=<sha> TAG=r5x tools/gpu_final_run.sh
export ALVA_COMMIT=${ALVA_COMMIT:-unknown}
T=${TAG:-r6}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${T}_pytest.log
tools/pmc_klt.sh > gpurun_out/${T}_pmc.log 2>&1
cp gpurun_out/r6_pmc_track_klt.json profiles/r6_pmc_track_klt.json
python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err
cp bench_detail.json gpurun_out/${T}_bench_detail.json
tail -c 300 gpurun_out/${T}_bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --quick > $R/gpurun_out/${T}_bench_prof.log 2>&1
cp $(find /tmp/prof_$T -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_kernel_stats_bench.csv
cd $R
python tools/orb_kernels.py 1280 720 4000 > gpurun_out/${T}_orb_kernels_720p.txt 2>&1
LINES_OUT=12 tools/pmc_traffic.sh ${T}_orb720 python tools/orb_kernels.py 1280 720 4000 > gpurun_out/${T}_orb720_pmc.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
cat gpurun_out/${T}_pytest.log
wc -c gpurun_out/${T}_bench_line.json
# optional: the sustained loop under the in-tree build against every ab_variants/lib_*.so (ALVA_LIB), alternating, when AB=1
if [ "${AB:-0}" = 1 ] && ls ab_variants/lib_*.so >/dev/null 2>&1; then
  bash tools/variant_run.sh "python tools/system_sustained.py" > gpurun_out/${T}_ab.txt 2>&1
  grep -n "^==\|frames/s" gpurun_out/${T}_ab.txt | cut -c1-160
fi
