# consolidated GPU run: tests, PMC passes (stamped), bench line (reads the fresh stamp), rocprofv3 kernel stats of the same command
export ALVA_COMMIT=${ALVA_COMMIT:-d34bb1a}
T=${TAG:-r3i}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${T}_pytest.log
tools/pmc_klt.sh > gpurun_out/${T}_pmc.log 2>&1
cp gpurun_out/r3_pmc_track_klt.json profiles/r3_pmc_track_klt.json
python bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --quick > $R/gpurun_out/${T}_bench_prof.log 2>&1
cp $(find /tmp/prof_$T -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_bench_kernel_stats.csv
cd $R
cat gpurun_out/${T}_pytest.log
