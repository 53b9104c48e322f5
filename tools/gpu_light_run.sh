# a short GPU run for commits that change host-side code only (the PMC stamp and the kernel statistics of the last consolidated run stay
# valid): tests, then the bench line.   usage (GPU box, repo root): ALVA_COMMIT=<sha> TAG=r5x tools/gpu_light_run.sh
export ALVA_COMMIT=${ALVA_COMMIT:-unknown}
T=${TAG:-r5}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err
cp bench_detail.json gpurun_out/${T}_bench_detail.json
tail -c 200 gpurun_out/${T}_bench.err
wc -c gpurun_out/${T}_bench_line.json
