# ORB (configs[2]) after a kernel change: parity tests, then per-kernel times with the fused pyramid launch and with the per-level chain
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_track_batch.py -x -q 2>&1 | tail -8 | tee gpurun_out/orb_ab_pytest.log
for m in fused chain; do
  echo "== pyramid: $m"
  ALVA_ORB_PYRAMID=$m timeout 300 python tools/orb_kernels.py 1280 720 4000
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/orb_ab_720p.txt
timeout 300 python tools/orb_kernels.py 640 480 2000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/orb_ab_480p.txt
