import os, sys
sys.path.insert(0, ".")
import bench
job = bench.SystemJob(0, 7, host_copy=False)
for _ in range(700): job.step()
os.environ["ALVA_BA_TIMING"] = "1"
for _ in range(40): job.step()
