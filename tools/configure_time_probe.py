import sys, time
sys.path.insert(0, ".")
from alvaar_amd.system import AlvaAR
import os
for cell in (40, 12):
    for i in range(2):
        t0 = time.perf_counter(); a = AlvaAR(640, 480, cell_size=cell, random_sampling=False); t1 = time.perf_counter(); a.close()
        print("cell", cell, "configure", round((t1 - t0) * 1e3, 1), "ms")
os.environ["ALVA_NO_WARMUP"] = "1"
t0 = time.perf_counter(); a = AlvaAR(640, 480, cell_size=12, random_sampling=False); t1 = time.perf_counter(); a.close()
print("no warm-up", round((t1 - t0) * 1e3, 1), "ms")
