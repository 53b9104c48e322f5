"""Phase stamps of k_pose_all (ALVA_KSTAMPS=1): start | compaction phase done | all slices gathered | the host's word seen, for workgroup 0,
the first workgroup without a compaction slice and the last one.  python tools/pose_all_stamps.py   (GPU box)"""
import os, sys
os.environ["ALVA_KSTAMPS"] = "1"
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
import bench_common as bc
from alvaar_amd import system as S
lib = S.lib
lib.alva_debug_kstamps.argtypes = [C.c_void_p]
job = bc.SystemJob(0, 7, host_copy=False)
for _ in range(700):
    job.step()
buf = np.zeros(4096, np.uint64)
lib.alva_debug_kstamps(buf.ctypes.data)
rows = []
for _ in range(60):
    job.step()
    lib.alva_debug_kstamps(buf.ctypes.data)
    b = buf[4048:4072].astype(np.int64).reshape(3, 8)[:, :4] * 0.01
    if (b[1:] > 0).all():
        rows.append(np.where(b > 0, b - b[0, 0], np.nan))
r = np.nanmedian(np.array(rows), 0)
np.set_printoptions(precision=1, suppress=True)
print("frames", len(rows), "median us since workgroup 0's start: [start, compaction done, slices gathered, go seen]")
for name, v in zip(("workgroup 0", "first without a slice", "last"), r):
    print(f"  {name:24s}", v)
