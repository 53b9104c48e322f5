import sys, time, threading, ctypes as C
sys.path.insert(0, ".")
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import alvaar_amd
from alvaar_amd.capi import lib, check
lib.alva_microbench_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
for T in (1, 2, 4, 8):
    ctxs = [alvaar_amd.Context(0, own_stream=True) for _ in range(T)]
    res = [None] * T
    def run(i):
        a = C.c_double(0)
        for _ in range(3):
            check(lib.alva_microbench_launch(ctxs[i].h, 2000, C.byref(a), None))
        res[i] = a.value
    th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(T, "threads: us per launch per thread", [round(r, 2) for r in res], "aggregate launches/s ~", round(T * 1e6 / (sum(res) / T)))
