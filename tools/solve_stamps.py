"""local-BA bench line, per-kernel times and (ALVA_KSTAMPS=1) the in-kernel phase stamps of k_solve on the 20 KF x 3000 pts problem"""
import sys, json
sys.path.insert(0, ".")
import bench_detail as bench  # noqa: E402
from alvaar_amd import capi
ctx = capi.Context(0)
r, pb = bench.bench_ba(ctx, reps=10)
print({k: v for k, v in r.items() if k != "note"})
peaks = (49.3, 40.7)
rf = bench.roofline_ba(ctx, pb, peaks)
print(rf["kernel_us_per_solve"], rf["largest"])
import os, ctypes as C, numpy as np
if os.environ.get("ALVA_KSTAMPS"):
    lib = capi.lib if hasattr(capi, "lib") else capi._load()
    lib.alva_debug_kstamps.argtypes = [C.c_void_p]
    buf = np.zeros(4096, np.uint64)
    ctx.local_ba(pb, 5, 0.0)
    lib.alva_debug_kstamps(buf.ctypes.data)
    b = buf.astype(np.int64)[3072:3072 + 64] * 0.01
    print("k_solve (last launch), us: load %.2f  factor %.2f  forward %.2f  backward %.2f  store %.2f  total %.2f" % (
        b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4], b[5] - b[0]))
    for kb in range(7):
        s = b[8 + 4 * kb: 12 + 4 * kb]
        nxt = b[8 + 4 * (kb + 1)] if kb < 6 else b[2]
        print("  block %d: diagonal %.2f  panel %.2f  trailing %.2f" % (kb, s[1] - s[0], s[2] - s[1], nxt - s[2]))
    a = buf.astype(np.int64)[3072 + 48:3072 + 52] * 0.01
    print("k_assemble (last launch), us: pair sums' rows / columns %.2f  camera block + gradient %.2f  points' scalars + reductions %.2f" % (a[1] - a[0], a[2] - a[1], a[3] - a[2]))
