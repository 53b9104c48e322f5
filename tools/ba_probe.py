"""Local BA phase times: ALVA_BA_TIMING=1 python tools/ba_probe.py"""
import os
import sys
os.environ["ALVA_BA_TIMING"] = "1"
sys.path.insert(0, ".")
import time  # noqa: E402
import alvaar_amd  # noqa: E402
from alvaar_amd import synth, capi  # noqa: E402
ctx = alvaar_amd.Context(0)
pb = synth.make_ba_problem(20, 3000, 42)
for _ in range(3):
    ctx.local_ba(pb, 5, 0.0)
t0 = time.perf_counter()
for _ in range(10):
    r = ctx.local_ba(pb, 5, 0.0)
print("ms per solve", (time.perf_counter() - t0) / 10 * 1e3, "iterations", int(r["info"][0]) - 1)
kt = capi.kernel_times(lambda: ctx.local_ba(pb, 5, 0.0), 5)
tot = 0
for k, (calls, us) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    print(f"{k:28s} launches/solve {calls / 5:5.1f}  avg {us:7.2f} us  total {calls / 5 * us:7.1f}")
    tot += calls / 5 * us
print("sum of kernel times per solve %.1f us" % tot)
