import sys; sys.path.insert(0, ".")
from alvaar_amd import capi, synth
ctx = capi.Context(0)
pb = synth.make_ba_problem(20, 3000, 42)
for _ in range(3): ctx.local_ba(pb, 5, 0.0)
kt = capi.kernel_times(lambda: ctx.local_ba(pb, 5, 0.0), 10)
tot = 0
for k, (c, us) in sorted(kt.items(), key=lambda kv: -kv[1][0]*kv[1][1]):
    print(f"{k:28s} calls/solve {c/10:5.1f} avg {us:6.1f} us  total {c/10*us:6.1f}")
    tot += c/10*us
print("sum", round(tot,1))
