"""Fold two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB per dispatch.  gfx950 corrections (MI355X_MICROARCH.md, "HBM"):
FETCH_SIZE counts a wide coalesced streaming read (16 B / lane) at HALF its bytes -> x2 for kernels whose loads are
16-B vectors (WIDE below, calibrated on k_level0<true>: 1 228 800 B of RGBA read with 16-B loads shows as ~628 KB);
narrow 1-4 B gathers and byte loads read 1:1 (calibrated on k_copy_level0: copies 307 200 B, shows 312 KB).
WRITE_SIZE is 1:1 (k_level0 writes 327 KB padded gray + 307 KB gray: shows 658 KB).
"""
import csv
import json
import re
import sys
from collections import defaultdict

WIDE = ("k_level0", "k_level0_batch", "k_bf_partial")   # kernels whose global reads are 16-B vectors per lane


def fold(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = name.split("(")[0].strip()
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


def main():
    fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
    import os
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes) over: " + os.environ.get("PMC_COMMAND", "bench.py's frame loop") + "; MI355X",
           "corrections": __doc__.split("\n\n", 2)[2].strip(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        wide = k.split("<")[0] in WIDE
        fb = f[1] * 1024 * (2 if wide else 1)
        wb = w[1] * 1024
        out["kernels"][k] = {"launches": max(f[0], w[0]), "FETCH_SIZE_KB_raw": round(f[1], 1), "WRITE_SIZE_KB_raw": round(w[1], 1),
                             "fetch_correction": 2 if wide else 1, "hbm_bytes_per_launch": int(fb + wb)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:25]:
        print(f"{k:32s} launches={v['launches']:5d} fetch_raw={v['FETCH_SIZE_KB_raw']:9.1f} KB write={v['WRITE_SIZE_KB_raw']:9.1f} KB -> {v['hbm_bytes_per_launch'] / 1e6:7.3f} MB/launch")


if __name__ == "__main__":
    main()
