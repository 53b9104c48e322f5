"""Fold the three counter passes of tools/pmc_klt.sh into one stamped summary (see there).  gfx950 corrections as in tools/pmc_summary.py
(FETCH_SIZE x2 only for kernels whose global reads are 16-B vectors; the tracker gathers 1-4 B: 1:1)."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["alvaar_amd/csrc/klt.hip", "alvaar_amd/csrc/stages_hip.hip", "alvaar_amd/csrc/track_slots.hpp"]
WIDE = ("k_level0",)


def fold(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0].strip().replace("alva_slam::", "")
            acc[name][0] += 1
            acc[name][1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


def main():
    fdir, wdir, tdir, out_path = sys.argv[1:5]
    fetch, write = fold(fdir, "FETCH_SIZE"), fold(wdir, "WRITE_SIZE")
    hit, miss = fold(tdir, "TCC_HIT_sum"), fold(tdir, "TCC_MISS_sum")
    out = {"source": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (three separate passes) over "
                     "python tools/system_sustained.py (FRAMES=%s); MI355X" % os.environ.get("FRAMES", "?"),
           "commit": os.environ.get("ALVA_COMMIT", "unknown"),
           "source_sha16": {s: hashlib.sha256(open(os.path.join(ROOT, s), "rb").read()).hexdigest()[:16] for s in SOURCES},
           "kernels": {}}
    for k in sorted(set(fetch) | set(write) | set(hit)):
        if not k.startswith("k_track") and not k.startswith("k_p3p") and not k.startswith("k_pnp") and not k.startswith("k_level0") and not k.startswith("k_pyr"):
            continue
        f, w = fetch.get(k, (0, 0.0)), write.get(k, (0, 0.0))
        h, m = hit.get(k, (0, 0.0)), miss.get(k, (0, 0.0))
        wide = k.split("<")[0] in WIDE
        e = {"launches": max(f[0], w[0], h[0]), "FETCH_SIZE_KB_raw": round(f[1], 1), "WRITE_SIZE_KB_raw": round(w[1], 1), "fetch_correction": 2 if wide else 1,
             "hbm_bytes_per_launch": int(f[1] * 1024 * (2 if wide else 1) + w[1] * 1024),
             "TCC_HIT_per_launch": round(h[1], 1), "TCC_MISS_per_launch": round(m[1], 1),
             "l2_hit_rate": (h[1] / (h[1] + m[1])) if (h[1] + m[1]) > 0 else None}
        out["kernels"][k] = e
        print(f"{k:28s} launches={e['launches']:5d} hbm={e['hbm_bytes_per_launch'] / 1e6:7.3f} MB/launch  L2 hit rate={e['l2_hit_rate']}")
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
