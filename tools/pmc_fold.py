"""Fold a rocprofv3 counter_collection.csv into per-kernel sums: python tools/pmc_fold.py <csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / len(n[k]):16.0f} per dispatch")
