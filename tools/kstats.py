"""Print the top kernels of a rocprofv3 *_kernel_stats.csv."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={r['Percentage']}")
