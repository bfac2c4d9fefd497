"""All counters of a rocprofv3 counter_collection.csv as one table: kernels (matching a pattern) x counters."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        if pat in row["Kernel_Name"]:
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
for c in counters:
    print(f"{c:28s}" + "".join(f"{(sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else float('nan')):>18.0f}" for k in sorted(acc)))
print(f"{'':28s}" + "".join(f"{k.split('(')[0][-17:]:>18s}" for k in sorted(acc)))
