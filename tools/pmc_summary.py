"""Per-kernel mean of one rocprofv3 PMC counter from a counter_collection.csv."""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
print(f"{'kernel':72s} {'launches':>8s} {'mean ' + counter:>20s}")
for name, vals in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:72]:72s} {len(vals):8d} {sum(vals) / len(vals):20.1f}")
