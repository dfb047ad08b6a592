"""Summarise rocprofv3 counter_collection CSVs: mean counter value per dispatch for kernels matching a substring."""
import csv
import sys
from collections import defaultdict

pat = sys.argv[1]
for path in sys.argv[2:]:
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
