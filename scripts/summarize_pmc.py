"""Summarise rocprofv3 --pmc CSVs: per-kernel mean of each counter (KB units for FETCH/WRITE_SIZE)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for sub in ("pmc_fetch", "pmc_write"):
    files = glob.glob(os.path.join(root, sub, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")[:90]
                ctr = row.get("Counter_Name", "?")
                try:
                    val = float(row.get("Counter_Value", "0"))
                except ValueError:
                    continue
                a = acc[(name, ctr)]
                a[0] += val
                a[1] += 1
    print(f"== {sub}: {len(files)} file(s)")
    for (name, ctr), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{ctr:12s} mean={tot / n:14.1f} n={n:6d}  {name}")
