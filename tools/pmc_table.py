#!/usr/bin/env python3
"""Mean per-dispatch PMC values per kernel from rocprofv3 --pmc CSV dirs: python tools/pmc_table.py dir1 dir2 ..."""
import collections
import csv
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in os.listdir(d):
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "tile_kernel" not in k and "global_" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
