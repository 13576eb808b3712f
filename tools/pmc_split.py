#!/usr/bin/env python3
"""Per-dispatch PMC means split by (kernel, grid size) -- separates the T0 and T1 launches of tile_kernel.
usage: python tools/pmc_split.py dir1 dir2 ..."""
import collections
import csv
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in os.listdir(d):
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                if "tile_kernel" not in r["Kernel_Name"] and "global_" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"][:44], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in sorted(acc):
    print(f"{k[0]}  workgroups={k[1]}  mean_us={sum(dur[k]) / len(dur[k]):.1f}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
