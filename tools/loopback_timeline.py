#!/usr/bin/env python3
"""Per-kernel anatomy of one rank's share with a self-exchange: summarises a `rocprofv3 --kernel-trace` of
`bench.py --loopback-world W --schedule S --no-parity --no-cpu-baseline` into a table (kernel, launches per tick, avg us, us per tick).
usage: python tools/loopback_timeline.py <rocprof dir> <ticks traced> [title]"""
import collections
import csv
import os
import sys

d, ticks = sys.argv[1], int(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else d
acc = collections.defaultdict(list)
for root, _, files in os.walk(d):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            for r in csv.DictReader(open(os.path.join(root, f))):
                name = r["Kernel_Name"]
                short = name.split("(")[0][:70]
                acc[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"# {title}\n\n| kernel | launches | per tick | avg µs | µs per tick |\n|---|---|---|---|---|")
tot = 0
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    v = acc[k]
    tot += sum(v)
    print(f"| `{k}` | {len(v)} | {len(v) / ticks:.1f} | {sum(v) / len(v) / 1e3:.1f} | {sum(v) / ticks / 1e3:.1f} |")
print(f"\nsum of kernel time: {tot / ticks / 1e3:.1f} µs per tick over {ticks} ticks (warm-up, timed, the two profiled eager ticks and the three exchange-timing ticks)")
