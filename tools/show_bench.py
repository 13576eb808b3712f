#!/usr/bin/env python3
"""Prints the headline numbers of bench.py JSON lines: python tools/show_bench.py a.json b.json"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{f}: {d['value']:.3e} {d['unit']}  ms/step {d['ms_per_step']:.3f}  job_frac_alg {r.get('job_frac_algorithmic', r.get('job_frac', 0)):.3f}  setup {d.get('setup_seconds', 0):.1f} s")
    print(f"   dominant: {r['kernel'][:40]} avg {r['kernel_avg_ms']:.4f} ms  achieved {r['achieved']:.0f} GB/s (frac {r['frac']:.3f})"
          f"  traffic {r.get('traffic')}")
    print("   per tick:", {k[:24]: round(v, 3) for k, v in r["per_slot_ms_per_tick"].items()})
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print(f"   cpu: {c['value']:.3e} (1 core), all cores {c['all_cores']['value']:.3e} on {c['all_cores']['cores']}")
