#!/usr/bin/env python3
"""Turns rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

  python tools/prof_summary.py --round r01 --kt gpurun_out/prof_r01_n256 --fetch gpurun_out/pmc_fetch_n256 \
         --write gpurun_out/pmc_write_n256 --key n256_tile512_gpus1

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts exactly half of a wide coalesced
read stream, so bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024. The x2 was checked on this access pattern:
tile_kernel<1> must read 28 B/particle + 8 B/constraint.
"""
import argparse
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_mean(d, counter):
    """Mean counter value per (kernel, workgroups per launch): T0 and T1 launches of tile_kernel differ in grid size."""
    acc = collections.defaultdict(list)
    for f in os.listdir(d):
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                if r["Counter_Name"] == counter:
                    acc[(r["Kernel_Name"], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def trace_mean(d):
    """Mean duration (ns) per (kernel, workgroups per launch) from the kernel trace."""
    acc = collections.defaultdict(list)
    for f in os.listdir(d):
        if f.endswith("kernel_trace.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
                acc[(r["Kernel_Name"], wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--key", required=True, help="e.g. n256_tile512_gpus1")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats_src = [f for f in os.listdir(args.kt) if f.endswith("kernel_stats.csv")][0]
    dst = os.path.join(out_dir, f"{args.round}_{args.key}_kernel_stats.csv")
    shutil.copy(os.path.join(args.kt, stats_src), dst)
    rows = list(csv.DictReader(open(dst)))
    md = [f"# rocprofv3 --kernel-trace --stats, {args.key} ({args.round})", "",
          "| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
    for r in rows:
        md.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | "
                  f"{float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
    traffic = {}
    tm = trace_mean(args.kt)
    if tm:
        md += ["", "## per launch shape (kernel trace; T0 and T1 launches of one kernel differ in workgroup count)", "",
               "| kernel | workgroups | launches | avg µs |", "|---|---|---|---|"]
        for k in sorted(tm):
            if "tile_kernel" in k[0] or "global_" in k[0] or "halo_" in k[0]:
                md.append(f"| `{k[0][:60]}` | {k[1]} | {tm[k][1]} | {tm[k][0] / 1e3:.2f} |")
    if args.fetch and args.write:
        fe = pmc_mean(args.fetch, "FETCH_SIZE"); wr = pmc_mean(args.write, "WRITE_SIZE")
        md += ["", "## HBM traffic per launch (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)", "",
               "| kernel | workgroups | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | bytes = 2·F·1024 + W·1024 | avg µs (kernel trace) | TB/s |",
               "|---|---|---|---|---|---|---|"]
        mid = sorted(k for k in fe if "tile_kernel<1" in k[0] and k in wr)      # mid-tick kernel: fewer workgroups = T0
        for k in sorted(fe):
            if k not in wr:
                continue
            b = 2 * fe[k][0] * 1024 + wr[k][0] * 1024
            us = tm.get(k, (0, 0))[0] / 1e3
            tbs = b / (us * 1e-6) / 1e12 if us else float("nan")
            md.append(f"| `{k[0][:60]}` | {k[1]} | {fe[k][0]:.1f} | {wr[k][0]:.1f} | {b / 1e6:.1f} MB | {us:.2f} | {tbs:.2f} |")
            if k in mid:
                if k == mid[0]: traffic["0"] = b
                if k == mid[-1]: traffic["1"] = b
        tj_path = os.path.join(out_dir, "hbm_traffic.json")
        tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
        # which sources the counters were measured on: bench.py reports it beside the figure, and checks the figure against
        # the compulsory-bytes model of the build it runs (a stale entry fails the bench, it does not slip through)
        import hashlib
        h = hashlib.sha256()
        csrc = os.path.join(ROOT, "softbodyunity_amd", "csrc")
        for f in sorted(os.listdir(csrc)):      # every source of the plugin (round 4: solver.hip / kernels.hip.hpp became several units)
            if f.endswith((".hip", ".hpp", ".cpp")):
                h.update(open(os.path.join(csrc, f), "rb").read())
        traffic["meta"] = {"round": args.round, "csrc_sha16": h.hexdigest()[:16],
                           "counters": "FETCH_SIZE, WRITE_SIZE (separate rocprofv3 --pmc passes); bytes = 2*F*1024 + W*1024"}
        tj[args.key] = traffic
        json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
    open(os.path.join(out_dir, f"{args.round}_{args.key}_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
