#!/bin/bash
# HBM bytes per launch of a rank's T0 and T1 tile kernels (rank 0 of 256^3 / W, RCCL self-exchange): separate FETCH_SIZE / WRITE_SIZE passes
W=${1:-8}; TAG=${2:-r04t}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${TAG}_lb${W}_$C -o pmc --output-format csv -- python3 $R/bench.py --loopback-world $W --schedule serial-eager --steps 4 --warmup 1 --no-parity --no-cpu-baseline --no-gpu-state > /dev/null 2> $R/gpurun_out/pmc_${TAG}_lb${W}_$C.err || exit 1
done
cd $R
python3 - <<PY
import csv, collections, glob
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_${TAG}_lb${W}_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                d[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    acc[C] = d
print("| kernel | grid | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes = 2 F 1024 + W 1024 |")
for k in sorted(acc["FETCH_SIZE"], key=lambda k: -sum(acc["FETCH_SIZE"][k])):
    f = acc["FETCH_SIZE"][k]; w = acc["WRITE_SIZE"].get(k, [0])
    fa, wa = sum(f) / len(f), sum(w) / max(len(w), 1)
    print(f"| {k[0]} | {k[1]} | {len(f)} | {fa:.1f} | {wa:.1f} | {(2 * fa + wa) * 1024 / 1e6:.1f} MB |")
PY
