#!/bin/bash
# What draws the package to its power cap? tools/clock_probe.py (ms per tick, shader clock, package power under load) on the product
# and on the timing-only ablation builds of the tile kernel: abl2 = rounds with their LDS traffic and barriers but no arithmetic
# (WRONG results), abl1 = no rounds at all (loads, MARK step, stores). Build first:
#   make -C softbodyunity_amd/csrc VARIANT=abl1 EXTRA=-DSB_ABLATE=1 ; make -C softbodyunity_amd/csrc VARIANT=abl2 EXTRA=-DSB_ABLATE=2
# usage on the GPU box: bash tools/power_ablation.sh <tag>
TAG=${1:?tag}
for v in "" abl2 abl1 ""; do
  SB_LIB_VARIANT=$v timeout -k 10 200 python tools/clock_probe.py > gpurun_out/${TAG}_power_${v:-product}.json 2>> gpurun_out/${TAG}_power.err || exit 1
  python - "$v" gpurun_out/${TAG}_power_${v:-product}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
own = d.get("own_card")
r = d["under_load_min_max"][own] if own else max(d["under_load_min_max"].values(), key=lambda v: v.get("power1_input", [0, 0])[1])
print("%-8s %.3f ms/tick  sclk %s MHz  power %s W (cap %s)  samples %d" % (sys.argv[1] or "product", min(d["ms_per_tick_5x60"]), r.get("sclk_MHz"), r.get("power1_input"), r.get("power1_cap", [0, 0])[1], r["samples"]))
PY
done
