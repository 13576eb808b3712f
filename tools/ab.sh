#!/bin/bash
# A/B timing of plugin variants on ONE box, interleaved: tools/ab.sh "<bench args>" variant1 variant2 ... ("" = product build)
ARGS="$1"; shift
for round in 1 2; do
  for v in "$@"; do
    name=${v:-product}
    SB_LIB_VARIANT=$v python bench.py $ARGS --no-cpu-baseline --no-parity --allow-stale-traffic > gpurun_out/ab_${name}_r${round}.json 2> gpurun_out/ab_${name}_r${round}.err
    python tools/show_bench.py gpurun_out/ab_${name}_r${round}.json | head -3
  done
done
