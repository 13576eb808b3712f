#!/bin/bash
# Interleaved A/B of plugin variants on ONE box at two sizes: tools/ab2.sh v1 v2 ...   ("product" = the product build)
# Prints ms/tick of the 256^3 (60 ticks) and 64^3 (600 ticks) cubes per variant and round. EXTRA_ARGS adds bench flags.
ROUNDS=${ROUNDS:-2}
VARIANTS=("$@")
for round in $(seq 1 $ROUNDS); do
  for name in "${VARIANTS[@]}"; do
    v=$name; [ "$v" = "product" ] && v=""
    for cfg in "256 60 5" "64 600 50"; do
      read -r n steps warm <<< "$cfg"
      out=gpurun_out/ab_${name}_n${n}_r${round}
      SB_LIB_VARIANT=$v python bench.py --n $n --steps $steps --warmup $warm --no-cpu-baseline --no-parity --allow-stale-traffic $EXTRA_ARGS \
          > $out.json 2> $out.err || { echo "$name n=$n FAILED"; tail -3 $out.err; continue; }
      python -c "
import json
d = json.load(open('$out.json'))
print('round $round %-10s n=%-4s %.4f ms/tick  dominant %.2f us  setup %.1f s' % ('$name', '$n', d['ms_per_step'], 1e3 * d['roofline']['kernel_avg_ms'], d['setup_seconds']))"
    done
  done
done
