#!/bin/bash
# Interleaved A/B of plugin variants over cube sizes: tools/size_ab.sh "96 128 160" product wt   -> ms per tick
SIZES=$1; shift
for n in $SIZES; do
  for round in 1 2; do
    for name in "$@"; do
      v=$name; [ "$v" = "product" ] && v=""
      SB_LIB_VARIANT=$v python bench.py --n $n --steps 100 --warmup 10 --no-cpu-baseline --no-parity --allow-stale-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('n=$n %-8s %.4f ms/tick  tiles %s' % ('$name', d['ms_per_step'], d['plan']['n_tiles'] if 'plan' in d else ''))"
    done
  done
done
