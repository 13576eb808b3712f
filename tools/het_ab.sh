#!/bin/bash
# Heterogeneous jelly cube (per-spring rest lengths: 8-byte slots): register-resident programs for full slots (product) against the
# round-2 condition (variant regc = -DSB_REG_COMPACT_ONLY: dictionary-coded tiles only), interleaved, 256^3 and 64^3
for round in 1 2 3; do
  for v in "" regc; do
    name=${v:-product}
    for n in 256 64; do
      steps=40; [ $n = 64 ] && steps=400
      SB_LIB_VARIANT=$v python bench.py --n $n --heterogeneous --steps $steps --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/hetab_${name}_n${n}_r${round}.json 2> gpurun_out/hetab_${name}_n${n}_r${round}.err
      python -c "
import json,sys; j=json.load(open('gpurun_out/hetab_${name}_n${n}_r${round}.json')); print('n=$n %-8s round $round  %.4f ms/tick  kernel %.2f us  model %.1f MB' % ('$name', j['ms_per_step'], 1e3*j['roofline']['kernel_avg_ms'], j['roofline']['model_bytes_per_launch']/1e6))"
    done
  done
done
