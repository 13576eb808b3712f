#!/bin/bash
# Heterogeneous jelly cube (per-spring rest lengths: 8-byte slots): register-resident programs for full slots (product) against the
# round-2 condition (variant regc = -DSB_REG_COMPACT_ONLY: dictionary-coded tiles only), interleaved, 256^3 and 64^3
# (build the variant first: make -C softbodyunity_amd/csrc VARIANT=regc EXTRA=-DSB_REG_COMPACT_ONLY -j8). Round 4: WITH the parity legs -- every run's final
# state is checked on the golden checksum (the host now reads the kernel build's own constants when it decides which tiles to
# lane-pack, so a variant whose kernels cannot decode the full-slot form is not handed it)
for round in 1 2; do
  for v in "" regc; do
    name=${v:-product}
    for n in 256 64; do
      steps=30; [ $n = 64 ] && steps=35
      SB_LIB_VARIANT=$v python bench.py --n $n --heterogeneous --steps $steps --warmup 5 --no-cpu-baseline --no-sustained > gpurun_out/hetab_${name}_n${n}_r${round}.json 2> gpurun_out/hetab_${name}_n${n}_r${round}.err
      python -c "
import json,sys; j=json.load(open('gpurun_out/hetab_${name}_n${n}_r${round}.json')); p=j['config']['parity']; print('n=$n %-8s round $round  %.4f ms/tick  kernel %.2f us  model %.1f MB  lane-packed tiles %s  golden bitwise %s  small leg bitwise %s' % ('$name', j['ms_per_step'], 1e3*j['roofline']['kernel_avg_ms'], j['roofline']['model_bytes_per_launch']/1e6, j['plan']['lane_packed_tiles'], p['golden']['bitwise'], p['small']['bitwise']))"
    done
  done
done
