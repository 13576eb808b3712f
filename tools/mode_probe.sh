#!/bin/bash
# Run-to-run timing modes of the heterogeneous 256^3 workload: 6 processes, allocation addresses beside the time
for i in 1 2 3 4 5 6; do
  SB_PRINT_ALLOC=1 python bench.py --heterogeneous --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/mode_het_$i.json 2> gpurun_out/mode_het_$i.err
  python -c "
import json; j=json.load(open('gpurun_out/mode_het_$i.json')); print('het run $i: %.3f ms/tick, kernel %.1f us' % (j['ms_per_step'], 1e3*j['roofline']['kernel_avg_ms']))"
  grep "\[alloc\]" gpurun_out/mode_het_$i.err | head -1
done
for i in 1 2 3; do
  SB_PRINT_ALLOC=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/mode_hom_$i.json 2> gpurun_out/mode_hom_$i.err
  python -c "
import json; j=json.load(open('gpurun_out/mode_hom_$i.json')); print('headline run $i: %.3f ms/tick, kernel %.1f us' % (j['ms_per_step'], 1e3*j['roofline']['kernel_avg_ms']))"
  grep "\[alloc\]" gpurun_out/mode_hom_$i.err | head -1
done
