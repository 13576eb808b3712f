#!/bin/bash
# Does sampling the card's clocks and power (bench.py config.gpu_state, sysfs reads every 10 ms) perturb the timed region? Interleaved A/B, one box.
for r in 1 2 3; do
  for f in "" "--no-gpu-state"; do
    python bench.py --no-cpu-baseline --no-parity --steps 40 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-16s %.4f ms/tick  kernel %.2f us' % ('$f' or 'sampling', d['ms_per_step'], 1e3*d['roofline']['kernel_avg_ms']))"
  done
done
