#!/bin/bash
# Follow-up of profiles/r04e_placement_modes.txt (the slow mode of the headline launch does not follow the placement of the arrays; it clusters
# in TIME): the same fresh-process runs with the card's clocks and power sampled over warm-up + timed region (bench.py config.gpu_state),
# and a run's position in the sequence -- does a slow run sit at a lower shader clock / higher power?
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/${1:-r04f}_mode_clock_probe.txt; : > $OUT
for k in $(seq 1 ${2:-16}); do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-sustained 2>/dev/null | python -c "
import sys, json, time
j = json.loads(sys.stdin.read()); g = (j['config'].get('gpu_state') or {}).get('min_max', {})
print('run %2d t=%s: %.4f ms/tick, launch %.2f us | sclk %s MHz, power %s W (cap %s), samples %s' % ($k, time.strftime('%H:%M:%S'), j['ms_per_step'], 1e3 * j['roofline']['kernel_avg_ms'], g.get('sclk_MHz'), g.get('power_W'), (g.get('power_cap_W') or [None])[0], (j['config'].get('gpu_state') or {}).get('samples')))" >> $OUT
done
cat $OUT
