#!/bin/bash
# Workgroup quantisation (profiles/r04t_loopback_w8_pmc.md: a rank's T1 launch of 4 481 workgroups takes + 45 % for + 10 % bytes: 2 048 four-wave
# workgroups fit the chip at once, 4 481 is two rounds and a fifth): per-launch times of the T0 and T1 kernels for 256-lane (default in this
# range) against 128-lane workgroups (3 072 at once, each slower) over cube sizes whose tile counts fall on either side of the steps.
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/${1:-r04u}_width_quantisation_ab.txt; : > $OUT
for n in 96 104 112 120 128 136 144 152 160 176; do
  for v in wide narrow; do
    if [ $v = narrow ]; then export SB_NARROW_MIN_TILES=1; else unset SB_NARROW_MIN_TILES; fi
    python bench.py --n $n --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-sustained --no-gpu-state 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; p = j['plan']
slots = r['per_slot_ms_per_tick']; cnt = r['per_slot_launches_per_tick']
k0 = [k for k in slots if 'on T0' in k][0]; k1 = [k for k in slots if 'on T1' in k][0]
print('n=%3d %-6s tiles T0 %5d T1 %5d | %.4f ms/tick | event pairs: T0 launch %.2f us, T1 launch %.2f us | packed %s' % ($n, '$v', p['n_tiles'][0], p['n_tiles'][1], j['ms_per_step'], 1e3 * slots[k0] / max(cnt[k0], 1), 1e3 * slots[k1] / max(cnt[k1], 1), p['lane_packed_tiles']))" >> $OUT
  done
done
unset SB_NARROW_MIN_TILES
cat $OUT
