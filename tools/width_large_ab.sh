#!/bin/bash
# The 128 / 256-lane threshold (10 240 tiles) was set in round 3 against 256-lane workgroups that staged their slots through LDS. With the
# 8-byte wide-packed slots of round 4 the two forms tie at 176^3 (10 648 tiles, profiles/r04u_width_quantisation_ab.txt): where do they part
# above it? 128-lane (default) against 256-lane (SB_NARROW_MIN_TILES above the tile count) at 192^3 .. 256^3, interleaved, two repeats.
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/${1:-r04w}_width_large_ab.txt; : > $OUT
for rep in 0 1; do
for n in 192 224 256; do
  for v in narrow wide; do
    if [ $v = wide ]; then export SB_NARROW_MIN_TILES=1000000; else unset SB_NARROW_MIN_TILES; fi
    python bench.py --n $n --steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-sustained --no-gpu-state 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; p = j['plan']
slots = r['per_slot_ms_per_tick']; cnt = r['per_slot_launches_per_tick']
k0 = [k for k in slots if 'on T0' in k][0]; k1 = [k for k in slots if 'on T1' in k][0]
print('rep $rep n=%3d %-6s tiles T0 %5d T1 %5d | %.4f ms/tick | event pairs: T0 launch %.2f us, T1 launch %.2f us | packed %s | golden %s' % ($n, '$v', p['n_tiles'][0], p['n_tiles'][1], j['ms_per_step'], 1e3 * slots[k0] / max(cnt[k0], 1), 1e3 * slots[k1] / max(cnt[k1], 1), p['lane_packed_tiles'], (j.get('parity') or {}).get('golden')))" >> $OUT
  done
done
done
unset SB_NARROW_MIN_TILES
cat $OUT
