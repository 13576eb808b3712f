#!/bin/bash
# Does lane packing move the launch size from which 128-lane workgroups pay (SB_NARROW_MIN_TILES, default 10 240 tiles)? Cube sizes around the
# threshold, default against narrow + packed forced, interleaved. Result (profiles/r03s2_narrow_threshold_ab.txt): no -- below 10 240 tiles the
# 256-lane launch stays ahead (160^3: 0.736 against 0.764 ms per tick).
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n=$1 %-14s %.4f ms/tick  packed %s tiles %s' % ('$2', d['ms_per_step'], d['plan']['lane_packed_tiles'], d['plan']['n_tiles']))"; }
for n in 96 128 144 160 176; do for r in 1 2; do
  python bench.py --cube-edge $n --no-cpu-baseline --no-parity --steps 100 --warmup 10 2>/dev/null | show $n default
  SB_NARROW_MIN_TILES=1 python bench.py --cube-edge $n --no-cpu-baseline --no-parity --steps 100 --warmup 10 2>/dev/null | show $n narrow+packed
done; done
