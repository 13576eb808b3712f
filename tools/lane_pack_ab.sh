#!/bin/bash
# Lane-packed slots (16 B per lane instead of 4 B per slot in register-resident spring tiles of narrow launches) against SB_NO_LANE_PACK=1:
# interleaved bench runs on one box. usage: bash tools/lane_pack_ab.sh [rounds] [extra bench flags]
R=${1:-3}; shift
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.4f ms/tick  kernel %.2f us  model bytes %.1f MB  packed tiles %s' % ('$1', d['ms_per_step'], 1e3*d['roofline']['kernel_avg_ms'], d['roofline']['model_bytes_per_launch']/1e6, d['plan']['lane_packed_tiles']))"; }
for r in $(seq 1 $R); do
  python bench.py --no-cpu-baseline --no-parity --steps 60 "$@" 2>/dev/null | show packed
  SB_NO_LANE_PACK=1 python bench.py --no-cpu-baseline --no-parity --steps 60 "$@" 2>/dev/null | show unpacked
done
