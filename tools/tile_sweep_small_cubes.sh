#!/bin/bash
# 64^3: ms per tick against --tile (run on the GPU box from the repository root)
for t in 512 384 256 192 128; do
  python bench.py --n 64 --tile $t --steps 400 --warmup 40 --no-cpu-baseline --no-parity --allow-stale-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('tile $t: %.4f ms/tick'%d['ms_per_step'])"
done
for t in 512 256 128; do
  python bench.py --n 96 --tile $t --steps 200 --warmup 20 --no-cpu-baseline --no-parity --allow-stale-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('n96 tile $t: %.4f ms/tick'%d['ms_per_step'])"
done
