#!/bin/bash
# VERDICT r3 item 7, ONE experiment: does the run-to-run mode of the headline launch (profiles/r03s2u_lane_pack_more_boxes.txt: 2.84 / 3.10 ms
# from process to process on one box) follow the RELATIVE placement of the two arrays every launch streams side by side (positions and
# previous positions, 201 MB each)? The previous-position array is shifted inside its allocation (sb_tuning.prev_offset_bytes), fresh
# process per run, settings interleaved; the allocation addresses are printed beside the times.
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/${1:-r04d}_placement_modes.txt; : > $OUT
for round in 1 2 3 4 5; do
  for off in 0 256 1024 4096 69632 1048576; do
    SB_PREV_OFFSET=$off SB_PRINT_ALLOC=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-sustained --no-gpu-state 2> gpurun_out/.alloc.err | python -c "
import sys, json, re
j = json.loads(sys.stdin.read())
a = open('gpurun_out/.alloc.err').read()
m = re.search(r'pos3 (0x[0-9a-f]+) prev (0x[0-9a-f]+) vel (0x[0-9a-f]+)', a)
pos, prev = (int(m.group(1), 16), int(m.group(2), 16)) if m else (0, 0)
print('round $round prev_offset %8d: %.4f ms/tick, launch %.2f us | pos3 %#x prev %#x  (prev - pos3) mod 4 KiB = %5d, mod 64 KiB = %6d, mod 2 MiB = %8d' % ($off, j['ms_per_step'], 1e3 * j['roofline']['kernel_avg_ms'], pos, prev, (prev - pos) % 4096, (prev - pos) % 65536, (prev - pos) % (2 << 20)))" >> $OUT
  done
done
cat $OUT
