#!/bin/bash
# round 4, second batch: the whole GPU suite on the build with the 256-lane packed slots, then their A/B: 128^3 / 160^3 on one GPU and the
# W = 8 loopback share (a rank's 4 096 tiles of 256^3), packed against SB_NO_WIDE_SLOTS=1, interleaved
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04c_gpu_tests.txt 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r04c_gpu_tests.txt
OUT=gpurun_out/r04c_wide_slots_ab.txt; : > $OUT
for round in 1 2 3; do
  for n in 128 160; do
    for v in packed unpacked; do
      if [ $v = unpacked ]; then export SB_NO_WIDE_SLOTS=1; else unset SB_NO_WIDE_SLOTS; fi
      python bench.py --n $n --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-sustained --no-gpu-state 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('round $round n=$n $v: %.4f ms/tick, dominant launch %.2f us, model bytes/launch %.1f MB, lane-packed tiles %s' % (j['ms_per_step'], 1e3 * j['roofline']['kernel_avg_ms'], j['roofline']['model_bytes_per_launch'] / 1e6, j['plan']['lane_packed_tiles']))" >> $OUT
    done
  done
done
unset SB_NO_WIDE_SLOTS
for round in 1 2; do
  for v in packed unpacked; do
    if [ $v = unpacked ]; then export SB_NO_WIDE_SLOTS=1; else unset SB_NO_WIDE_SLOTS; fi
    echo "-- round $round, W = 8 loopback share, $v" >> $OUT
    timeout -k 10 300 python tools/lb_w8_timing.py 100 8 serial >> $OUT 2>&1
  done
done
unset SB_NO_WIDE_SLOTS
cat $OUT
