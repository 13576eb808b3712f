#!/bin/bash
# kernel trace of rank 0's share of 256^3 / W with an RCCL self-exchange, serialised eager and overlapped eager schedules
W=${1:-8}; TAG=${2:-r03t}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for S in serial-eager overlap-eager; do
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${TAG}_lb${W}_$S -o kt --output-format csv -- python3 $R/bench.py --loopback-world $W --schedule $S --steps 20 --warmup 3 --no-parity --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_lb${W}_$S.out 2> $R/gpurun_out/prof_${TAG}_lb${W}_$S.err || exit 1
  python3 $R/tools/loopback_timeline.py $R/gpurun_out/prof_${TAG}_lb${W}_$S 28 "rank 0 of 256^3 / $W, RCCL self-exchange, schedule $S ($TAG)" > $R/gpurun_out/${TAG}_loopback_w${W}_${S}_timeline.md
  cat $R/gpurun_out/${TAG}_loopback_w${W}_${S}_timeline.md
done
