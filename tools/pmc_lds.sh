#!/bin/bash
# LDS bank-conflict counters of the 256^3 bench without ("0 0 0") and with ("1 8 0") the planner's bank-aware lane order.
# usage on the GPU box: bash tools/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- "0 0 0" "1 8 0"
for m in "$@"; do
  read -r a b c <<< "$m"
  if [ "$a" = "0" ]; then export SB_NO_BANK_ORDER=1; else unset SB_NO_BANK_ORDER; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/pmc_lds_${a}_${b}_${c} -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --allow-stale-traffic > /dev/null 2> $R/gpurun_out/pmc_lds_${a}_${b}_${c}.err || echo fail
  echo "mode $a mod $b key $c"; python3 $R/tools/pmc_split.py $R/gpurun_out/pmc_lds_${a}_${b}_${c} | grep -A3 "tile_kernel<1" | grep -v INSTS
done
