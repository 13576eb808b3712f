#!/bin/bash
# PMC passes for the 256^3 bench, one counter group per pass (MI355X_MICROARCH.md: never mix with traces).
# usage (on the GPU box): bash tools/pmc_run.sh <tag> ; results under gpurun_out/pmc_<group>_<tag>/
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_${name}_${TAG} -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --allow-stale-traffic > $R/gpurun_out/pmc_${name}_${TAG}.out 2> $R/gpurun_out/pmc_${name}_${TAG}.err || return 1
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum && run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum && run hit TCC_HIT_sum TCC_MISS_sum
