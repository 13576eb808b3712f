#!/bin/bash
# SQ / LDS counter passes for the 256^3 bench (one group per pass). usage on the GPU box: bash tools/pmc_sq.sh <tag> [extra bench.py arguments, e.g. --heterogeneous]
TAG=${1:-x}
shift
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_${name}_${TAG} -o pmc --output-format csv -- python3 $R/bench.py $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-parity --allow-stale-traffic > $R/gpurun_out/pmc_${name}_${TAG}.out 2> $R/gpurun_out/pmc_${name}_${TAG}.err || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run sq4 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU
