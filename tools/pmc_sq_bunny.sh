#!/bin/bash
# SQ counter passes for config 5 (100 k surrogate, tools/bunny_run.py). usage on the GPU box: bash tools/pmc_sq_bunny.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" -d $R/gpurun_out/pmcb_${name}_${TAG} -o pmc --output-format csv -- python3 $R/tools/bunny_run.py > $R/gpurun_out/pmcb_${name}_${TAG}.out 2> $R/gpurun_out/pmcb_${name}_${TAG}.err || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS
cd $R && python tools/pmc_table.py gpurun_out/pmcb_sq1_${TAG} gpurun_out/pmcb_sq2_${TAG} gpurun_out/pmcb_sq3_${TAG} > gpurun_out/${TAG}_bunny_sq_counters.txt
cat gpurun_out/${TAG}_bunny_sq_counters.txt
