#!/bin/bash
# The tet surrogate at BUNNY_VERTS vertices (default 1 000 000) on the GPU box: kernel trace + planner statistics + SQ counter passes,
# the mesh generated once and kept in /tmp between the passes. usage: bash tools/bunny_large_profile.sh <tag>
TAG=${1:?tag}
R=$GRAFT_REPO_ROOT
export BUNNY_VERTS=${BUNNY_VERTS:-1000000} BUNNY_CACHE=/tmp/bunny_${BUNNY_VERTS:-1000000}.pkl
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bunny_run.py > $R/gpurun_out/${TAG}_bunny_large_plain.out 2>&1 || exit 1          # generates and caches the mesh
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_bunny_large -o kt --output-format csv -- python3 $R/tools/bunny_run.py > /dev/null 2> $R/gpurun_out/prof_${TAG}_bunny_large.err || exit 1
run() { local name=$1; shift
  rocprofv3 --pmc "$@" -d $R/gpurun_out/pmcbl_${name}_${TAG} -o pmc --output-format csv -- python3 $R/tools/bunny_run.py > /dev/null 2> $R/gpurun_out/pmcbl_${name}_${TAG}.err || exit 1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS
cd $R
python tools/bunny_summary.py gpurun_out/prof_${TAG}_bunny_large $TAG > gpurun_out/${TAG}_bunny_large_summary.md || exit 1
python tools/pmc_table.py gpurun_out/pmcbl_sq1_${TAG} gpurun_out/pmcbl_sq2_${TAG} gpurun_out/pmcbl_sq3_${TAG} > gpurun_out/${TAG}_bunny_large_sq_counters.txt || exit 1
cat gpurun_out/${TAG}_bunny_large_summary.md
