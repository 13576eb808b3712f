#!/bin/bash
# Config 5 kernel trace with the steps of every launch removed (the fixed cost per launch). Build the variant first (in the container):
#   make -C softbodyunity_amd/csrc VARIANT=abl1 EXTRA=-DSB_ABLATE=1
# usage on the GPU box: bash tools/bunny_ablate.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SB_LIB_VARIANT=abl1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_abl1_bunny -o kt --output-format csv -- python3 $R/tools/bunny_run.py > $R/gpurun_out/prof_abl1_bunny.out 2> $R/gpurun_out/prof_abl1_bunny.err || exit 1
cd $R
python tools/bunny_summary.py gpurun_out/prof_abl1_bunny abl1 | head -16
