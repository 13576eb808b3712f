#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SB_LIB_VARIANT=abl1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_abl1_bunny -o kt --output-format csv -- python3 $R/tools/bunny_run.py > $R/gpurun_out/prof_abl1_bunny.out 2> $R/gpurun_out/prof_abl1_bunny.err || exit 1
cd $R
python tools/bunny_summary.py gpurun_out/prof_abl1_bunny abl1 | head -16
