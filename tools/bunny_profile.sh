#!/bin/bash
# Config 5 evidence on the GPU box: rocprofv3 kernel trace of the 100k surrogate + planner statistics (groups per tile, lanes busy).
# usage: bash tools/bunny_profile.sh <tag>
TAG=${1:?tag}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_bunny -o kt --output-format csv -- python3 $R/tools/bunny_run.py > $R/gpurun_out/prof_${TAG}_bunny.out 2> $R/gpurun_out/prof_${TAG}_bunny.err || exit 1
cd $R
python tools/bunny_summary.py gpurun_out/prof_${TAG}_bunny $TAG > gpurun_out/${TAG}_bunny100k_summary.md || exit 1
cat gpurun_out/${TAG}_bunny100k_summary.md
