#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 500 python -m pytest tests/test_gpu_lane_pack.py tests/test_bench_contract.py -m gpu -q -p no:cacheprovider > gpurun_out/r04e_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04e_tests.txt
timeout -k 10 600 bash tools/profile_round.sh r04e > gpurun_out/r04e_profile_round.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r04e_profile_round.log
timeout -k 10 400 bash tools/placement_modes.sh r04e > /dev/null 2>&1; echo "placement rc=$?"; cat gpurun_out/r04e_placement_modes.txt
