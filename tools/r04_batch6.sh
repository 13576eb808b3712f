#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_peer_transport.py -m gpu -q -p no:cacheprovider > gpurun_out/r04h_tests.txt 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r04h_tests.txt
timeout -k 10 500 python tools/lb_w8_timing.py 100 8 rccl > gpurun_out/r04h_loopback_w8_schedules.txt 2>&1; cat gpurun_out/r04h_loopback_w8_schedules.txt
