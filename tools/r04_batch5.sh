#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_bench_contract.py tests/test_gpu_group.py -m gpu -q -p no:cacheprovider > gpurun_out/r04g_tests.txt 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r04g_tests.txt
for W in 2 8; do
  python bench.py --loopback-world $W --steps 40 --warmup 3 --no-cpu-baseline --no-parity --no-sustained > gpurun_out/r04g_bench_loopback_w${W}_auto.json 2> gpurun_out/r04g_bench_loopback_w${W}_auto.err; echo "lb W=$W rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/r04g_bench_loopback_w${W}_auto.json')); c = j['config']; print('W=$W', j['ms_per_step'], c['halo_schedule'], c.get('auto_calibration'), c.get('exchange'))"
done
