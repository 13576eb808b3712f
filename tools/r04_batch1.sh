#!/bin/bash
# round 4, first measurement batch on the GPU box: the new bench line (N = 1), the N = 2 / 4 launches with the schedule A/B block (one GPU:
# peer transport), the W = 8 loopback shares per schedule, the group host models
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -m gpu -q -p no:cacheprovider > gpurun_out/r04b_bench_contract_tests.txt 2>&1; echo "contract rc=$?"
tail -3 gpurun_out/r04b_bench_contract_tests.txt
timeout -k 10 300 python bench.py > gpurun_out/r04b_bench_n256.json 2> gpurun_out/r04b_bench_n256.err; echo "bench rc=$?"
for N in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N --transport peer --steps 10 --warmup 3 > gpurun_out/r04b_bench_${N}proc_one_gpu_ab.json 2> gpurun_out/r04b_bench_${N}proc_one_gpu_ab.err; echo "N=$N rc=$?"
done
timeout -k 10 500 python tools/lb_w8_timing.py 100 8 > gpurun_out/r04b_loopback_w8_schedules.txt 2>&1; echo "lb rc=$?"
timeout -k 10 500 python tools/group_host_models.py 256 8 30 > gpurun_out/r04b_group_host_models.txt 2> gpurun_out/r04b_group_host_models.err; echo "group rc=$?"
tail -5 gpurun_out/r04b_group_host_models.txt
