#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04i_gpu_tests.txt 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r04i_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04i_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04i_smoke.txt
for N in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N --transport peer --steps 10 --warmup 3 > gpurun_out/r04i_bench_${N}proc_one_gpu_ab.json 2> gpurun_out/r04i_bench_${N}proc_one_gpu_ab.err; echo "N=$N rc=$?"
done
timeout -k 10 400 python tools/group_host_models.py 256 8 30 > gpurun_out/r04i_group_host_models.txt 2> gpurun_out/r04i_group_host_models.err; echo "group rc=$?"; grep host_model gpurun_out/r04i_group_host_models.txt | cut -c1-260
