#!/bin/bash
# bench.py's multi-rank path on a ONE-GPU box: N processes share the GPU (a box allows 6), ghosts travel through the peer-store
# transport (RCCL refuses several ranks on one device). Exercises everything of the N > 1 launch except RCCL and xGMI: the gloo control
# plane, per-rank planning of the split mesh, the golden-checksum leg summed over ranks, max-over-ranks timing.
# usage on the GPU box: bash tools/bench_multiproc_one_gpu.sh [N=4] [extra bench flags]
N=${1:-4}; shift
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
    bench.py --gpus $N --transport peer --steps 5 --warmup 2 "$@"
