#!/bin/bash
# Long multi-process runs on the ONE GPU of the box (peer-store transport, sharded authoring) against the single-process run of the
# same tick count: the order-independent state checksum must be identical (bench.py config.parity.golden.checksum).
# usage on the GPU box: bash tools/multiproc_soak.sh
R=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
OUT=$R/gpurun_out/multiproc_soak.txt
: > $OUT
run() { # edge ticks procs extra
  local n=$1 t=$2 p=$3; shift 3
  if [ $p -eq 1 ]; then
    timeout -k 10 500 python $R/bench.py --n $n --steps $t --warmup 2 --no-cpu-baseline --no-sustained "$@" 2>/dev/null
  else
    timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $p --master-addr 127.0.0.1 --master-port $((29650 + p)) \
      $R/bench.py --gpus $p --transport peer --no-ab --no-sustained --cube-edge $n --steps $t --warmup 2 "$@" 2>/dev/null
  fi | python -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('n=$n ticks=%d procs=$p $*: %.4f ms/tick checksum %s finite %s' % (j['steps']+j['warmup'], j['ms_per_step'], j['config']['parity']['golden']['checksum'], j['config']['finite']))" >> $OUT || exit 1
}
run 64 3000 1 && run 64 3000 2 && run 64 3000 4 && run 64 3000 4 --heterogeneous && run 64 3000 1 --heterogeneous && run 256 400 1 && run 256 400 4 || exit 1
cat $OUT
