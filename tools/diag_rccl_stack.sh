#!/bin/bash
# Runs tools/diag_rccl_stack.py configurations one after the other, each in its own process with the backtrace helper preloaded.
# Output: gpurun_out/$1_*.txt
tag=${1:-r03a}
mkdir -p gpurun_out
gcc -O1 -g -shared -fPIC -o tools/probe/libsegv_bt.so tools/probe/segv_bt.c -ldl || exit 1
run() {
    name=$1; shift
    echo "== $name: $*" | tee gpurun_out/${tag}_$name.txt
    LD_PRELOAD=$PWD/tools/probe/libsegv_bt.so timeout -k 10 240 python3 tools/diag_rccl_stack.py "$@" >> gpurun_out/${tag}_$name.txt 2>&1
    echo "rc=$?" | tee -a gpurun_out/${tag}_$name.txt
    tail -n 6 gpurun_out/${tag}_$name.txt
}
run plain_serial_graph --graph
run torch_serial_graph --torch-first --graph
run torch_overlap_graph --torch-first --overlap --graph
run plain_overlap_graph --overlap --graph
run plain_4solvers_graph --graph --solvers 4
run torch_4solvers_graph --torch-first --graph --solvers 4
run plain_3together_graph --graph --solvers 3 --together
exit 0
