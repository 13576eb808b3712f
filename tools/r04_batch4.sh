#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests/test_gpu_lane_pack.py tests/test_abi.py tests/test_c_harness.py -q -p no:cacheprovider > gpurun_out/r04f_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04f_tests.txt
python bench.py > gpurun_out/r04f_bench_n256.json 2> gpurun_out/r04f_bench_n256.err; echo "bench rc=$?"
python bench.py --n 64 --steps 200 --warmup 20 > gpurun_out/r04f_bench_n64.json 2> gpurun_out/r04f_bench_n64.err; echo "bench64 rc=$?"
python bench.py --heterogeneous --no-cpu-baseline > gpurun_out/r04f_bench_n256het.json 2> gpurun_out/r04f_bench_n256het.err; echo "het rc=$?"
BUNNY_CACHE=/tmp/bunny100k.pkl python tools/bunny_time.py r04f > gpurun_out/r04f_bunny100k.txt 2>&1; BUNNY_CACHE=/tmp/bunny100k.pkl python tools/bunny_time.py r04f >> gpurun_out/r04f_bunny100k.txt 2>&1; cat gpurun_out/r04f_bunny100k.txt
timeout -k 10 400 bash tools/mode_clock_probe.sh r04f 16 > /dev/null 2>&1; cat gpurun_out/r04f_mode_clock_probe.txt
python -c "
import json
for f in ('n256', 'n64', 'n256het'):
    j = json.load(open('gpurun_out/r04f_bench_%s.json' % f)); print(f, j['ms_per_step'], j.get('sustained_ms_per_step'), j['roofline']['frac'], j['config']['parity'].get('golden', {}).get('bitwise'))"
