#!/bin/bash
# One round's committed evidence, run on the GPU box: bash tools/profile_round.sh <round tag, e.g. r01g>
# kernel trace + stats (n256, n64; the timed region of the default bench run: 3 warm-up + 20 timed ticks), FETCH_SIZE / WRITE_SIZE passes (n256), bench JSON lines (n256 with CPU baseline, n64).
TAG=${1:?round tag}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 256 64; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_n$n -o kt --output-format csv -- python3 $R/bench.py --n $n --steps ${STEPS:-20} --warmup ${WARMUP:-3} --no-cpu-baseline --no-parity --no-sustained --allow-stale-traffic > $R/gpurun_out/prof_${TAG}_n$n.out 2> $R/gpurun_out/prof_${TAG}_n$n.err || exit 1
done
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_${TAG} -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --allow-stale-traffic > /dev/null 2> $R/gpurun_out/pmc_fetch_${TAG}.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_${TAG} -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --allow-stale-traffic > /dev/null 2> $R/gpurun_out/pmc_write_${TAG}.err || exit 1
# the data-layout worst case beside the headline's best case: per-particle masses, per-spring rest lengths (bench.py --heterogeneous)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_n256het -o kt --output-format csv -- python3 $R/bench.py --heterogeneous --steps ${STEPS:-20} --warmup ${WARMUP:-3} --no-cpu-baseline --no-parity --no-sustained > $R/gpurun_out/prof_${TAG}_n256het.out 2> $R/gpurun_out/prof_${TAG}_n256het.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_${TAG}_het -o pmc --output-format csv -- python3 $R/bench.py --heterogeneous --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained > /dev/null 2> $R/gpurun_out/pmc_fetch_${TAG}_het.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_${TAG}_het -o pmc --output-format csv -- python3 $R/bench.py --heterogeneous --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained > /dev/null 2> $R/gpurun_out/pmc_write_${TAG}_het.err || exit 1
cd $R
python tools/prof_summary.py --round $TAG --kt gpurun_out/prof_${TAG}_n256het --fetch gpurun_out/pmc_fetch_${TAG}_het --write gpurun_out/pmc_write_${TAG}_het --key n256het_tile512_gpus1 > /dev/null || exit 1
python tools/prof_summary.py --round $TAG --kt gpurun_out/prof_${TAG}_n256 --fetch gpurun_out/pmc_fetch_${TAG} --write gpurun_out/pmc_write_${TAG} --key n256_tile512_gpus1 > /dev/null || exit 1
python tools/prof_summary.py --round $TAG --kt gpurun_out/prof_${TAG}_n64 --key n64_tile512_gpus1 > /dev/null || exit 1
python bench.py > gpurun_out/${TAG}_bench_n256.json 2> gpurun_out/${TAG}_bench_n256.err || exit 1
python bench.py --n 64 --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_n64.json 2> gpurun_out/${TAG}_bench_n64.err || exit 1
python bench.py --heterogeneous --no-cpu-baseline > gpurun_out/${TAG}_bench_n256het.json 2> gpurun_out/${TAG}_bench_n256het.err || exit 1
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* profiles/hbm_traffic.json gpurun_out/profiles_out/
tail -c 1500 gpurun_out/${TAG}_bench_n256.json
