#!/bin/bash
# EXPERIMENT: one persistent launch per tick (SB_PERSISTENT=1) against the hipGraph of kernel launches, small cubes.
# bench.py's golden / small / live legs check the bits of the very runs that are timed.
for n in 64 48 96; do
  for mode in graph persistent graph persistent; do
    if [ $mode = persistent ]; then export SB_PERSISTENT=1; else unset SB_PERSISTENT; fi
    timeout -k 10 240 python bench.py --n $n --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/persist_${n}_${mode}.json 2> gpurun_out/persist_${n}_${mode}.err
    rc=$?
    python - <<PY
import json
try:
    j = json.load(open("gpurun_out/persist_${n}_${mode}.json"))
    p = j["config"]["parity"]
    print("n=$n %-10s rc=$rc  %.4f ms/tick  golden %s small %s" % ("$mode", j["ms_per_step"], (p.get("golden") or {}).get("bitwise"), (p.get("small") or {}).get("bitwise")))
except Exception as e:
    print("n=$n $mode rc=$rc FAILED", e); print(open("gpurun_out/persist_${n}_${mode}.err").read()[-600:])
PY
    [ $rc -ne 0 ] && [ $mode = persistent ] && { echo "stopping: persistent run failed"; exit 0; }
  done
done
