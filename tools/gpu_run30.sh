#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/run30.log
D=$PWD/walking-controllers_amd/csrc/build/diag
for v in "" tNO_GLUE tNO_POST tNO_ATOMIC; do
  if [ -n "$v" ]; then export WCQP_LIB_PATH=$D/libwcqp_$v.so; fi
  echo "variant $v" >> gpurun_out/run30.log
  timeout -k 10 200 python bench.py --workload tick --batch 8192 --steps 500 --warmup 20 --no-graph --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])" >> gpurun_out/run30.log
done
cat gpurun_out/run30.log
