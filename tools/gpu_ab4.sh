#!/bin/bash
# bench.py (cold-input IK kernel) with the product library and named variants, at B = 65536
mkdir -p gpurun_out; L=gpurun_out/ab4.log; : > $L
D=$PWD/walking-controllers_amd/csrc/build/diag
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export WCQP_LIB_PATH=$D/libwcqp_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --batch 65536 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', j['ms_per_step'], j['kernels']['ik_ms'], j['solved'])" >> $L
done
cat $L
