#!/bin/bash
# 3 waves per SIMD builds of ik4 against the product library: parity tests, then the qp bench at 4096 / 16384 / 65536
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/w3.log; : > $L
for lib in "" w3k3 w3k2; do
  if [ -n "$lib" ]; then export WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so; fi
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_tick_pipeline.py -m gpu -q -x 2>&1 | tail -1
  for B in 4096 16384 65536; do
    S=$((819200 / B)); [ $S -gt 200 ] && S=200
    echo "== lib ${lib:-product} batch $B" >> $L
    timeout -k 10 300 python bench.py --batch $B --steps $S --warmup 5 --no-cpu-baseline >> $L 2>&1 || { tail -20 $L; exit 1; }
  done
  echo "== lib ${lib:-product} tick tables 8192" >> $L
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 500 --warmup 24 --no-cpu-baseline --tick-tables 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  tick value %.3e ms/tick %.4f' % (d['value'], d['ms_per_step']))" >> $L
done
grep -v amdgpu.ids $L | python3 tools/pipe_summary.py; grep "tick value" $L
