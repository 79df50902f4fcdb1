#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/run27.log
for s in 1 4 1000; do
  WCQP_BENCH_EVENT_STRIDE=$s timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline >> gpurun_out/run27.log 2>&1 || exit 1
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ragged" >> gpurun_out/run27.log 2>&1 || { tail -30 gpurun_out/run27.log; exit 1; }
grep -v amdgpu.ids gpurun_out/run27.log | cut -c1-260
