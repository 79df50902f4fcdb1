#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/run26.log 2>&1 || { tail -40 gpurun_out/run26.log; exit 1; }
: > gpurun_out/run26_sweep.log
for B in 64 256 1024 2048 4096 8192 16384 32768 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> gpurun_out/run26_sweep.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/run26.log | tail -3; grep '"B"' gpurun_out/run26_sweep.log
