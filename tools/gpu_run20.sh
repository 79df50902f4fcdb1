#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/run20.log
for B in 64 256 1024 2048 4096 8192 16384 32768; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> gpurun_out/run20.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/run20.log
