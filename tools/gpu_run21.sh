#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ik" > gpurun_out/run21.log 2>&1 || { tail -40 gpurun_out/run21.log; exit 1; }
for B in 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> gpurun_out/run21.log 2>&1 || exit 1
done
L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
for args in "64 100 4" "65536 0.5 4"; do
  WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py $args >> gpurun_out/run21.log 2>&1 || { tail -20 gpurun_out/run21.log; exit 1; }
done
grep -v amdgpu.ids gpurun_out/run21.log | tail -8
