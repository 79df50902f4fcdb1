#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
D=$PWD/walking-controllers_amd/csrc/build/diag
timeout -k 10 60 tools/_build/dpp_selftest > gpurun_out/run16.log 2>&1 || { cat gpurun_out/run16.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -q -x >> gpurun_out/run16.log 2>&1 || { tail -30 gpurun_out/run16.log; exit 1; }
for B in 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> gpurun_out/run16.log 2>&1 || exit 1
  WCQP_LIB_PATH=$D/libwcqp_w3.so timeout -k 10 120 python tools/time_alg.py $B 0.5 >> gpurun_out/run16.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/run16.log | tail -12
