#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/run22.log
L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
for args in "64 0.5 4" "65536 0.5 4"; do
  WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py $args >> gpurun_out/run22.log 2>&1 || { tail -20 gpurun_out/run22.log; exit 1; }
done
grep -v amdgpu.ids gpurun_out/run22.log | tail -8
