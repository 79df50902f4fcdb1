#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
: > gpurun_out/stamps.log
for args in "64 100 3" "65536 0.5 3" "65536 0.5 2"; do
  WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py $args >> gpurun_out/stamps.log 2>&1 || exit 1
done
grep median gpurun_out/stamps.log
