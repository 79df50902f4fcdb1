#!/bin/bash
# A/B of a diagnostic library variant against the product library: IK parity tests on the product library, then
# both timed at two batch sizes.   gpurun -- 'bash tools/gpu_ab.sh <variant-name>'
set -o pipefail
mkdir -p gpurun_out; L=gpurun_out/ab.log; : > $L
D=$PWD/walking-controllers_amd/csrc/build/diag
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ik" >> $L 2>&1 || { tail -30 $L; exit 1; }
for B in 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || exit 1
  WCQP_LIB_PATH=$D/libwcqp_$1.so timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || exit 1
done
grep -v amdgpu.ids $L | tail -6
