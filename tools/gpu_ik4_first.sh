#!/bin/bash
# first GPU contact of the base-eliminated IK kernel: parity tests of the IK, then timings (3 vs 2 waves per SIMD)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/ik4_first.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ik" > $L 2>&1 || { tail -40 $L; exit 1; }
for B in 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || { tail -20 $L; exit 1; }
  WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_w2.so timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || { tail -20 $L; exit 1; }
  timeout -k 10 120 python tools/time_alg.py $B 100 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L | cut -c1-400 | tail -12
