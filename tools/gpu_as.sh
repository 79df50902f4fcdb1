#!/bin/bash
# active-set work on ik4: IK + tick parity tests, IK kernel timings, per-phase stamps
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/as.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_tick_pipeline.py -m gpu -q -x > $L 2>&1 || { tail -40 $L; exit 1; }
for B in 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so timeout -k 10 120 python tools/stamps4.py 4096 0.5 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 300 python bench.py --steps 200 --warmup 20 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L | cut -c1-1200 | tail -12
