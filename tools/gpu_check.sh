#!/bin/bash
# One gpurun call that answers "is the current tree correct and how fast is it":
#   gpurun --timeout 1100 -- 'bash tools/gpu_check.sh'           # GPU test suite, IK kernels side by side, bench lines
#   gpurun --timeout 1100 -- 'bash tools/gpu_check.sh stamps'    # + per-phase cycles of the IK kernels (build the
#                                                                #   diagnostic library first: tools/build_variant.sh stamps -DWCQP_IK_STAMPS)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/check.log
timeout -k 10 60 tools/_build/dpp_selftest > $L 2>&1 || { cat $L; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x >> $L 2>&1 || { tail -40 $L; exit 1; }
for B in 1024 4096 65536; do
  timeout -k 10 120 python tools/time_alg.py $B 0.5 >> $L 2>&1 || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 300 python bench.py --batch 65536 --steps 50 --warmup 10 --no-cpu-baseline >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline >> $L 2>&1 || { tail -20 $L; exit 1; }
if [ "$1" = "stamps" ]; then
  S=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
  for args in "64 0.5 4" "65536 0.5 4" "64 0.5 3"; do
    WCQP_LIB_PATH=$S timeout -k 10 100 python tools/stamps.py $args >> $L 2>&1 || { tail -20 $L; exit 1; }
  done
fi
grep -v amdgpu.ids $L | cut -c1-420 | tail -14
