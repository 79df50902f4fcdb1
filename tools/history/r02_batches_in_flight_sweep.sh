#!/bin/bash
# the qp bench over batch sizes and batches in flight: one line each (tools/pipe_summary.py)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/pipe.log; : > $L
for B in ${BATCHES:-4096 8192 16384 32768}; do
  for P in ${PIPES:-1 2 3}; do
    S=$((819200 / B)); [ $S -gt 200 ] && S=200
    echo "== batch $B pipelines $P steps $S" >> $L
    timeout -k 10 300 python bench.py --batch $B --steps $S --warmup 5 --pipelines $P --no-cpu-baseline >> $L 2>&1 || { tail -20 $L; exit 1; }
  done
done
grep -v amdgpu.ids $L | python3 tools/pipe_summary.py
