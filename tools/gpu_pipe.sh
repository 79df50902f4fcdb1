#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/pipe.log; : > $L
for P in 1 2 3; do
  for S in 200 20; do
    echo "== pipelines $P steps $S" >> $L
    timeout -k 10 300 python bench.py --steps $S --warmup $((S/10+2)) --pipelines $P >> $L 2>&1 || { tail -20 $L; exit 1; }
  done
done
echo "== default, 65536" >> $L
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --batch 65536 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L | python3 tools/pipe_summary.py
