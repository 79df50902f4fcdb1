#!/bin/bash
set -o pipefail
L=gpurun_out/pipe2.log; : > $L
for P in 3 4; do echo "== pipelines $P steps 200" >> $L; timeout -k 10 300 python bench.py --steps 200 --warmup 20 --pipelines $P --no-cpu-baseline >> $L 2>&1 || exit 1
 echo "== pipelines $P steps 20" >> $L; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --pipelines $P --no-cpu-baseline >> $L 2>&1 || exit 1; done
echo "== 65536 poll" >> $L; timeout -k 10 300 python bench.py --steps 50 --warmup 5 --batch 65536 --no-cpu-baseline >> $L 2>&1 || exit 1
echo "== 65536 nopoll" >> $L; WCQP_BENCH_NOPOLL=1 timeout -k 10 300 python bench.py --steps 50 --warmup 5 --batch 65536 --no-cpu-baseline >> $L 2>&1 || exit 1
echo "== 65536 pipelines 2" >> $L; timeout -k 10 300 python bench.py --steps 50 --warmup 5 --batch 65536 --pipelines 2 --no-cpu-baseline >> $L 2>&1 || exit 1
echo "== 16384" >> $L; timeout -k 10 300 python bench.py --steps 100 --warmup 5 --batch 16384 --no-cpu-baseline >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L | python3 tools/pipe_summary.py
