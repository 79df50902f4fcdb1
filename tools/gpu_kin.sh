#!/bin/bash
# kinematics kernel work: its parity tests, the tick tests that run it, bench lines
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/kin.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_tick_pipeline.py -m gpu -q -x -k "kin or hull or tick" > $L 2>&1 || { tail -40 $L; exit 1; }
timeout -k 10 300 python bench.py --workload kin --batch 65536 --steps 50 --warmup 5 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 300 python bench.py --workload kin --batch 8192 --steps 50 --warmup 5 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 200 --warmup 20 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L | cut -c1-700 | tail -8
