#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/run28.log 2>&1 || { tail -40 gpurun_out/run28.log; exit 1; }
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-cpu-baseline >> gpurun_out/run28.log 2>&1 || { tail -20 gpurun_out/run28.log; exit 1; }
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-graph --no-cpu-baseline >> gpurun_out/run28.log 2>&1 || { tail -20 gpurun_out/run28.log; exit 1; }
timeout -k 10 300 python bench.py --workload tick --steps 1000 --warmup 20 --no-cpu-baseline >> gpurun_out/run28.log 2>&1 || { tail -20 gpurun_out/run28.log; exit 1; }
grep -v amdgpu.ids gpurun_out/run28.log | cut -c1-330 | tail -6
