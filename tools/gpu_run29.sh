#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tickf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tickf -- python $R/bench.py --workload tick --batch 8192 --steps 300 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_tickf.log 2>&1
cat $R/gpurun_out/prof_tickf/*/*kernel_stats.csv | cut -c1-60,200-420 | head -6
