#!/bin/bash
# End-of-round profiles: rocprofv3 kernel stats of the bench command, PMC passes (separate runs), final bench lines.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_b65536 -- python $R/bench.py --steps 50 --warmup 10 --batch 65536 --no-cpu-baseline > $R/gpurun_out/prof_final_b65536.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_tick -- python $R/bench.py --workload tick --batch 8192 --steps 300 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_final_tick.log 2>&1
cd $R
bash tools/pmc/collect.sh > gpurun_out/pmc_collect.log 2>&1
bash tools/pmc/collect_ik.sh ik3_final 4 > /dev/null 2>&1
python tools/pmc/summarize_ik.py gpurun_out/pmc_ik > gpurun_out/pmc_ik_summary.json
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > gpurun_out/bench_final_b65536.json 2>> gpurun_out/bench_final.err
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_tick.json 2>> gpurun_out/bench_final.err
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-graph --no-cpu-baseline > gpurun_out/bench_final_tick_nograph.json 2>> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json | cut -c1-400
