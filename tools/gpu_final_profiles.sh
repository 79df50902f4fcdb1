mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_tick -- python $R/bench.py --workload tick --batch 8192 --steps 300 --warmup 20 > $R/gpurun_out/prof_final_tick.log 2>&1
cd $R
bash tools/pmc/collect.sh > gpurun_out/pmc_collect.log 2>&1
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > gpurun_out/bench_final_b65536.json 2>> gpurun_out/bench_final.err
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 > gpurun_out/bench_final_tick.json 2>> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
