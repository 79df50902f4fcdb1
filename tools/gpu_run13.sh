mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_13.json 2> gpurun_out/bench_13.err; cat gpurun_out/bench_13.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > gpurun_out/bench_13_b65536.json 2>> gpurun_out/bench_13.err; cat gpurun_out/bench_13_b65536.json
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 > gpurun_out/bench_13_tick.json 2>> gpurun_out/bench_13.err; cat gpurun_out/bench_13_tick.json
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-graph > gpurun_out/bench_13_tick_ng.json 2>> gpurun_out/bench_13.err; cat gpurun_out/bench_13_tick_ng.json
