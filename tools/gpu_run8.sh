mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 > gpurun_out/bench_tick_graph.json 2> gpurun_out/bench_tick.err; cat gpurun_out/bench_tick_graph.json; tail -3 gpurun_out/bench_tick.err
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-graph > gpurun_out/bench_tick_nograph.json 2>> gpurun_out/bench_tick.err; cat gpurun_out/bench_tick_nograph.json
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_8.json 2>> gpurun_out/bench_tick.err; cat gpurun_out/bench_8.json
