mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_16.log | tail -4
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 > gpurun_out/bench_16_tick.json 2> gpurun_out/bench_16.err; cat gpurun_out/bench_16_tick.json
timeout -k 10 600 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 20 --no-graph > gpurun_out/bench_16_tick_ng.json 2>> gpurun_out/bench_16.err; cat gpurun_out/bench_16_tick_ng.json
WCQP_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/bench_16_rehearse2.json 2> gpurun_out/bench_16_rehearse2.err; cat gpurun_out/bench_16_rehearse2.json; tail -3 gpurun_out/bench_16_rehearse2.err
WCQP_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 50 --warmup 5 --workload tick --batch 1024 > gpurun_out/bench_16_rehearse2_tick.json 2>> gpurun_out/bench_16_rehearse2.err; cat gpurun_out/bench_16_rehearse2_tick.json
