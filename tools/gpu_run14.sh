mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_14.log | tail -4
timeout -k 10 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_14.json 2> gpurun_out/bench_14.err; cat gpurun_out/bench_14.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > gpurun_out/bench_14_b65536.json 2>> gpurun_out/bench_14.err; cat gpurun_out/bench_14_b65536.json
