mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_9.log | tail -30
timeout -k 10 300 python tools/ab_ik.py > gpurun_out/ab_ik.json 2> gpurun_out/ab_ik.err; cat gpurun_out/ab_ik.json; tail -3 gpurun_out/ab_ik.err
