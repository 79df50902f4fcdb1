mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu_7.log | tail -25
