set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_debug_ik.py > gpurun_out/debug_ik.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_2.log | tail -15
