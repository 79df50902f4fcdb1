set -x
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu_1.log | tail -40
