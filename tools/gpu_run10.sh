mkdir -p gpurun_out
timeout -k 10 60 tools/_build/dpp_selftest > gpurun_out/dpp_selftest.log 2>&1; cat gpurun_out/dpp_selftest.log
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_10.log | tail -5
timeout -k 10 300 python tools/ab_ik.py > gpurun_out/ab_ik.json 2> gpurun_out/ab_ik.err; tail -2 gpurun_out/ab_ik.err
