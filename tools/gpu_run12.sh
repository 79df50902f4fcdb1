mkdir -p gpurun_out
timeout -k 10 200 python tools/time_alg.py > gpurun_out/time_alg.log 2>&1
grep ms gpurun_out/time_alg.log
