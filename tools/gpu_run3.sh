set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu_3.log | tail -8
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_3.json 2> gpurun_out/bench_3.err; tail -3 gpurun_out/bench_3.err; cat gpurun_out/bench_3.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > gpurun_out/bench_3_b65536.json 2>> gpurun_out/bench_3.err; cat gpurun_out/bench_3_b65536.json
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof3.log; find $GRAFT_REPO_ROOT/gpurun_out/prof3 -name "*stats*" | head
