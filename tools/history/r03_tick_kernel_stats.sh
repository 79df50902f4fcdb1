#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc/bench_4096_FETCH_SIZE $O/pmc/bench_4096_WRITE_SIZE
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/bench_4096_$C -- python3 $R/bench.py --steps 88 --warmup 5 --batch 4096 --no-cpu-baseline > $O/pmc_bench_4096_$C.log 2>&1 < /dev/null
done
cd $R
python3 tools/pmc/summarize.py $O/pmc > $O/pmc_summary.json 2> $O/pmc_summary.err
python3 tools/pmc/make_traffic.py $O/pmc_summary.json > $O/traffic.json 2> $O/traffic.err
python3 -c "
import json; d=json.load(open('$O/traffic.json')); print(d['per_batch']['4096'])"
