#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
n=bench_b4096_one_batch_at_a_time
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 --pipelines 1 > $O/prof_$n.log 2>&1 || tail -3 $O/prof_$n.log
f=$(ls $O/prof_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
head -4 $O/${n}_kernel_stats.csv | cut -c1-200
cd $R
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 > $O/bench_b4096.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_b4096_driver.json 2>> $O/bench.err
python3 -c "
import json
for f in ('bench_b4096','bench_b4096_driver'):
    d=json.loads(open('$O/'+f+'.json').read()); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('timed_region'), r.get('traffic'))
"
