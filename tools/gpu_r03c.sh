#!/bin/bash
# kernel-level times of the tick variants (rocprofv3 --kernel-trace --stats)
set -o pipefail
O=gpurun_out/r03c
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "kin_s1:--streams 1" "kin_s3:" "tables_s1:--tick-tables --streams 1"; do
  n=${v%%:*}; extra=${v#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$n -- python3 $R/bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $extra > $R/$O/$n.json 2> $R/$O/$n.err < /dev/null || { tail -20 $R/$O/$n.err; exit 1; }
  f=$(find $R/$O/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n $f"
  if [ -n "$f" ]; then cut -d, -f1-4,8 "$f" | sed 's/void (anonymous namespace):://' | cut -c1-150 | head -6; cp "$f" $R/$O/${n}_kernel_stats.csv; fi
  rm -rf $R/$O/prof_$n
done
