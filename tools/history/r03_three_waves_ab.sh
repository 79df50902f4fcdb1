#!/bin/bash
O=gpurun_out/r03w
mkdir -p $O
b() { n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; return; }
  python3 - $O/$n.json $n <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[2], "value %.3e us/step %.2f" % (d["value"], 1e3 * d["ms_per_step"]), d["solved"].get("golden_active_set_mismatches"), d["solved"].get("ik_fail"))
PY
}
for lib in "" w3k3 w3k2; do
  if [ -n "$lib" ]; then export WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so; fi
  b plan_${lib:-prod}
  b plan65536_${lib:-prod} --batch 65536 --steps 50 --warmup 5
  b tick_kin_${lib:-prod} --workload tick --batch 8192 --steps 1000 --warmup 24
  b tick_tab_${lib:-prod} --workload tick --batch 8192 --steps 1000 --warmup 24 --tick-tables
done
