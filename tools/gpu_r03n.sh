#!/bin/bash
set -o pipefail
O=gpurun_out/r03n
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
run() { n=$1; shift
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline "$@" > $O/tick_$n.json 2> $O/tick_$n.err || { tail -20 $O/tick_$n.err; exit 1; }
  python3 - $O/tick_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.3e us/tick %.2f frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], d["roofline"]["frac"]), {k: d["solved"][k] for k in ("mpc_fail", "ik_fail", "robots_with_ik_fail", "ik_hot_start_tried")})
PY
}
run kin_fused
run kin_fused_s1 --streams 1
run kin_fused_k1 --ticks-per-launch 1
run kin_fused_k8 --ticks-per-launch 8
run kin_compact --tick-kin-handoff compact
run tables --tick-tables
