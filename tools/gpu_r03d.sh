#!/bin/bash
set -o pipefail
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_tick_pipeline.py tests/test_kinematics.py -m gpu -q -x > $O/pytest_tick.log 2>&1 || { tail -60 $O/pytest_tick.log; exit 1; }
tail -1 $O/pytest_tick.log
for v in "tables:--tick-tables" "kin:" "kin_s1:--streams 1"; do
  n=${v%%:*}; extra=${v#*:}
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $extra > $O/tick_$n.json 2> $O/tick_$n.err || { tail -20 $O/tick_$n.err; exit 1; }
  python3 - $O/tick_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.3e us/tick %.2f frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], d["roofline"]["frac"]), {k: d["solved"][k] for k in ("mpc_fail", "ik_fail", "robots_with_ik_fail")})
PY
done
timeout -k 10 900 python tools/walk_diag.py --big 65536 > $O/walk_diag.jsonl 2> $O/walk_diag.err || { tail -20 $O/walk_diag.err; exit 1; }
cut -c1-400 $O/walk_diag.jsonl
