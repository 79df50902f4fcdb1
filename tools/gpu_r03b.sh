#!/bin/bash
# round 3: skewed tick (IK(t) + MPC(t+1) in one launch) and the compact kinematics -> IK hand-off: parity tests, then the tick bench
set -o pipefail
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_tick_pipeline.py -m gpu -q -x > $O/pytest_tick.log 2>&1 || { tail -60 $O/pytest_tick.log; exit 1; }
tail -1 $O/pytest_tick.log
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for v in "tables:--tick-tables" "kin:" "kin_dense:--tick-dense-handoff" "kin_s1:--streams 1" "kin_s2:--streams 2" "tables_s1:--tick-tables --streams 1" "tables_s3:--tick-tables --streams 3"; do
  n=${v%%:*}; extra=${v#*:}
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $extra > $O/tick_$n.json 2> $O/tick_$n.err || { tail -20 $O/tick_$n.err; exit 1; }
  python3 - $O/tick_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.3e us/tick %.2f frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], d["roofline"]["frac"]), {k: d["solved"][k] for k in ("mpc_fail", "ik_fail", "robots_with_ik_fail")})
PY
done
