#!/bin/bash
# round 3, first GPU call: suite at HEAD, tick bench baselines (constant Jacobians / kinematics), walk diagnostics
set -o pipefail
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-tables > $O/tick_tables.json 2> $O/tick_tables.err || { tail -20 $O/tick_tables.err; exit 1; }
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline > $O/tick_kin.json 2> $O/tick_kin.err || { tail -20 $O/tick_kin.err; exit 1; }
python3 - <<'EOF'
import json
for f in ("tick_tables", "tick_kin"):
    d = json.loads(open("gpurun_out/r03a/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value %.3e ms/tick %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), d["solved"])
EOF
timeout -k 10 900 python tools/walk_diag.py > $O/walk_diag.jsonl 2> $O/walk_diag.err || { tail -20 $O/walk_diag.err; exit 1; }
cat $O/walk_diag.jsonl | cut -c1-1500
