#!/bin/bash
set -o pipefail
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "plan or timed or enqueue or horizon" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --horizon 200 > $O/n200_plan_$i.json 2> $O/n200.err || { tail $O/n200.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --horizon 200 --plan-ways 0 > $O/n200_noplan_$i.json 2>> $O/n200.err || { tail $O/n200.err; exit 1; }
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03t/n200_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1], "value %.3e us/step %.2f frac %.3f region %.3f %s" % (d["value"], 1e3*d["ms_per_step"], r["frac"], r["timed_region"]["frac"], r["kernel"]), d["solved"]["ik"], d["solved"]["mpc"])
PY
