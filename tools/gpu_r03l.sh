#!/bin/bash
set -o pipefail
O=gpurun_out/r03l
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1 || { tail -80 $O/pytest.log; exit 1; }
tail -12 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03l/bench_driver.json").read().strip().splitlines()[-1])
print("driver form: value %.3e us/step %.2f frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], d["roofline"]["frac"]), d["solved"], {k: d["cpu_baseline"][k] for k in ("value", "mpc_warm_qps", "cores")})
PY
