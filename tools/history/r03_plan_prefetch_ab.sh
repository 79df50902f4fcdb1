#!/bin/bash
set -o pipefail
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "plan or timed or enqueue" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
b() { n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.err || { tail -20 $O/$n.err; exit 1; }
  python3 - $O/$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "value %.3e us/step %.2f frac %.3f region_frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], r["frac"], r.get("timed_region", {}).get("frac", 0)), {k: d["solved"].get(k) for k in ("golden_max_abs_err", "golden_active_set_mismatches")})
PY
}
for pf in 1 0 1 0; do
  export WCQP_PLAN_PREFETCH=$pf
  b drv_pf${pf}_$RANDOM --steps 20 --warmup 5
  b s200_pf${pf}_$RANDOM
done
export WCQP_PLAN_PREFETCH=1
b b65536_pf1 --batch 65536 --steps 50 --warmup 5
b s200_w2_pf1 --plan-ways 2
export WCQP_PLAN_PREFETCH=0
b b65536_pf0 --batch 65536 --steps 50 --warmup 5
