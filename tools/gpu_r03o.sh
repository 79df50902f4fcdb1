#!/bin/bash
set -o pipefail
O=gpurun_out/r03o
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "plan or timed or enqueue" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
b() { n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.err || { tail -20 $O/$n.err; exit 1; }
  python3 - $O/$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "value %.3e us/step %.2f frac %.3f (resident %.3f) region_frac %.3f" % (d["value"], 1e3 * d["ms_per_step"], r["frac"], r.get("frac_resident_inputs", 0), r.get("timed_region", {}).get("frac", 0)), r["kernel"], {k: d["solved"].get(k) for k in ("ik", "mpc", "golden_max_abs_err", "golden_active_set_mismatches")})
PY
}
for w in 1 2 3 4; do b drv_w$w --steps 20 --warmup 5 --plan-ways $w; done
b drv_w0 --steps 20 --warmup 5 --plan-ways 0
for w in 1 2 3 4; do b s200_w$w --plan-ways $w; done
b s200_w0 --plan-ways 0
b b65536_w1 --batch 65536 --steps 50 --warmup 5 --plan-ways 1
b b65536_w2 --batch 65536 --steps 50 --warmup 5 --plan-ways 2
b b65536_w0 --batch 65536 --steps 50 --warmup 5 --plan-ways 0
