#!/bin/bash
set -o pipefail
O=gpurun_out/r03u
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu-baseline --horizon 200 > $O/bench_b4096_n200.json 2> $O/n200.err || { tail $O/n200.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.json 2> $O/drv.err || { tail $O/drv.err; exit 1; }
WCQP_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_gloo_rehearsal.json 2> $O/g2.err || { tail $O/g2.err; exit 1; }
python3 - <<'PY'
import json
for f in ("bench_b4096_n200", "bench_default_driver", "bench_gpus2_gloo_rehearsal"):
    d = json.loads([l for l in open("gpurun_out/r03u/%s.json" % f).read().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
    print(f, "n_gpus", d["n_gpus"], "value %.3e us/step %.2f frac %.3f traffic %s" % (d["value"], 1e3 * d["ms_per_step"], r["frac"], r.get("traffic")), d["solved"].get("golden_active_set_mismatches"))
PY
