#!/bin/bash
set -o pipefail
O=gpurun_out/r03m
mkdir -p $O
true
true
b() { n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.err || { tail -20 $O/$n.err; exit 1; }
  python3 - $O/$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.3e us/step %.2f frac %.3f region_frac %s host_us %.2f" % (d["value"], 1e3 * d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("timed_region", {}).get("frac"), d["config"].get("host_enqueue_us_per_step", 0)), d["config"].get("timed_steps_enqueued_as"), {k: d["solved"].get(k) for k in ("golden_max_abs_err", "golden_active_set_mismatches")})
PY
}
for i in 1 2 3; do b drv_graph_$i --steps 20 --warmup 5; done
for i in 1 2 3; do b drv_nograph_$i --steps 20 --warmup 5 --no-step-graph; done
b s200_graph
b s200_nograph --no-step-graph
b b65536_graph --batch 65536 --steps 50 --warmup 5
