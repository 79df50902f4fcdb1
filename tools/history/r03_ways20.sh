#!/bin/bash
# fixed ways of the plan: 20-step driver form and 200 steps, up to one record per workgroup (ways = steps)
mkdir -p gpurun_out/ways20
i=0
for cfg in "20 4" "20 10" "20 20" "20 4" "20 20" "200 4" "200 20" "200 50" "200 200" "200 4" "200 50"; do set -- $cfg
  i=$((i+1))
  timeout -k 10 250 python bench.py --steps $1 --warmup 5 --plan-ways $2 --no-cpu-baseline > gpurun_out/ways20/r$i.json 2> gpurun_out/ways20/r$i.err || { tail -5 gpurun_out/ways20/r$i.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/ways20/r$i.json") if l.startswith("{")][-1]
r=d["roofline"]
print("steps=$1 ways=$2 value=%.4g in-kernel us/step=%.2f frac=%.3f timed_frac=%.3f sets=%d" % (d["value"], r["avg_ms_per_step"]*1e3, r["frac"], r["timed_region"]["frac"], d["config"]["input_sets"]), flush=True)
PY
done
