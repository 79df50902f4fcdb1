#!/bin/bash
mkdir -p gpurun_out/ways20
i=0
for w in 4 25 100 4 100; do
  i=$((i+1))
  timeout -k 10 280 python bench.py --batch 65536 --steps 100 --warmup 100 --plan-ways $w --no-cpu-baseline > gpurun_out/ways20/b$i.json 2> gpurun_out/ways20/b$i.err || { tail -5 gpurun_out/ways20/b$i.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/ways20/b$i.json") if l.startswith("{")][-1]
r=d["roofline"]
print("B=65536 steps=100 ways=$w value=%.4g in-kernel us/step=%.2f frac=%.3f timed_frac=%.3f sets=%d" % (d["value"], r["avg_ms_per_step"]*1e3, r["frac"], r["timed_region"]["frac"], d["config"]["input_sets"]), flush=True)
PY
done
