#!/bin/bash
mkdir -p gpurun_out/ways20
for w in 1 2 3 4 5 8 10 20 4; do
  timeout -k 10 250 python bench.py --steps 20 --warmup 5 --plan-ways $w --no-cpu-baseline > gpurun_out/ways20/w$w.json 2> gpurun_out/ways20/w$w.err || { tail -5 gpurun_out/ways20/w$w.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/ways20/w$w.json") if l.startswith("{")][-1]
r=d["roofline"]
print("ways=$w value=%.4g in-kernel us/step=%.2f frac=%.3f timed_frac=%.3f" % (d["value"], r["avg_ms_per_step"]*1e3, r["frac"], r["timed_region"]["frac"]), flush=True)
PY
done
