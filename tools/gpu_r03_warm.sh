#!/bin/bash
mkdir -p gpurun_out/warm
for cfg in "4096 20 5" "4096 20 3000" "4096 20 10000" "4096 200 20" "4096 200 5000"; do
  set -- $cfg
  timeout -k 10 250 python bench.py --batch $1 --steps $2 --warmup $3 --no-cpu-baseline > gpurun_out/warm/b$1_s$2_w$3.json 2> gpurun_out/warm/b$1_s$2_w$3.err || { tail -5 gpurun_out/warm/b$1_s$2_w$3.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/warm/b$1_s$2_w$3.json") if l.startswith("{")][-1]
r=d["roofline"]
print("B=$1 steps=$2 warmup=$3 value=%.4g ms/step=%.5f in-kernel us/step=%.2f frac=%.3f timed_frac=%.3f" % (d["value"], d["ms_per_step"], r["avg_ms_per_step"]*1e3, r["frac"], r["timed_region"]["frac"]), flush=True)
PY
done
