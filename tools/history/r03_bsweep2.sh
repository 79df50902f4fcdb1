#!/bin/bash
mkdir -p gpurun_out/bsweep
for cfg in "32768 100 4 12" "32768 100 4 6" "16384 100 4 22" "16384 100 4 6" "4096 200 4 88" "4096 200 4 44"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --batch $1 --steps $2 --warmup 8 --plan-ways $3 --input-sets $4 --no-cpu-baseline > gpurun_out/bsweep/b$1_w$3_s$4.json 2> gpurun_out/bsweep/b$1_w$3_s$4.err || exit 1
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/bsweep/b$1_w$3_s$4.json") if l.startswith("{")][-1]
r=d["roofline"]
print("B=$1 ways=$3 sets=$4 value=%.4g in-kernel us/step=%.2f ns/robot=%.3f frac=%.3f timed_frac=%.3f" % (d["value"], r["avg_ms_per_step"]*1e3, r["avg_ms_per_step"]*1e6/$1, r["frac"], r["timed_region"]["frac"]), flush=True)
PY
done
