#!/bin/bash
# plan kernel: in-kernel time per robot-step against the batch size and the number of ways
mkdir -p gpurun_out/bsweep
for cfg in "4096 200 4" "8192 200 4" "8192 200 2" "16384 100 4" "16384 100 2" "16384 100 1" "32768 100 2" "32768 100 4" "65536 100 4" "65536 100 1" "4096 200 8"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --batch $1 --steps $2 --warmup 8 --plan-ways $3 --no-cpu-baseline > gpurun_out/bsweep/b$1_w$3.json 2> gpurun_out/bsweep/b$1_w$3.err || exit 1
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/bsweep/b$1_w$3.json") if l.startswith("{")][-1]
r=d["roofline"]
print("B=$1 ways=$3 value=%.4g in-kernel us/step=%.2f ns/robot=%.3f frac=%.3f timed_frac=%.3f" % (d["value"], r["avg_ms_per_step"]*1e3, r["avg_ms_per_step"]*1e6/$1, r["frac"], r["timed_region"]["frac"]), flush=True)
PY
done
