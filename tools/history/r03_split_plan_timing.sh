#!/bin/bash
mkdir -p gpurun_out/split
for cfg in "4096 20" "4096 200" "65536 50"; do set -- $cfg
timeout -k 10 300 python tools/split_plan_timing.py $1 $2 > gpurun_out/split/b$1_s$2.json 2> gpurun_out/split/b$1_s$2.err || { tail -5 gpurun_out/split/b$1_s$2.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/split/b$1_s$2.json"))
print("B=$1 steps=$2", {k: (round(v["us_per_step_median"],2), "%.4g" % v["qp_per_s_median"]) for k,v in d.items() if isinstance(v, dict)})
PY
done
