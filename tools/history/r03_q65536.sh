#!/bin/bash
# work-queue plan (group-major units) against 4 fixed ways with EVERY step reading input arrays of its own (--input-sets = steps):
# group-major order brings the records of one robot group close together in time, and records that share an input set would then
# find it in the caches
mkdir -p gpurun_out/q65536
i=0
for a in "--steps 200 --warmup 20 --input-sets 220 --plan-ways 4" "--steps 200 --warmup 20 --input-sets 220 --plan-queue 1" "--batch 16384 --steps 100 --warmup 100 --input-sets 200 --plan-ways 4" "--batch 16384 --steps 100 --warmup 100 --input-sets 200 --plan-queue 1" "--batch 65536 --steps 100 --warmup 100 --input-sets 200 --plan-ways 4" "--batch 65536 --steps 100 --warmup 100 --input-sets 200 --plan-queue 1"; do
  i=$((i+1))
  timeout -k 10 400 python bench.py --no-cpu-baseline $a > gpurun_out/q65536/d$i.json 2> gpurun_out/q65536/d$i.err || { tail -5 gpurun_out/q65536/d$i.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/q65536/d$i.json") if l.startswith("{")][-1]
r=d["roofline"]
print("$a: value=%.4g in-kernel us/step=%.2f frac=%.3f timed_frac=%.3f golden mism=%s sets=%s" % (d["value"], r["avg_ms_per_step"]*1e3, r["frac"], r["timed_region"]["frac"], d["solved"].get("golden_active_set_mismatches"), d["config"]["input_sets"]), flush=True)
PY
done
