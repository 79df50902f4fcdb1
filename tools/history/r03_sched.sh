#!/bin/bash
# hipcc scheduling strategies for the IK / plan / tick kernels (-mllvm -amdgpu-sched-strategy=...), diagnostic builds against the product library
mkdir -p gpurun_out/sched
D=$PWD/walking-controllers_amd/csrc/build/diag
i=0
for a in "--steps 200 --warmup 20" "--steps 20 --warmup 5" "--workload tick --batch 8192 --steps 1000 --warmup 24 --tick-tables" "--workload tick --batch 8192 --steps 1000 --warmup 24"; do
  for v in product ilp mclause minreg product; do
    i=$((i+1))
    lib=""; [ $v != product ] && lib=$D/libwcqp_$v.so
    WCQP_LIB_PATH=$lib timeout -k 10 280 python bench.py --no-cpu-baseline $a > gpurun_out/sched/r$i.json 2> gpurun_out/sched/r$i.err || { tail -5 gpurun_out/sched/r$i.err; exit 1; }
    python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/sched/r$i.json") if l.startswith("{")][-1]
r=d["roofline"]
print("$a [$v]: value=%.4g ms/step=%.5f frac=%.3f golden mism=%s" % (d["value"], d["ms_per_step"], r["frac"], d["solved"].get("golden_active_set_mismatches")), flush=True)
PY
  done
done
