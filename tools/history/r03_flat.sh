#!/bin/bash
mkdir -p gpurun_out/queue
timeout -k 10 280 python tests/helpers/plan_check.py || exit 1
timeout -k 10 600 python -m pytest tests/test_tick_pipeline.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for cfg in "4096 200 0" "4096 200 1" "4096 20 0" "4096 20 1" "65536 100 0" "65536 100 1"; do
  set -- $cfg
  timeout -k 10 250 python bench.py --batch $1 --steps $2 --warmup 5 --plan-queue $3 --no-cpu-baseline > gpurun_out/queue/b$1_s$2_q$3.json 2> gpurun_out/queue/b$1_s$2_q$3.err || { tail -5 gpurun_out/queue/b$1_s$2_q$3.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/queue/b$1_s$2_q$3.json") if l.startswith("{")][-1]
r=d["roofline"]
print("B=$1 steps=$2 queue=$3 value=%.4g in-kernel us/step=%.2f ns/robot=%.3f frac=%.3f timed_frac=%.3f golden=%s/%s rows=%s" % (d["value"], r["avg_ms_per_step"]*1e3, r["avg_ms_per_step"]*1e6/$1, r["frac"], r["timed_region"]["frac"], d["solved"].get("golden_max_abs_err"), d["solved"].get("golden_active_set_mismatches"), d["solved"].get("golden_rows_checked")), flush=True)
PY
done
for a in "" "--tick-tables"; do
  timeout -k 10 250 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $a > gpurun_out/queue/tick$a.json 2> gpurun_out/queue/tick$a.err || { tail -5 gpurun_out/queue/tick$a.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/queue/tick$a.json") if l.startswith("{")][-1]
print("tick $a value=%.4g ms/step=%.5f" % (d["value"], d["ms_per_step"]), flush=True)
PY
done
