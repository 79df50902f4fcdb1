#!/bin/bash
mkdir -p gpurun_out/sc
timeout -k 10 900 python -m pytest tests/test_kinematics.py tests/test_tick_pipeline.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for a in "" "--tick-tables"; do
  timeout -k 10 250 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $a > gpurun_out/sc/tick$a.json 2> gpurun_out/sc/tick$a.err || { tail -5 gpurun_out/sc/tick$a.err; exit 1; }
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/sc/tick$a.json") if l.startswith("{")][-1]
print("tick $a value=%.4g ms/step=%.5f" % (d["value"], d["ms_per_step"]), flush=True)
PY
done
