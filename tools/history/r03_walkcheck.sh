#!/bin/bash
# the shipped walk scenario on the final kernels: 65536 robots x 1024 closed-loop ticks (bench line) + tools/walk_diag.py at 65536
mkdir -p gpurun_out/walkcheck
timeout -k 10 500 python bench.py --workload tick --batch 65536 --steps 1000 --warmup 24 --no-cpu-baseline > gpurun_out/walkcheck/bench_tick_kin_b65536_1024ticks.json 2> gpurun_out/walkcheck/b.err || { tail -5 gpurun_out/walkcheck/b.err; exit 1; }
python - <<'PY'
import json
d=[json.loads(l) for l in open("gpurun_out/walkcheck/bench_tick_kin_b65536_1024ticks.json") if l.startswith("{")][-1]
print("tick 65536 x 1024: value %.4g ms/tick %.5f solved %s" % (d["value"], d["ms_per_step"], json.dumps(d["solved"])[:600]), flush=True)
PY
timeout -k 10 600 python tools/walk_diag.py --big 65536 > gpurun_out/walkcheck/walk_diag.jsonl 2> gpurun_out/walkcheck/w.err || { tail -5 gpurun_out/walkcheck/w.err; exit 1; }
cut -c1-420 gpurun_out/walkcheck/walk_diag.jsonl
