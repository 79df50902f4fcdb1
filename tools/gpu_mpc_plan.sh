#!/bin/bash
# BASELINE configs 2 and 3 on their own: MPC-only and IK-only plans against a launch per batch (+ the plan parity check)
mkdir -p gpurun_out/mpcplan
timeout -k 10 400 python tests/helpers/plan_check.py || exit 1
for cfg in "4096 400" "65536 100"; do set -- $cfg
timeout -k 10 300 python tools/mpc_plan_timing.py $1 $2 > gpurun_out/mpcplan/b$1.json 2> gpurun_out/mpcplan/b$1.err || { tail -5 gpurun_out/mpcplan/b$1.err; exit 1; }
timeout -k 10 300 python tools/ik_plan_timing.py $1 $(( $2 / 2 )) > gpurun_out/mpcplan/ik_b$1.json 2> gpurun_out/mpcplan/ik_b$1.err || { tail -5 gpurun_out/mpcplan/ik_b$1.err; exit 1; }
python - <<PY
import json
for f in ("b$1", "ik_b$1"):
    d=json.load(open("gpurun_out/mpcplan/%s.json" % f))
    print(f, "sets", d["input_sets"], {k: (round(v["us_per_batch"],2), "%.3g" % (v.get("mpc_qp_per_s") or v.get("ik_qp_per_s")), round(v["hbm_frac"],3)) for k,v in d.items() if isinstance(v, dict)})
PY
done
for a in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
timeout -k 10 300 python bench.py --no-cpu-baseline $a > gpurun_out/mpcplan/bench.json 2> gpurun_out/mpcplan/bench.err || { tail -5 gpurun_out/mpcplan/bench.err; exit 1; }
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/mpcplan/bench.json") if l.startswith("{")][-1]
r=d["roofline"]
print("bench $a: value=%.4g in-kernel us/step=%.2f frac=%.3f golden mism=%s" % (d["value"], r["avg_ms_per_step"]*1e3, r["frac"], d["solved"].get("golden_active_set_mismatches")))
PY
done
