#!/bin/bash
# tick pipeline: its parity tests, then the tick bench with and without the IK hot start (constant Jacobians / per-tick kinematics)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/tick.log
timeout -k 10 900 python -m pytest tests/test_tick_pipeline.py tests/test_gpu_parity.py -m gpu -q -x > $L 2>&1 || { tail -40 $L; exit 1; }
tail -1 $L
for extra in "--tick-tables" "--tick-tables --tick-cold-ik" "" "--tick-cold-ik"; do
  echo "== tick 8192 $extra" >> $L
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $extra >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l)
    elif l.startswith('{'):
        d=json.loads(l); print('  value %.3e ms/tick %.4f' % (d['value'], d['ms_per_step']), {k:v for k,v in d.get('tick',{}).items() if 'hot' in k or 'fail' in k})
"
