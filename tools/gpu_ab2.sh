#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; L=gpurun_out/ab2.log; : > $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_tick_pipeline.py -m gpu -q -x >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 600 python tools/stress_ik.py 8000 2>/dev/null | python -c "
import sys, json
bad = 0
for l in sys.stdin:
    d = json.loads(l)
    if d['alg'] == 4: print(d['vmax'], d['form'], 'status_mismatch', d['status_mismatch'], 'set_mismatch', d['set_mismatch'], 'max_dq_diff', d['max_dq_diff'], 'max_iters', d['max_iters'])
" >> $L 2>&1
for v in 100 0.5 0.3; do for B in 4096 65536; do timeout -k 10 100 python tools/time_alg.py $B $v >> $L 2>/dev/null; done; done
grep -v amdgpu.ids $L | tail -20
