#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/kin2.log; : > $L
for lib in kp2 kp3; do
  [ -n "$lib" ] && export WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so
  echo "== lib ${lib:-product}" >> $L
  timeout -k 10 300 python bench.py --workload kin --batch 65536 --steps 50 --warmup 5 >> $L 2>&1 || { tail -20 $L; exit 1; }
  timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 200 --warmup 20 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline'].get('frac'))
"
