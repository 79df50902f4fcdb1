#!/bin/bash
# time the product library and named diagnostic variants (IK kernel alone) at several bound tightnesses
mkdir -p gpurun_out; L=gpurun_out/ab3.log; : > $L
D=$PWD/walking-controllers_amd/csrc/build/diag
for lib in "" "$@"; do
  for v in 0.5 0.3; do for B in 4096 65536; do
    if [ -n "$lib" ]; then WCQP_LIB_PATH=$D/libwcqp_$lib.so timeout -k 10 100 python tools/time_alg.py $B $v >> $L 2>/dev/null; else timeout -k 10 100 python tools/time_alg.py $B $v >> $L 2>/dev/null; fi
  done; done
done
python - <<'PY'
import json
for l in open('gpurun_out/ab3.log'):
    if l.startswith('{'):
        d = json.loads(l); print(d['lib'], d['B'], 'alg4 ms', round(d['ms']['4'], 5))
PY
