#!/bin/bash
set -o pipefail
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_tick_pipeline.py -m gpu -q -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for a in "kin:" "kin2:" "tables:--tick-tables" "tables2:--tick-tables" "compact:--tick-kin-handoff compact" "kin_k1:--ticks-per-launch 1"; do n=${a%%:*}; x=${a#*:}
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $x > $O/tick_$n.json 2> $O/tick_$n.err || { tail $O/tick_$n.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('$O/tick_$n.json').read().splitlines() if l.startswith('{')][-1]); print('$n', '%.3e' % d['value'], '%.2f us' % (1e3*d['ms_per_step']), d['solved']['ik_fail'])"
done
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so python tools/stamps_tick.py 8192 200 kin | cut -c1-900
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so python tools/stamps_tick.py 8192 200 tables | cut -c1-900
