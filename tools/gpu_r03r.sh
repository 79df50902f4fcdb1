#!/bin/bash
set -o pipefail
O=gpurun_out/r03r
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for a in "kin:" "tables:--tick-tables"; do n=${a%%:*}; x=${a#*:}
timeout -k 10 300 python bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline $x > $O/tick_$n.json 2> $O/tick_$n.err || { tail $O/tick_$n.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/tick_$n.json').read().strip().splitlines()[-1]); print('$n', '%.3e' % d['value'], '%.2f us' % (1e3*d['ms_per_step']))"
done
