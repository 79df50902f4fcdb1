#!/bin/bash
# does ONE plan launch fill the card?  two independent bench processes on the same GPU at the same time vs one
O=gpurun_out/r03v
mkdir -p $O
python3 -c "
import torch
p=torch.cuda.get_device_properties(0); print('CUs', p.multi_processor_count, p.name, 'mem GB', p.total_memory/2**30)"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 20 > $O/solo.json 2> $O/solo.err
(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 20 > $O/pair_a.json 2> $O/pair_a.err &
 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 20 > $O/pair_b.json 2> $O/pair_b.err &
 wait)
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 20 --plan-ways 8 > $O/solo_w8.json 2> $O/solo_w8.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 20 --batch 8192 > $O/solo_b8192.json 2> $O/solo_b8192.err
python3 - <<'PY'
import json
for f in ("solo", "pair_a", "pair_b", "solo_w8", "solo_b8192"):
    try:
        d = json.loads([l for l in open("gpurun_out/r03v/%s.json" % f).read().splitlines() if l.startswith("{")][-1])
        print(f, "value %.3e us/step %.2f" % (d["value"], 1e3 * d["ms_per_step"]))
    except Exception as e:
        print(f, "failed", e)
PY
