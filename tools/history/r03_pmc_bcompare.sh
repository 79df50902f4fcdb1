#!/bin/bash
# why does the plan kernel spend 1.69 ns per robot-step at 4096 robots and 1.31 at 32768?  PMC passes at both sizes.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/bcmp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 < /dev/null
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU"
P2="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
P4="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RD_UNCACHED_32B_sum"
P5="TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"
n=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  for cfg in "4096 88" "32768 24"; do
    set -- $cfg
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc/plan_$1_p$n -- python3 $R/bench.py --steps $2 --warmup 5 --batch $1 --no-cpu-baseline > $O/pmc_$1_p$n.log 2>&1 < /dev/null || tail -3 $O/pmc_$1_p$n.log
  done
  echo "pass $n" 
  n=$((n+1))
done
cd $R
python3 tools/pmc/summarize.py $O/pmc > $O/pmc_summary.json 2> $O/pmc_summary.err
python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
for k in sorted(d):
    v=d[k].get("qp_plan_kernel")
    if v: print(k, {c:(x["total"], x["launches"]) for c,x in v.items()})
PY
