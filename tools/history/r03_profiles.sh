#!/bin/bash
# Round-3 evidence: rocprofv3 kernel stats of the bench commands, PMC passes (separate runs, --kernel-trace only), bench lines.
# Everything under gpurun_out/r03/; the summaries worth keeping are copied to profiles/r03_* by tools/r03_collect.py.
#   gpurun --timeout 1100 -- 'bash tools/history/r03_profiles.sh'
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, bench args...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1 < /dev/null || tail -3 $O/prof_$n.log
  f=$(ls $O/prof_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/prof_$n
  echo "profiled $n"
}
# (warm-up = steps: the warm-up steps are a plan launch of their own, and rocprofv3's average over the launches of qp_plan_kernel is then over launches of ONE length)
prof bench_b4096_driver --steps 20 --warmup 20
prof bench_b4096 --steps 200 --warmup 200
prof bench_b4096_launch_per_step --steps 200 --warmup 20 --plan-ways 0
prof bench_b65536 --steps 50 --warmup 50 --batch 65536
prof tick_kin_b8192 --workload tick --batch 8192 --steps 1000 --warmup 24
prof tick_kin_compact_b8192 --workload tick --batch 8192 --steps 1000 --warmup 24 --tick-kin-handoff compact --streams 1
prof tick_tables_b8192 --workload tick --batch 8192 --steps 1000 --warmup 24 --tick-tables
# PMC: HBM traffic (FETCH_SIZE to be doubled on gfx950: MI355X_MICROARCH.md), one counter per pass.  Step counts chosen so that every
# qp_plan_kernel launch of a run holds the same number of records (88 at 4096 robots, 24 at 65536)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/bench_4096_$C -- python3 $R/bench.py --steps 88 --warmup 88 --batch 4096 --no-cpu-baseline > $O/pmc_bench_4096_$C.log 2>&1 < /dev/null
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/bench_65536_$C -- python3 $R/bench.py --steps 24 --warmup 24 --batch 65536 --no-cpu-baseline > $O/pmc_bench_65536_$C.log 2>&1 < /dev/null
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/tickkin_8192_$C -- python3 $R/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $O/pmc_tickkin_$C.log 2>&1 < /dev/null
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/ticktab_8192_$C -- python3 $R/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --tick-tables --no-cpu-baseline > $O/pmc_ticktab_$C.log 2>&1 < /dev/null
  echo "pmc $C"
done
# PMC: instruction mix / pipe activity of the plan kernel (B = 4096, 88 records per launch) and of the fused tick kernel
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
n=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc/plan_4096_p$n -- python3 $R/bench.py --steps 88 --warmup 88 --batch 4096 --no-cpu-baseline > $O/pmc_plan_p$n.log 2>&1 < /dev/null || tail -3 $O/pmc_plan_p$n.log
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc/tickkin_8192_p$n -- python3 $R/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $O/pmc_tickkin_p$n.log 2>&1 < /dev/null || tail -3 $O/pmc_tickkin_p$n.log
  n=$((n+1))
done
echo "pmc detail"
cd $R
python3 tools/pmc/summarize.py $O/pmc > $O/pmc_summary.json 2> $O/pmc_summary.err
python3 tools/pmc/make_traffic.py $O/pmc_summary.json > $O/traffic.json 2> $O/traffic.err
# the bench lines below quote roofline.traffic from profiles/traffic.json when its source hash is the current one: this run's own PMC passes
[ -s $O/traffic.json ] && cp $O/traffic.json $R/profiles/traffic.json
# (the raw per-dispatch counter CSVs are tens of MB with one input set per step: only the summaries travel back)
rm -rf $O/pmc
# bench lines (no profiler attached)
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 > $O/bench_b4096.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_b4096_driver.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --plan-ways 0 > $O/bench_b4096_driver_launch_per_step.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --plan-ways 0 > $O/bench_b4096_launch_per_step.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --plan-ways 2 > $O/bench_b4096_ways2.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --plan-ways 4 > $O/bench_b4096_ways4.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --plan-ways 4 > $O/bench_b4096_driver_ways4.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > $O/bench_b65536.json 2>> $O/bench.err
# (the card takes tens of ms of sustained load to reach its steady state at this batch size: the same line behind 100 and 300 warm-up steps)
timeout -k 10 300 python3 bench.py --steps 100 --warmup 100 --batch 65536 --no-cpu-baseline > $O/bench_b65536_warmup100.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 300 --warmup 300 --batch 65536 --no-cpu-baseline > $O/bench_b65536_warmup300_steps300.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --plan-queue 1 > $O/bench_b4096_plan_queue.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --ik-form osqp --no-cpu-baseline > $O/bench_b4096_osqp.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --horizon 200 --no-cpu-baseline > $O/bench_b4096_n200.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --ticks-per-launch 1 > $O/bench_tick_kin_b8192_one_tick_per_launch.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-kin-handoff compact > $O/bench_tick_kin_compact_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-kin-handoff dense > $O/bench_tick_kin_dense_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-tables > $O/bench_tick_tables_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-tables --ticks-per-launch 1 > $O/bench_tick_tables_b8192_one_tick_per_launch.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 65536 --steps 200 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b65536.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 65536 --steps 200 --warmup 24 --no-cpu-baseline --tick-tables > $O/bench_tick_tables_b65536.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload kin --batch 65536 --steps 50 --warmup 10 > $O/bench_kin_b65536.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
