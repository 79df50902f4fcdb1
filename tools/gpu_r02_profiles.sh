#!/bin/bash
# Round-2 profiles: rocprofv3 kernel stats of the bench commands, PMC passes (separate runs, --kernel-trace only), bench
# lines.  Everything under gpurun_out/r02/; the summaries worth keeping are copied to profiles/ by hand.
#   gpurun --timeout 1100 -- 'bash tools/gpu_r02_profiles.sh'
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, bench args...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1 || tail -3 $O/prof_$n.log
  f=$(ls $O/prof_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  echo "profiled $n"
}
prof bench_b4096 --steps 200 --warmup 20
prof bench_b4096_driver --steps 20 --warmup 5
prof bench_b4096_two_launches --steps 200 --warmup 20 --streams 2 --pipelines 1
prof bench_b4096_one_batch_at_a_time --steps 200 --warmup 20 --pipelines 1
prof bench_b65536 --steps 50 --warmup 10 --batch 65536
prof bench_b4096_osqp --steps 200 --warmup 20 --ik-form osqp
prof bench_b4096_n200 --steps 200 --warmup 20 --horizon 200
prof tick_kin_b8192 --workload tick --batch 8192 --steps 300 --warmup 24 --streams 1
prof tick_tables_b8192 --workload tick --batch 8192 --steps 300 --warmup 24 --streams 1 --tick-tables
prof kin_b65536 --workload kin --batch 65536 --steps 50 --warmup 10
# PMC: HBM traffic (FETCH_SIZE to be doubled on gfx950: MI355X_MICROARCH.md), one counter per pass
for C in FETCH_SIZE WRITE_SIZE; do
  for B in 4096 65536; do
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/bench_${B}_$C -- python3 $R/bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline > $O/pmc_bench_${B}_$C.log 2>&1
  done
  echo "pmc $C"
done
# PMC: the IK kernel alone at 65536 (instruction mix, LDS, MFMA busy)
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
n=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_ik/ik4_p$n -- python3 $R/tools/pmc/ik_only.py 65536 5 > $O/pmc_ik4_p$n.log 2>&1 || tail -3 $O/pmc_ik4_p$n.log
  n=$((n+1))
done
cd $R
python3 tools/pmc/summarize.py $O/pmc > $O/pmc_summary.json 2>&1
python3 tools/pmc/make_traffic.py $O/pmc_summary.json > $O/traffic.json 2>&1
python3 tools/pmc/summarize_ik.py $O/pmc_ik > $O/pmc_ik4_detail.json 2>&1
# bench lines (no profiler attached)
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 > $O/bench_b4096.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b4096_driver.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --ik-jac auto > $O/bench_b4096_driver_auto.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --pipelines 1 > $O/bench_b4096_one_batch_at_a_time.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --pipelines 1 --streams 2 > $O/bench_b4096_two_launches.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline > $O/bench_b65536.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --ik-form osqp --no-cpu-baseline > $O/bench_b4096_osqp.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --horizon 200 --no-cpu-baseline > $O/bench_b4096_n200.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-tables > $O/bench_tick_tables_b8192.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload tick --batch 65536 --steps 200 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b65536.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --workload kin --batch 65536 --steps 50 --warmup 10 > $O/bench_kin_b65536.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$(basename $f): $(cut -c1-230 $f)"; done
