#!/bin/bash
# One entry point for the GPU-box experiments of a round:   gpurun --timeout N -- 'tools/gpurun.sh <experiment> [args]'
# Every experiment writes under gpurun_out/<round>/<experiment>/ and prints a short digest; summaries that DESIGN.md cites are copied
# into profiles/ by hand (named per round).  Steps are joined so that nothing runs after a GPU step that failed or timed out.
set -o pipefail
R=r04
exp=$1; shift
O=gpurun_out/$R/$exp
mkdir -p $O
fail() { echo "FAILED: $1"; tail -40 "$2" 2>/dev/null; exit 1; }

last_json() { python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no JSON line (%r)" % (e,)); continue
    r = d.get("roofline", {})
    print(f.split("/")[-1], "value %.4g %s, %.3f us/step, frac %.3f, kernel %s" % (d["value"], d["unit"], 1e3 * d["ms_per_step"], r.get("frac", float("nan")), str(r.get("kernel"))[:40]),
          {k: d["solved"].get(k) for k in ("golden_active_set_mismatches", "golden_max_abs_err", "golden_rows_checked", "robots_with_ik_fail") if k in d.get("solved", {})},
          {k: (round(v["value"] / 1e9, 4), round(v["us_per_tick"], 2)) for k, v in d.get("tick", {}).items() if isinstance(v, dict) and "value" in v})
PY
}

case $exp in
suite)          # the GPU suite, smoke, the driver's bench command
    timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 || fail pytest $O/pytest.log
    tail -1 $O/pytest.log
    python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $O/smoke.log
    tail -1 $O/smoke.log
    T0=$SECONDS
    timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || fail bench $O/bench_driver.err
    echo "driver bench command: $((SECONDS - T0)) s wall"
    last_json $O/bench_driver.json
    ;;
bench)          # new bench-related tests, the driver's command, a 2-rank gloo rehearsal
    timeout -k 10 900 python -m pytest tests/test_bench_line.py tests/test_sharding_gloo.py tests/test_robots.py -m gpu -q -x > $O/pytest.log 2>&1 || fail pytest $O/pytest.log
    tail -1 $O/pytest.log
    timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || fail bench $O/bench_driver.err
    last_json $O/bench_driver.json
    WCQP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --tick-ticks 200 > $O/bench_gpus2_gloo_rehearsal.json 2> $O/g2.err || fail "bench gpus 2" $O/g2.err
    last_json $O/bench_gpus2_gloo_rehearsal.json
    ;;
diet)           # after a kernel edit: the GPU suite, the two bench forms, and the dynamic instruction mix of the plan / fused-tick kernels (PMC)
    tag=${1:-cur}
    timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest_$tag.log 2>&1 || fail pytest $O/pytest_$tag.log
    tail -1 $O/pytest_$tag.log
    timeout -k 10 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-tick > $O/bench200_$tag.json 2> $O/bench200_$tag.err || fail bench200 $O/bench200_$tag.err
    timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20_$tag.json 2> $O/bench20_$tag.err || fail bench20 $O/bench20_$tag.err
    last_json $O/bench200_$tag.json $O/bench20_$tag.json
    R0=$PWD; cd /tmp && export TMPDIR=/tmp
    P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
    P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA"
    n=1
    for P in "$P1" "$P2"; do
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc_$tag/plan_4096_p$n -- python3 $R0/bench.py --steps 88 --warmup 88 --repeats 1 --batch 4096 --no-cpu-baseline --no-tick > $R0/$O/pmc_plan_p$n.log 2>&1 < /dev/null || fail "pmc plan $n" $R0/$O/pmc_plan_p$n.log
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc_$tag/tickkin_8192_p$n -- python3 $R0/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $R0/$O/pmc_tickkin_p$n.log 2>&1 < /dev/null || fail "pmc tick $n" $R0/$O/pmc_tickkin_p$n.log
        n=$((n+1))
    done
    cd $R0
    python3 tools/pmc/summarize.py $O/pmc_$tag > $O/pmc_summary_$tag.json 2> $O/pmc_summary_$tag.err
    rm -rf $O/pmc_$tag
    python3 - $O/pmc_summary_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def digest(prefix, kern, units):
    c = {}
    for p in ("p1", "p2"):
        for k, v in d.get("%s_%s" % (prefix, p), {}).get(kern, {}).items():
            c[k] = v["total"]
    if not c: return None
    g = lambda k: c.get(k, 0.0)
    return {"valu": g("SQ_INSTS_VALU") / units, "lds": g("SQ_INSTS_LDS") / units, "salu": g("SQ_INSTS_SALU") / units, "vmem_rd": g("SQ_INSTS_VMEM_RD") / units, "smem": g("SQ_INSTS_SMEM") / units,
            "mfma": g("SQ_INSTS_MFMA") / units, "wave_cycles": 4 * g("SQ_WAVE_CYCLES") / units, "valu_issue_share": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1),
            "any_issue_share": g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), "lds_conflict_share": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)}
# bench.py --steps 88 --warmup 88 --repeats 1: the warm-up plan, the timed plan and the roofline pass's 4 launches, 88 records x 1024 robot groups each
launches = d.get("plan_4096_p1", {}).get("qp_plan_kernel", {}).get("SQ_WAVES", {}).get("launches", 0)
print("qp_plan_kernel per wave-record:", json.dumps(digest("plan_4096", "qp_plan_kernel", max(1, launches) * 88 * 1024)))
print("ik4 fused tick per wave-tick:", json.dumps(digest("tickkin_8192", "ik4_tick_kernel", 224 * 2048)))
PY
    ;;
pipe)           # record-ahead Jacobian loads in the plan kernel, A/B against the same walk without them (WCQP_PLAN_NO_PIPE). NEEDS profiles/r04_record_ahead.patch
                # applied to csrc/ik4.hip (git apply) and a rebuild: the experiment was negative and the code is not in the tree (profiles/r04_record_ahead_ab.txt)
    grep -q WCQP_PLAN_NO_PIPE walking-controllers_amd/csrc/*.hip || { echo 'pipe: check out csrc of commit dd95896, apply profiles/r04_record_ahead.patch and rebuild first'; exit 1; }
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_line.py -m gpu -q -x > $O/pytest.log 2>&1 || fail pytest $O/pytest.log
    tail -1 $O/pytest.log
    for rep in 1 2; do
      for v in pipe nopipe; do
        [ $v = nopipe ] && export WCQP_PLAN_NO_PIPE=1 || unset WCQP_PLAN_NO_PIPE
        timeout -k 10 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-tick > $O/bench200_${v}_$rep.json 2> $O/bench200_${v}_$rep.err || fail "bench200 $v" $O/bench200_${v}_$rep.err
        timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tick > $O/bench20_${v}_$rep.json 2> $O/bench20_${v}_$rep.err || fail "bench20 $v" $O/bench20_${v}_$rep.err
        timeout -k 10 400 python bench.py --steps 50 --warmup 10 --batch 65536 --no-cpu-baseline --no-tick > $O/bench65536_${v}_$rep.json 2> $O/bench65536_${v}_$rep.err || fail "bench65536 $v" $O/bench65536_${v}_$rep.err
        last_json $O/bench200_${v}_$rep.json $O/bench20_${v}_$rep.json $O/bench65536_${v}_$rep.json
      done
    done
    unset WCQP_PLAN_NO_PIPE
    ;;
twopass)        # single-batch latency form: one IK launch against a flagging pass + a compacted pass over the flagged robots (best case)
    timeout -k 10 300 python tools/two_pass_timing.py > $O/two_pass.json 2> $O/two_pass.err || fail twopass $O/two_pass.err
    cat $O/two_pass.json
    ;;
profiles)       # end-of-round evidence -> gpurun_out/r04/profiles/ (tools/r04_collect.py copies the summaries into profiles/r04_*)
    R0=$PWD; cd /tmp && export TMPDIR=/tmp
    prof() { # name, bench args...
        n=$1; shift
        timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R0/$O/prof_$n -- python3 $R0/bench.py --no-cpu-baseline "$@" > $R0/$O/prof_$n.log 2>&1 < /dev/null || fail "prof $n" $R0/$O/prof_$n.log
        f=$(ls $R0/$O/prof_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R0/$O/${n}_kernel_stats.csv
        rm -rf $R0/$O/prof_$n
        echo "profiled $n"
    }
    # (warm-up = steps: the warm-up steps are a plan launch of their own, and rocprofv3's average over the launches of the plan kernel is then over launches of ONE length)
    prof bench_b4096_driver --steps 20 --warmup 20
    prof bench_b4096 --steps 200 --warmup 200 --no-tick
    prof bench_b65536 --steps 50 --warmup 50 --batch 65536 --no-tick
    prof tick_kin_b8192 --workload tick --batch 8192 --steps 1000 --warmup 24
    prof tick_tables_b8192 --workload tick --batch 8192 --steps 1000 --warmup 24 --tick-tables
    # PMC: HBM traffic (FETCH_SIZE to be doubled on gfx950: MI355X_MICROARCH.md), one counter per pass; every plan-kernel launch of a run holds 88 / 24 records
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R0/$O/pmc/bench_4096_$C -- python3 $R0/bench.py --steps 88 --warmup 88 --repeats 1 --batch 4096 --no-cpu-baseline --no-tick > $R0/$O/pmc_bench_4096_$C.log 2>&1 < /dev/null || fail "pmc $C 4096" $R0/$O/pmc_bench_4096_$C.log
        timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R0/$O/pmc/bench_65536_$C -- python3 $R0/bench.py --steps 24 --warmup 24 --repeats 1 --batch 65536 --no-cpu-baseline --no-tick > $R0/$O/pmc_bench_65536_$C.log 2>&1 < /dev/null || fail "pmc $C 65536" $R0/$O/pmc_bench_65536_$C.log
        timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R0/$O/pmc/tickkin_8192_$C -- python3 $R0/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $R0/$O/pmc_tickkin_$C.log 2>&1 < /dev/null || fail "pmc $C tick" $R0/$O/pmc_tickkin_$C.log
        timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R0/$O/pmc/ticktab_8192_$C -- python3 $R0/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --tick-tables --no-cpu-baseline > $R0/$O/pmc_ticktab_$C.log 2>&1 < /dev/null || fail "pmc $C ticktab" $R0/$O/pmc_ticktab_$C.log
        echo "pmc $C"
    done
    P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
    P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA"
    n=1
    for P in "$P1" "$P2"; do
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc/plan_4096_p$n -- python3 $R0/bench.py --steps 88 --warmup 88 --repeats 1 --batch 4096 --no-cpu-baseline --no-tick > $R0/$O/pmc_plan_p$n.log 2>&1 < /dev/null || fail "pmc plan $n" $R0/$O/pmc_plan_p$n.log
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc/tickkin_8192_p$n -- python3 $R0/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $R0/$O/pmc_tickkin_p$n.log 2>&1 < /dev/null || fail "pmc tick $n" $R0/$O/pmc_tickkin_p$n.log
        n=$((n+1))
    done
    cd $R0
    python3 tools/pmc/summarize.py $O/pmc > $O/pmc_summary.json 2> $O/pmc_summary.err
    python3 tools/pmc/make_traffic.py $O/pmc_summary.json > $O/traffic.json 2> $O/traffic.err
    # the bench lines below quote roofline.traffic from profiles/traffic.json when its source hash is the current one: this run's own PMC passes
    [ -s $O/traffic.json ] && cp $O/traffic.json profiles/traffic.json
    rm -rf $O/pmc
    ;;
benchlines)     # the round's bench lines, no profiler attached (after `profiles`, with its traffic.json copied into profiles/ by tools/r04_collect.py)
    O=gpurun_out/$R/profiles; mkdir -p $O
    # bench lines (no profiler attached)
    timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_b4096_driver.json 2> $O/bench.err || fail "bench driver" $O/bench.err
    timeout -k 10 600 python3 bench.py --steps 200 --warmup 20 --no-tick > $O/bench_b4096.json 2>> $O/bench.err || fail "bench 200" $O/bench.err
    timeout -k 10 600 python3 bench.py --steps 100 --warmup 10 --batch 65536 --no-cpu-baseline --no-tick > $O/bench_b65536.json 2>> $O/bench.err || fail "bench 65536" $O/bench.err
    timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tick --plan-ways 0 > $O/bench_b4096_driver_launch_per_step.json 2>> $O/bench.err || fail "bench lps" $O/bench.err
    timeout -k 10 600 python3 bench.py --steps 200 --warmup 20 --ik-form osqp --no-cpu-baseline --no-tick > $O/bench_b4096_osqp.json 2>> $O/bench.err || fail "bench osqp" $O/bench.err
    timeout -k 10 600 python3 bench.py --steps 200 --warmup 20 --horizon 200 --no-cpu-baseline --no-tick > $O/bench_b4096_n200.json 2>> $O/bench.err || fail "bench n200" $O/bench.err
    timeout -k 10 600 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b8192.json 2>> $O/bench.err || fail "bench tick" $O/bench.err
    timeout -k 10 600 python3 bench.py --workload tick --batch 8192 --steps 1000 --warmup 24 --no-cpu-baseline --tick-tables > $O/bench_tick_tables_b8192.json 2>> $O/bench.err || fail "bench ticktab" $O/bench.err
    timeout -k 10 600 python3 bench.py --workload tick --batch 65536 --steps 200 --warmup 24 --no-cpu-baseline > $O/bench_tick_kin_b65536.json 2>> $O/bench.err || fail "bench tick 65536" $O/bench.err
    WCQP_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --tick-ticks 200 > $O/bench_gpus2_gloo_rehearsal.json 2>> $O/bench.err || fail "bench gpus 2" $O/bench.err
    last_json $O/bench_*.json
    ;;
stage)          # CoM / neck Jacobians through LDS a record ahead: semantics of the instruction, parity of the plan forms, A/B against -DWCQP_PLAN_NO_STAGE
                # NEEDS profiles/r04_lds_stage.patch applied (git apply) and both libraries rebuilt: the experiment was negative, the code is not in the tree
    grep -q WCQP_PLAN_NO_STAGE walking-controllers_amd/csrc/ik4.hip || { echo 'stage: check out csrc of commit 620ab08, apply profiles/r04_lds_stage.patch, rebuild, tools/build_variant.sh nostage -DWCQP_PLAN_NO_STAGE first'; exit 1; }
    hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_dma_test.hip -o /tmp/lds_dma_test 2> $O/ubench_build.err || fail "ubench build" $O/ubench_build.err
    timeout -k 5 60 /tmp/lds_dma_test > $O/lds_dma_test.txt 2>&1 || fail "lds_dma_test" $O/lds_dma_test.txt
    tail -1 $O/lds_dma_test.txt
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_line.py -m gpu -q -x > $O/pytest.log 2>&1 || fail pytest $O/pytest.log
    tail -1 $O/pytest.log
    "$0" ab product nostage
    ;;
fuzz)           # both fuzz tools on the final kernels, on instances no earlier round has seen (seed offset 4000; robot groups 5..8), all three parameter sets
    timeout -k 10 1000 python tools/fuzz_vs_oracle.py 42 256 4000 > $O/fuzz_vs_oracle.jsonl 2> $O/fuzz.err || fail fuzz $O/fuzz.err
    tail -1 $O/fuzz_vs_oracle.jsonl
    timeout -k 10 1000 python tools/fuzz_tick_vs_oracle.py 4 24 300 5 > $O/fuzz_tick_vs_oracle.jsonl 2> $O/fuzz_tick.err || fail fuzz_tick $O/fuzz_tick.err
    tail -1 $O/fuzz_tick_vs_oracle.jsonl
    ;;
tstamps)        # the tick kernel by phase, both forms (library: tools/build_variant.sh tstamps -DWCQP_TICK_STAMPS)
    for m in kin tables; do
      WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so timeout -k 10 300 python tools/stamps_tick.py 8192 200 $m > $O/tstamps_$m.json 2> $O/tstamps_$m.err || fail "tstamps $m" $O/tstamps_$m.err
      cat $O/tstamps_$m.json
    done
    ;;
kstamps)        # the kinematics phase of the fused tick by sub-phase (s_memtime; library built with tools/build_variant.sh kstamps -DWCQP_TICK_KSTAMPS -DWCQP_TICK_STAMPS)
    WCQP_KSTAMPS=1 WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_kstamps.so timeout -k 10 300 python tools/stamps_tick.py 8192 200 kin > $O/kstamps.json 2> $O/kstamps.err || fail kstamps $O/kstamps.err
    cat $O/kstamps.json
    ;;
xcd)            # XCD-aware robot-group order of the plan kernel against the plain order (-DWCQP_PLAN_NO_XCD_MAP): parity, three bench forms, HBM fetch bytes
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bench_line.py -m gpu -q -x > $O/pytest.log 2>&1 || fail pytest $O/pytest.log
    tail -1 $O/pytest.log
    "$0" ab product noxcd || exit 1
    R0=$PWD; cd /tmp && export TMPDIR=/tmp
    for lib in product noxcd; do
        L=$R0/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so; [ $lib = product ] && L=$R0/walking-controllers_amd/libwcqp.so
        export WCQP_LIB_PATH=$L
        for bs in "4096 88" "65536 24"; do
            read b st <<< "$bs"
            timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R0/$O/pmc_$lib/bench_${b}_FETCH_SIZE -- python3 $R0/bench.py --steps $st --warmup $st --repeats 1 --batch $b --no-cpu-baseline --no-tick > $R0/$O/pmc_${lib}_$b.log 2>&1 < /dev/null || fail "pmc $lib $b" $R0/$O/pmc_${lib}_$b.log
        done
        unset WCQP_LIB_PATH
        ( cd $R0 && python3 tools/pmc/summarize.py $O/pmc_$lib > $O/pmc_summary_$lib.json 2> $O/pmc_summary_$lib.err; rm -rf $O/pmc_$lib )
    done
    cd $R0
    python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for lib in ("product", "noxcd"):
    d = json.load(open("%s/pmc_summary_%s.json" % (O, lib)))
    for run, ks in d.items():
        k = ks.get("qp_plan_kernel", {}).get("FETCH_SIZE")
        if k: print(lib, run, "FETCH_SIZE mean per launch %.6g (KiB, to be doubled on gfx950: tools/pmc/make_traffic.py) over %d launches" % (k["mean_per_launch"], k["launches"]))
PY
    ;;
queue)          # the work-queue form of the plan with fewer steal attempts at the exit (-DWCQP_PLAN_STEAL=n) against the fixed ways, driver form and 200 steps
    libs=("$@")
    for rep in 1 2; do
      for lib in "${libs[@]}"; do
        L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so; [ "$lib" = product ] && L=
        for q in 0 1; do
          for cfg in "20 5" "200 20"; do
            read st w <<< "$cfg"
            WCQP_LIB_PATH=$L timeout -k 10 400 python bench.py --steps $st --warmup $w --plan-queue $q --no-cpu-baseline --no-tick > $O/${lib}_q${q}_s${st}_$rep.json 2> $O/${lib}_q${q}_s${st}_$rep.err || fail "queue $lib $q $cfg" $O/${lib}_q${q}_s${st}_$rep.err
            echo -n "$lib queue=$q rep $rep: "; last_json $O/${lib}_q${q}_s${st}_$rep.json
          done
        done
      done
    done
    ;;
pmc2)           # instruction-cache and LDS-stall counters of the plan kernel and the fused tick (one group per run, --kernel-trace only)
    R0=$PWD; cd /tmp && export TMPDIR=/tmp
    P1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
    P2="SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
    P3="SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
    n=1
    for P in "$P1" "$P2" "$P3"; do
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc/plan_4096_p$n -- python3 $R0/bench.py --steps 88 --warmup 88 --repeats 1 --batch 4096 --no-cpu-baseline --no-tick > $R0/$O/pmc_plan_p$n.log 2>&1 < /dev/null || fail "pmc plan $n" $R0/$O/pmc_plan_p$n.log
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R0/$O/pmc/tickkin_8192_p$n -- python3 $R0/bench.py --workload tick --batch 8192 --steps 200 --warmup 24 --no-cpu-baseline > $R0/$O/pmc_tickkin_p$n.log 2>&1 < /dev/null || fail "pmc tick $n" $R0/$O/pmc_tickkin_p$n.log
        n=$((n+1))
    done
    cd $R0
    python3 tools/pmc/summarize.py $O/pmc > $O/pmc2_summary.json 2> $O/pmc2_summary.err
    rm -rf $O/pmc
    python3 - $O/pmc2_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for run, ks in sorted(d.items()):
    for k, c in ks.items():
        if k in ("qp_plan_kernel", "ik4_tick_kernel"):
            print(run, k, {n: "%.4g" % v["total"] for n, v in c.items()})
PY
    ;;
ab)             # the three bench forms of the plan kernel for each library variant given ("product" = the tree's library), twice, interleaved
    libs=("$@")
    for rep in 1 2; do
      for lib in "${libs[@]}"; do
        tag=$lib; L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so
        [ "$lib" = product ] && L=
        for cfg in "4096 20 5" "4096 200 20" "65536 50 10"; do
          read b st w <<< "$cfg"
          WCQP_LIB_PATH=$L timeout -k 10 400 python bench.py --batch $b --steps $st --warmup $w --no-cpu-baseline --no-tick > $O/${tag}_b${b}_s${st}_$rep.json 2> $O/${tag}_b${b}_s${st}_$rep.err || fail "ab $tag $cfg" $O/${tag}_b${b}_s${st}_$rep.err
          echo -n "$tag rep $rep: "; last_json $O/${tag}_b${b}_s${st}_$rep.json
        done
      done
    done
    ;;
split)          # one combined plan against IK-only + MPC-only plans enqueued together, for the libraries given (product = "")
    for lib in "$@"; do
        tag=${lib:-product}
        [ "$lib" = product ] && lib=
        for cfg in "4096 20" "4096 200" "65536 50"; do set -- $cfg
            WCQP_LIB_PATH=${lib:+$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_$lib.so} timeout -k 10 300 python tools/split_plan_timing.py $1 $2 > $O/${tag}_b$1_s$2.json 2> $O/${tag}_b$1_s$2.err || fail "split $tag $cfg" $O/${tag}_b$1_s$2.err
            python3 - $O/${tag}_b$1_s$2.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "B=%d steps=%d" % (d["batch"], d["steps"]), {k: (round(v["us_per_step_median"], 2), "%.4g" % v["qp_per_s_median"]) for k, v in d.items() if isinstance(v, dict)})
PY
        done
    done
    ;;
*)
    echo "unknown experiment $exp"; exit 2 ;;
esac
