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
    timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || fail bench $O/bench_driver.err
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
