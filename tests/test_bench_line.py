"""The JSON line of bench.py (the driver's contract + what SURVEY.md 8d / VERDICT r3 ask of it), on a small batch so that it runs in
seconds: every object is there, the timed work was checked (goldens for the cold-start batches, the CPU restatement for the ticks)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_default_line_carries_roofline_tick_and_both_cpu_baselines():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "1024", "--tick-batch", "512", "--tick-ticks", "60",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["steps"] == 6 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["repeats"] == 5 and d["value_min"] <= d["value"] <= d["value_max"]
    assert abs(d["value"] - 2 * 1024 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]          # value IS the median region
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "qp_plan_kernel" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    assert rf["algorithmic_bytes_per_launch"] == 6296 * 1024 * 6 and rf["single_launch_form"]["frac"] > 0
    s = d["solved"]
    assert s["ik"] == 1024 and s["mpc"] == 1024 and s["golden_active_set_mismatches"] == 0 and s["golden_max_abs_err"] <= 1e-9
    assert s["golden_rows_checked"] == 6 * 2 * 1024                       # 6 ways (one per step), both QPs, every row
    for name in ("fused_kinematics", "constant_jacobians"):
        t = d["tick"][name]
        assert "error" not in t, t
        assert t["value"] > 0 and t["us_per_tick"] > 0 and t["roofline"]["bound"] == "latency" and t["roofline"]["own_hbm_bytes_per_robot_tick"] > 0
        assert t["solved"]["ticks_executed"] == 60 + 24 and t["solved"]["mpc_fail"] == 0
        oc = t["oracle_check"]
        assert oc["ok"] is True and oc["ticks"] == 16 and len(oc["robots"]) >= 5 and oc["max_abs_err_u0"] <= 1e-9 and oc["max_abs_err_dq"] <= 1e-8, oc
    assert d["tick"]["fused_kinematics"]["solved"]["robots_with_ik_fail"] == 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["same_algorithm_qps"] > cb["value"]
