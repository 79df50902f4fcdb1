#!/usr/bin/env python3
"""Fuzz of the device solvers against the exact oracle (oracle/qp_spec.py) over fresh seeds - more instances and more parameter
corners than the GPU suite has time for: IK (both forms, velocity limits from 0.15 to 1.0 rad/s, every kernel that serves the
default path and its fall-backs), MPC (horizons 7 ... 200, disturbed states, every contact configuration).  Prints one JSON line per
case and a summary; exits non-zero on the first violation of the suite's parity definition: |x - x*| <= 1e-9, active
sets bit-exact where the strict-complementarity margin exceeds 1e-7, oracle-infeasible <=> WCQP_STATUS_INFEASIBLE.
   python tools/fuzz_vs_oracle.py [n_seeds] [instances per case] [seed offset]
Round 4: the cases rotate through the three parameter sets the reference ships (tests/robots.py: weights, gains, postures, additional
rotations, Q / R, CoM heights, hull tolerances) instead of one; a seed offset gives every round instances no earlier run has seen."""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import walking_controllers_amd as wca
from oracle import qp_spec as qs
SOL_TOL, MARGIN = 1e-9, 1e-7
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
OFF = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sys.path.insert(0, os.path.join(R, "tests"))
import robots as rb
tot = dict(ik_checked=0, ik_infeasible=0, ik_active_sets_compared=0, ik_with_active_bounds=0, mpc_checked=0, mpc_active_sets_compared=0, mpc_with_active_rows=0,
           max_err_ik=0.0, max_err_mpc=0.0)
t0 = time.time()
for s in range(S):
    rng = np.random.default_rng(9000 + OFF + s)
    robot = rb.NAMES[s % len(rb.NAMES)]
    # ---- IK
    vmax = float(rng.choice([0.15, 0.2, 0.25, 0.3, 0.4, 0.55, 0.7, 1.0]))
    form = "qpoases" if s % 3 != 2 else "osqp"
    alg = [0, 0, 3, 4][s % 4]                      # default (base elimination) twice as often; 32-lane and general 16-lane kernels
    rr = rb.ROBOTS[robot]
    b = wca.synth.synth_ik_batch(B, seed=31000 + OFF + s, additional_rotation=np.array(rr["additional_rotation"]), posture_deg=np.array(rr["reg_deg"], float))
    p = rb.ik_params(qs, robot, vmax)
    ik = rb.ik_solver(wca, robot, form, vmax, algorithm=alg)
    out = ik.solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    err = 0.0; ninf = 0; ncmp = 0; nact = 0
    for i in range(B):
        x = qs.ik_inputs_from_batch(b, i)
        try:
            r = qs.ik_exact(p, x, form)
        except qs.QPInfeasible:
            assert out["status"][i] == wca.STATUS_INFEASIBLE, ("ik", s, i, int(out["status"][i]))
            ninf += 1
            continue
        except qs.QPOracleError as ex:
            # the oracle could not certify ITS OWN point (its KKT certificate failed): nothing to compare with; say what the device said
            print(json.dumps(dict(case="ik: oracle uncertified", seed=31000 + OFF + s, robot=robot, instance=i, v_max=vmax, form=form, oracle=str(ex), device_status=int(out["status"][i]))), flush=True)
            tot["ik_oracle_uncertified"] = tot.get("ik_oracle_uncertified", 0) + 1
            continue
        assert out["status"][i] == wca.STATUS_SOLVED, ("ik", s, i, int(out["status"][i]))
        e = float(np.abs(out["dq"][i] - r["dq"]).max())
        assert e <= SOL_TOL, ("ik", s, i, e)
        err = max(err, e)
        nact += bool(r["lower"] or r["upper"])
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active_lower"][i]) == sum(1 << j for j in r["lower"]) and int(out["active_upper"][i]) == sum(1 << j for j in r["upper"]), ("ik set", s, i)
            ncmp += 1
    print(json.dumps(dict(case="ik", seed=31000 + OFF + s, robot=robot, form=form, v_max=vmax, algorithm=alg, instances=B, infeasible=ninf, active_sets_compared=ncmp,
                          with_active_bounds=nact, max_abs_err=err)), flush=True)
    tot["ik_checked"] += B; tot["ik_infeasible"] += ninf; tot["ik_active_sets_compared"] += ncmp; tot["ik_with_active_bounds"] += nact; tot["max_err_ik"] = max(tot["max_err_ik"], err)
    # ---- MPC
    N = int(rng.choice([7, 20, 50, 63, 64, 100, 200]))
    mp = rb.mpc_params(qs, robot, horizon=N)
    c = qs.mpc_constants(mp)
    mb = wca.synth.synth_mpc_batch(B, seed=41000 + OFF + s, uprev_sigma=float(rng.choice([0.005, 0.03, 0.06])), x0_sigma=float(rng.choice([0.01, 0.03])), horizon=N)
    mo = rb.mpc_solver(wca, robot, horizon=N).solve_host(mb["x0"], mb["ref"], mb["u_prev"], mb["hull_A"], mb["hull_b"], mb["hull_nc"])
    err = 0.0; ncmp = 0; nact = 0
    for i in range(B):
        r = qs.mpc_exact(c, mb["x0"][i], mb["ref"][i], mb["u_prev"][i], mb["hull_A"][i], mb["hull_b"][i], int(mb["hull_nc"][i]))
        e = float(np.abs(mo["u0"][i] - r["u0"]).max())
        assert e <= SOL_TOL, ("mpc", s, i, e)
        err = max(err, e)
        nact += bool(r["active"])
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(mo["active"][i]) == sum(1 << k for k in r["active"]), ("mpc set", s, i)
            ncmp += 1
    print(json.dumps(dict(case="mpc", seed=41000 + OFF + s, robot=robot, horizon=N, instances=B, active_sets_compared=ncmp, with_active_rows=nact, max_abs_err=err)), flush=True)
    tot["mpc_checked"] += B; tot["mpc_active_sets_compared"] += ncmp; tot["mpc_with_active_rows"] += nact; tot["max_err_mpc"] = max(tot["max_err_mpc"], err)
tot["seconds"] = round(time.time() - t0, 1)
print(json.dumps(dict(summary=tot)), flush=True)
