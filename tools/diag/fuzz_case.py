#!/usr/bin/env python3
"""One case of tools/fuzz_vs_oracle.py looked at instance by instance: the default kernel and both fall-backs against the exact oracle and
against each other, with the oracle's conditioning figures.   python tools/diag/fuzz_case.py <seed offset> <case index s> [instances]"""
import json, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import walking_controllers_amd as wca
from oracle import qp_spec as qs
import robots as rb
OFF, s = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(9000 + OFF + s)
robot = rb.NAMES[s % len(rb.NAMES)]
vmax = float(rng.choice([0.15, 0.2, 0.25, 0.3, 0.4, 0.55, 0.7, 1.0]))
form = "qpoases" if s % 3 != 2 else "osqp"
rr = rb.ROBOTS[robot]
b = wca.synth.synth_ik_batch(B, seed=31000 + OFF + s, additional_rotation=np.array(rr["additional_rotation"]), posture_deg=np.array(rr["reg_deg"], float))
p = rb.ik_params(qs, robot, vmax)
outs = {alg: rb.ik_solver(wca, robot, form, vmax, algorithm=alg).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"]) for alg in (0, 4, 3)}
rows = []
for i in range(B):
    try:
        r = qs.ik_exact(p, qs.ik_inputs_from_batch(b, i), form)
    except qs.QPOracleError:
        continue
    e = {alg: float(np.abs(o["dq"][i] - r["dq"]).max()) for alg, o in outs.items()}
    H, g, A, lb, ub, lbA, ubA = qs.ik_assemble_qpoases(p, qs.ik_inputs_from_batch(b, i))
    act = sorted(r["lower"]) + sorted(r["upper"])
    # conditioning of the KKT of the optimal working set: equality rows + the active bounds
    E = np.zeros((len(act), 29)); E[np.arange(len(act)), [6 + a for a in act]] = 1.0
    K = np.block([[H, A.T, E.T], [A, np.zeros((15, 15)), np.zeros((15, len(act)))], [E, np.zeros((len(act), 15)), np.zeros((len(act), len(act)))]])
    rows.append(dict(i=i, err=e, n_active=len(act), mu_min=float(r["mu_min_active"]), slack_min=float(r["slack_min_inactive"]), cond_kkt=float(np.linalg.cond(K)),
                     iters=int(outs[0]["iters"][i]) if "iters" in outs[0] else None, d34=float(np.abs(outs[0]["dq"][i] - outs[4]["dq"][i]).max())))
rows.sort(key=lambda r: -r["err"][0])
print(json.dumps(dict(robot=robot, form=form, v_max=vmax, seed=31000 + OFF + s)))
for r in rows[:8]:
    print(json.dumps(r))
print(json.dumps(dict(median_err={alg: float(np.median([r["err"][alg] for r in rows])) for alg in (0, 4, 3)}, median_cond=float(np.median([r["cond_kkt"] for r in rows])))))
