"""
Generates the golden vectors under tests/golden/ from the exact fp64 oracle
(oracle/qp_spec.py) on the seeded synthetic workloads of
walking-controllers_amd/synth.py.  The reference holds no fixtures for this path
("parity unpinned"), so these vectors pin the ORACLE'S exact
optimum; inputs are regenerated from (seed, index) and a few rows of raw input
are stored too, so that the generator itself stays pinned.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import walking_controllers_amd as wca  # noqa: E402
from oracle import qp_spec as qs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def mpc_golden(name, count, seed, **kw):
    c = qs.mpc_constants(qs.MPCParams())
    b = wca.synth.synth_mpc_batch(count, seed=seed, **kw)
    u0 = np.zeros((count, 2)); act = np.zeros(count, np.uint32)
    mu_min = np.zeros(count); slack_min = np.zeros(count); margin = np.zeros(count)
    for i in range(count):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        u0[i] = r["u0"]
        act[i] = sum(1 << e for e in r["active"])
        mu_min[i] = r["mu_min_active"]; slack_min[i] = r["slack_min_inactive"]; margin[i] = r["margin"]
    np.savez_compressed(os.path.join(HERE, name), count=count, seed=seed, kw=repr(kw),
                        u0=u0, active=act, mu_min_active=mu_min, slack_min_inactive=slack_min, margin=margin,
                        in_x0=b["x0"][:4], in_ref=b["ref"][:4], in_u_prev=b["u_prev"][:4],
                        in_hull_A=b["hull_A"][:4], in_hull_b=b["hull_b"][:4], in_hull_nc=b["hull_nc"][:4])
    print(name, "active-count hist", np.bincount([bin(a).count("1") for a in act]))


def ik_golden(name, count, seed, form, v_max):
    p = qs.IKParams(v_max=v_max * np.ones(23))
    b = wca.synth.synth_ik_batch(count, seed=seed)
    dq = np.zeros((count, 23)); lo = np.zeros(count, np.uint32); up = np.zeros(count, np.uint32)
    mu_min = np.zeros(count); slack_min = np.zeros(count); ferr = np.zeros((count, 12))
    status = np.zeros(count, np.int32)
    for i in range(count):
        x = qs.ik_inputs_from_batch(b, i)
        try:
            r = qs.ik_exact(p, x, form)
        except qs.QPInfeasible:
            status[i] = 2
            continue
        dq[i] = r["dq"]
        lo[i] = sum(1 << j for j in r["lower"]); up[i] = sum(1 << j for j in r["upper"])
        mu_min[i] = r["mu_min_active"]; slack_min[i] = r["slack_min_inactive"]
        ferr[i, :6] = r["foot_err_left"]; ferr[i, 6:] = r["foot_err_right"]
    np.savez_compressed(os.path.join(HERE, name), count=count, seed=seed, form=form, v_max=v_max,
                        dq=dq, active_lower=lo, active_upper=up, status=status,
                        mu_min_active=mu_min, slack_min_inactive=slack_min, foot_err=ferr,
                        in_J_left=b["J_left"][:2], in_J_com=b["J_com"][:2], in_q=b["q"][:2], in_state=b["state"][:2])
    print(name, "active-count hist", np.bincount([bin(int(a) | int(c)).count("1") for a, c in zip(lo, up)]),
          "infeasible", int((status == 2).sum()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ik4096":              # only the full-batch config-3 golden (round 4)
        ik_golden("ik_qpoases_v050_b4096.npz", 4096, 4321, "qpoases", 0.5)
        sys.exit(0)
    mpc_golden("mpc_cfg2_b4096.npz", 4096, 1234)                       # BASELINE config 2
    mpc_golden("mpc_stress_b1024.npz", 1024, 77, uprev_sigma=0.04)     # hull rows active
    ik_golden("ik_qpoases_v050_b1024.npz", 1024, 4321, "qpoases", 0.5)
    ik_golden("ik_qpoases_v050_b4096.npz", 4096, 4321, "qpoases", 0.5)   # BASELINE config 3 at its full batch: every row bench.py times
    ik_golden("ik_qpoases_v030_b512.npz", 512, 4321, "qpoases", 0.30)
    ik_golden("ik_osqp_b512.npz", 512, 4321, "osqp", 1.0)
