"""Differential stress of the IK kernels against each other (diagnostic): many instances, tight bounds, both forms."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rep = []
for seed in (101, 202):
    b = wca.synth.synth_ik_batch(B, seed=seed)
    args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    for form, vmax in ((wca.IK_FORM_QPOASES, 0.5), (wca.IK_FORM_QPOASES, 0.3), (wca.IK_FORM_QPOASES, 0.2), (wca.IK_FORM_QPOASES, 0.12), (wca.IK_FORM_OSQP, 0.3)):
        outs = {a: wca.IkSolver(form=form, v_max=vmax, algorithm=a).solve_host(*args) for a in (5, 4, 3)}
        ref = outs[3]
        for a in (5, 4):
            o = outs[a]
            both = (o["status"] == 0) & (ref["status"] == 0)
            rep.append(dict(seed=seed, form=int(form), vmax=vmax, alg=a, n=B, solved=int((o["status"] == 0).sum()), solved_ref=int((ref["status"] == 0).sum()),
                            status_mismatch=int((o["status"] != ref["status"]).sum()),
                            max_dq_diff=float(np.abs(o["dq"][both] - ref["dq"][both]).max(initial=0.0)),
                            set_mismatch=int(((o["active_lower"][both] != ref["active_lower"][both]) | (o["active_upper"][both] != ref["active_upper"][both])).sum()),
                            mean_active=float(np.mean([bin(int(x)).count("1") for x in (ref["active_lower"][both] | ref["active_upper"][both])[:2000]])),
                            max_iters=int(o["iters"].max())))
            print(json.dumps(rep[-1]), flush=True)
