import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, walking_controllers_amd as wca
B = 4096
b = wca.synth.synth_ik_batch(B, seed=4321)
res = {}
for alg in (1, 2, 3):
    s = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5, algorithm=alg)
    res[alg] = s.solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"], want_foot_err=False)
    print(alg, "status hist", np.bincount(res[alg]["status"]), "dq[0,:3]", res[alg]["dq"][0, :3])
print("diff 2 vs 3", np.abs(res[2]["dq"] - res[3]["dq"]).max(), "diff 1 vs 2", np.abs(res[1]["dq"] - res[2]["dq"]).max())
