import sys, numpy as np
sys.path.insert(0, '.')
import walking_controllers_amd as wca
B = 1000003
b = wca.synth.synth_ik_batch(B, seed=77)
args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5)
out = ik.solve_host(*args)
print("solved", int((out["status"] == 0).sum()), "of", B)
idx = np.array([0, 1, 2, 3, 4095, 4096, 500000, B - 4, B - 3, B - 2, B - 1])
sub = {k: v[idx] for k, v in b.items() if hasattr(v, "shape") and v.shape[0] == B}
ref = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5, algorithm=3).solve_host(sub["J_left"], sub["J_right"], sub["J_neck"], sub["J_com"], sub["q"], sub["state"])
print("max diff vs 32-lane kernel on sampled rows", float(np.abs(out["dq"][idx] - ref["dq"]).max()), (out["status"][idx] == ref["status"]).all())
m = wca.MpcSolver(horizon=50)
mb = wca.synth.synth_mpc_batch(B, seed=5)
mo = m.solve_host(mb["x0"], mb["ref"], mb["u_prev"], mb["hull_A"], mb["hull_b"], mb["hull_nc"])
print("mpc solved", int(np.isin(mo["status"], (0, 3)).sum()), "of", B)
