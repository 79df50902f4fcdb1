import sys; sys.path.insert(0, '.')
import numpy as np, walking_controllers_amd as wca
B = 160; b = wca.synth.synth_ik_batch(B, seed=99); vmax = 0.22
s = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax)
out = s.solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
np.set_printoptions(linewidth=200, precision=5)
for i in (86, 87, 110, 111, 133, 132):
    print(i, "status", out["status"][i], "iters", out["iters"][i], "lo", bin(out["active_lower"][i]), "up", bin(out["active_upper"][i]))
    print("   dq", out["dq"][i])
print("status hist", np.bincount(out["status"]), "nan rows", np.isnan(out["dq"]).any(axis=1).sum())
# single-instance batches for the same robots (partner lane group idle)
for i in (87, 110, 133):
    o1 = s.solve_host(b["J_left"][i:i+1], b["J_right"][i:i+1], b["J_neck"][i:i+1], b["J_com"][i:i+1], b["q"][i:i+1], b["state"][i:i+1])
    print("single", i, "status", o1["status"][0], "iters", o1["iters"][0], "maxabs", np.abs(o1["dq"][0]).max())
