"""Runs only the IK kernel (one algorithm) a few times: the PMC passes of collect_ik.sh wrap this."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import walking_controllers_amd as wca
B = int(sys.argv[1]); alg = int(sys.argv[2]); vmax = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
dev = torch.device("cuda", 0)
ib = wca.synth.synth_ik_batch(B, seed=4321)
d = {k: torch.from_numpy(ib[k]).to(dev) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}
dq = torch.zeros(B, 23, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
sp = torch.cuda.current_stream().cuda_stream
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=alg, jacobian_structure=wca.IK_JAC_MIXED)
for _ in range(12):
    ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(),
                    d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), 0, 0, 0, 0, sp)
torch.cuda.synchronize()
print("ok", B, alg, int((st == 0).sum()))
