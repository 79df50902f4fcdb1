"""Diagnostic: per-phase cycle shares of ik2_kernel from s_memtime stamps (stamp build only)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vmax = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dev = torch.device("cuda", 0)
ib = wca.synth.synth_ik_batch(B, seed=4321)
d = {k: torch.from_numpy(ib[k]).to(dev) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}
dq = torch.zeros(B, 23, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
alg = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nw = (B + 3) // 4 if alg == 4 else (B + 1) // 2
dbg = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=alg)
sp = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(), d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), 0, 0, dbg.data_ptr(), 0, sp)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nw, 16)[:, :10].astype(np.float64)
seg = np.diff(t, axis=1)
names = ["loads", "rhs+grad", "gauss-jordan", "rows->tables", "Hr build", "sweep", "x_N,x_B", "active set", "outputs"]
full = dbg.cpu().numpy().reshape(nw, 16).astype(np.float64)
gj = {"gi: init": float(np.median(full[:, 10] - full[:, 7])), "first bound (straight-line)": float(np.median(full[:, 14] - full[:, 10])), "general loop": float(np.median(full[:, 15] - full[:, 14])), "cert+rest": float(np.median(full[:, 8] - full[:, 15]))} if alg == 4 else {}
print(json.dumps({"gj_detail": gj}))
print(json.dumps({"B": B, "vmax": vmax, "median_cycles": dict(zip(names, np.median(seg, 0).tolist())),
                  "total_median": float(np.median(t[:, 9] - t[:, 0])), "span_all_waves": float(t[:, 9].max() - t[:, 0].min())}))
