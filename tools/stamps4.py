"""Diagnostic: per-phase cycles of ik4_kernel from s_memtime stamps (stamp build only: tools/build_variant.sh stamps -DWCQP_IK_STAMPS)."""
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
nw = (B + 3) // 4
dbg = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=5, jacobian_structure=1)
sp = torch.cuda.current_stream().cuda_stream
flush = torch.zeros(64 << 20, dtype=torch.float64, device=dev)      # 512 MiB: evicts the inputs from the Infinity Cache
for _ in range(3):
    flush.add_(1.0)
    ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(), d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), 0, 0, dbg.data_ptr(), 0, sp)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 16).astype(np.float64)
order = [0, 1, 2, 3, 4, 10, 11, 5, 6, 7, 8, 9]
names = ["loads", "rhs", "pattern+dB", "transform", "C^T stores", "mfma", "tile+rows", "sweep", "y,x", "active set", "outputs"]
t = full[:, order]
seg = np.diff(t, axis=1)
print(json.dumps({"B": B, "vmax": vmax, "median_cycles": dict(zip(names, np.median(seg, 0).tolist())),
                  "p90_active_set": float(np.percentile(seg[:, 9], 90)), "max_active_set": float(seg[:, 9].max()),
                  "total_median": float(np.median(t[:, -1] - t[:, 0])), "total_max": float((t[:, -1] - t[:, 0]).max()),
                  "span_all_waves": float(t[:, -1].max() - t[:, 0].min())}))
