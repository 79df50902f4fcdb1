"""Diagnostic: where the fixed cost around bench.py's timed region goes (idle -> first launch, end barrier)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import walking_controllers_amd as wca
dev = torch.device("cuda", 0); B = 4096
ib = wca.synth.synth_ik_batch(B, seed=4321); mb = wca.synth.synth_mpc_batch(B, seed=1234)
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in {**ib, **{k: mb[k] for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")}}.items()}
dq = torch.zeros(B, 23, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev); ms = torch.zeros(B, dtype=torch.int32, device=dev)
ik = wca.IkSolver(v_max=0.5, jacobian_structure=wca.IK_JAC_MIXED); mpc = wca.MpcSolver()
s0 = torch.cuda.current_stream(dev); s1 = torch.cuda.Stream(dev)
def step():
    mpc.solve_device(B, d["x0"].data_ptr(), d["ref"].data_ptr(), 51, d["u_prev"].data_ptr(), d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr(), u0.data_ptr(), ms.data_ptr(), 0, 0, s1.cuda_stream)
    ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(), d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), 0, 0, 0, 0, s0.cuda_stream)
for _ in range(30): step()
torch.cuda.synchronize()
res = {}
for n in (1, 5, 20, 40, 100, 400):
    ts = []
    for rep in range(7):
        torch.cuda.synchronize(); time.sleep(0.002)
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts = np.array(ts)
    res[n] = {"enqueue_us": round(1e6 * np.median(ts[:, 0]), 1), "total_us": round(1e6 * np.median(ts[:, 1]), 1), "per_step_us": round(1e6 * np.median(ts[:, 1]) / n, 2)}
print(json.dumps(res))
