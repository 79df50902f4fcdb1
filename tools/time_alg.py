import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536; dev = torch.device("cuda", 0)
ib = wca.synth.synth_ik_batch(B, seed=4321)
d = {k: torch.from_numpy(ib[k]).to(dev) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}
dq = torch.zeros(B, 23, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
sp = torch.cuda.current_stream().cuda_stream
out = {}
for alg, js in ((3, 0), (4, 0), (5, 0), (5, 1)):      # 32-lane, general 16-lane, base elimination with / without the fall-back launch
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=float(sys.argv[2]) if len(sys.argv) > 2 else 100.0, algorithm=alg, jacobian_structure=js)
    run = lambda: ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(), d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), 0, 0, 0, 0, sp)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    out[f"{alg}/{js}"] = e0.elapsed_time(e1) / 20
    out[f"{alg}/{js}_solved"] = int((st == 0).sum())
print(json.dumps({"B": B, "lib": os.path.basename(os.environ.get("WCQP_LIB_PATH", "libwcqp.so")), "ms": out}))
