#!/usr/bin/env python3
"""BASELINE config 3 on its own (batched QP-IK): the IK-only plan (one launch walks through the batches: qp_plan_kernel without its MPC
share) against a launch per batch; cold input sets, HIP events.   python tools/ik_plan_timing.py [batch] [steps]"""
import json, math, os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5, jacobian_structure=wca.IK_JAC_MIXED)
ib = wca.synth.synth_ik_batch(B, seed=4321)
keys = ("J_left", "J_right", "J_neck", "J_com", "q", "state")
base = {k: torch.from_numpy(np.ascontiguousarray(ib[k])).to(dev) for k in keys}
set_bytes = B * 5056
K = int(max(13, min(256, -(-(1 << 30) // set_bytes))))
sets = [base] + [{k: torch.roll(v, shifts=j * max(1, B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
st = torch.cuda.Stream(dev)
sp = st.cuda_stream
res = {"batch": B, "steps": S, "input_sets": K, "input_MB_per_set": set_bytes / 1e6, "algorithmic_bytes_per_qp": 5240}
def outs():
    return dict(dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev),
                lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev), it=torch.zeros(B, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
for ways in (1, 2, 4, 8):
    while math.gcd(K, ways) != 1 and ways > 1:
        ways += 1
    o = [outs() for _ in range(ways)]
    recs = (wca.capi.QpStep * S)()
    for n in range(S):
        d, q = sets[n % K], o[n % ways]
        r = recs[n]
        r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = (d[k].data_ptr() for k in keys)
        r.dq, r.ik_status, r.active_lower, r.active_upper, r.iters = q["dq"].data_ptr(), q["st"].data_ptr(), q["lo"].data_ptr(), q["up"].data_ptr(), q["it"].data_ptr()
    plan = wca.capi.QpPlan(None, ik, B, recs, ways=ways)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.enqueue(sp); plan.enqueue(sp)
    torch.cuda.synchronize()
    e0.record(st)
    reps = 5
    for _ in range(reps):
        plan.enqueue(sp)
    e1.record(st)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps / S
    res["plan_ways_%d" % ways] = {"us_per_batch": us, "ik_qp_per_s": B / us * 1e6, "hbm_frac": 5240 * B / (us * 1e-6) / 8e12}
    assert all(int((q["st"] == 0).sum()) == B for q in o)
    plan.close()
q = outs()
def one(d):
    ik.solve_device(B, *(d[k].data_ptr() for k in keys), q["dq"].data_ptr(), q["st"].data_ptr(), q["lo"].data_ptr(), q["up"].data_ptr(), 0, q["it"].data_ptr(), sp)
for n in range(K):
    one(sets[n % K])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for n in range(S):
    one(sets[n % K])
e1.record(st)
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / S
res["launch_per_batch"] = {"us_per_batch": us, "ik_qp_per_s": B / us * 1e6, "hbm_frac": 5240 * B / (us * 1e-6) / 8e12}
print(json.dumps(res))
