#!/usr/bin/env python3
"""BASELINE config 2 on its own (batched DCM-MPC): the MPC-only plan (one launch walks through the batches, wcqp_qp_plan_* with records
that have no IK part) against a launch per batch; cold input sets, HIP events.   python tools/mpc_plan_timing.py [batch] [steps]"""
import json, math, os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
mpc = wca.MpcSolver(horizon=50)
mb = wca.synth.synth_mpc_batch(B, seed=1234)
base = {k: torch.from_numpy(np.ascontiguousarray(mb[k])).to(dev) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")}
set_bytes = B * 1040
K = int(max(13, min(512, -(-(1 << 30) // set_bytes))))          # > 1 GiB in total, or 512 sets
sets = [base] + [{k: torch.roll(v, shifts=j * max(1, B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
N1 = base["ref"].shape[1]
st = torch.cuda.Stream(dev)
sp = st.cuda_stream
res = {"batch": B, "steps": S, "input_sets": K, "input_MB_per_set": set_bytes / 1e6, "algorithmic_bytes_per_qp": 1056}
def outs():
    return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), ms=torch.zeros(B, dtype=torch.int32, device=dev),
                ma=torch.zeros(B, dtype=torch.int32, device=dev), mm=torch.zeros(B, dtype=torch.float64, device=dev))
torch.cuda.synchronize()
for ways in (1, 2, 4, 8, 16, 28):
    while math.gcd(K, ways) != 1 and ways > 1:
        ways += 1
    o = [outs() for _ in range(ways)]
    recs = (wca.capi.QpStep * S)()
    for n in range(S):
        d, q = sets[n % K], o[n % ways]
        r = recs[n]
        r.x0, r.ref, r.ref_len, r.u_prev = d["x0"].data_ptr(), d["ref"].data_ptr(), N1, d["u_prev"].data_ptr()
        r.hull_A, r.hull_b, r.hull_nc = d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr()
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = q["u0"].data_ptr(), q["ms"].data_ptr(), q["ma"].data_ptr(), q["mm"].data_ptr()
    plan = wca.capi.QpPlan(mpc, None, B, recs, ways=ways)
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
    res["plan_ways_%d" % ways] = {"us_per_batch": us, "mpc_qp_per_s": B / us * 1e6, "hbm_frac": 1056 * B / (us * 1e-6) / 8e12}
    assert all(int((q["ms"] == 0).sum()) + int((q["ms"] == 3).sum()) == B for q in o)
    plan.close()
# a launch per batch on one stream
q = outs()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def one(d):
    mpc.solve_device(B, d["x0"].data_ptr(), d["ref"].data_ptr(), N1, d["u_prev"].data_ptr(), d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr(),
                     q["u0"].data_ptr(), q["ms"].data_ptr(), q["ma"].data_ptr(), q["mm"].data_ptr(), sp)
for n in range(K):
    one(sets[n % K])
torch.cuda.synchronize()
e0.record(st)
for n in range(S):
    one(sets[n % K])
e1.record(st)
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / S
res["launch_per_batch"] = {"us_per_batch": us, "mpc_qp_per_s": B / us * 1e6, "hbm_frac": 1056 * B / (us * 1e-6) / 8e12}
print(json.dumps(res))
