#!/usr/bin/env python3
"""BASELINE workload (IK + MPC per step) as ONE combined plan (qp_plan_kernel: the MPC on the IK's lanes) against an IK-only plan and an
MPC-only plan enqueued together on two streams (ik_plan_kernel + mpc_plan_kernel); every step its own input arrays; wall time from the
first enqueue to the completion of both streams (host clock around polled events, like bench.py).   python tools/split_plan_timing.py [B] [steps]"""
import json, math, os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
mpc = wca.MpcSolver(horizon=50)
ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5, jacobian_structure=wca.IK_JAC_MIXED)
mb, ib = wca.synth.synth_mpc_batch(B, seed=1234), wca.synth.synth_ik_batch(B, seed=4321)
mk, ikk = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc"), ("J_left", "J_right", "J_neck", "J_com", "q", "state")
base = {k: torch.from_numpy(np.ascontiguousarray(mb[k])).to(dev) for k in mk}
base.update({k: torch.from_numpy(np.ascontiguousarray(ib[k])).to(dev) for k in ikk})
K = max(2 * S + 8, -(-(1 << 30) // (B * 6096)))
sets = [base] + [{k: torch.roll(v, shifts=j * max(1, B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
N1 = base["ref"].shape[1]
wj = int(max(1, min(S, max(4, -(-16384 // ((B + 3) // 4))))))
wm = 17
def outs():
    return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), ms=torch.zeros(B, dtype=torch.int32, device=dev), ma=torch.zeros(B, dtype=torch.int32, device=dev),
                mm=torch.zeros(B, dtype=torch.float64, device=dev), dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev),
                lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev), it=torch.zeros(B, dtype=torch.int32, device=dev))
O = [outs() for _ in range(max(wj, wm))]
def recs(first, with_mpc, with_ik, nout):
    r_ = (wca.capi.QpStep * S)()
    for n in range(S):
        d, o, r = sets[(first + n) % K], O[n % nout], r_[n]
        if with_mpc:
            r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = d["x0"].data_ptr(), d["ref"].data_ptr(), N1, d["u_prev"].data_ptr(), d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr()
            r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr()
        if with_ik:
            r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = (d[k].data_ptr() for k in ikk)
            r.dq, r.ik_status, r.active_lower, r.active_upper, r.iters = o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr(), o["it"].data_ptr()
    return r_
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def timed(plans_streams):
    e = [torch.cuda.Event() for _ in plans_streams]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (p, st), ev in zip(plans_streams, e):
        p.enqueue(st.cuda_stream)
    for (p, st), ev in zip(plans_streams, e):
        ev.record(st)
    while not all(ev.query() for ev in e):
        pass
    return (time.perf_counter() - t0) * 1e6
res = {"batch": B, "steps": S, "input_sets": K, "ik_ways": wj, "mpc_ways": wm}
rows = {"combined": [], "split_two_streams": [], "split_one_stream": []}
for rep in range(6):
    first = (rep * S) % K
    both = wca.capi.QpPlan(mpc, ik, B, recs(first, True, True, wj), ways=wj)
    pi = wca.capi.QpPlan(None, ik, B, recs(first, False, True, wj), ways=wj)
    pm = wca.capi.QpPlan(mpc, None, B, recs(first, True, False, wm), ways=wm)
    if rep == 0:                       # the kernels' first launches
        both.enqueue(s1.cuda_stream); pi.enqueue(s1.cuda_stream); pm.enqueue(s2.cuda_stream); torch.cuda.synchronize()
        first = S % K
        both.close(); pi.close(); pm.close()
        both = wca.capi.QpPlan(mpc, ik, B, recs(first, True, True, wj), ways=wj)
        pi = wca.capi.QpPlan(None, ik, B, recs(first, False, True, wj), ways=wj)
        pm = wca.capi.QpPlan(mpc, None, B, recs(first, True, False, wm), ways=wm)
    rows["combined"].append(timed([(both, s1)]))
    rows["split_two_streams"].append(timed([(pi, s1), (pm, s2)]))
    rows["split_one_stream"].append(timed([(pi, s1), (pm, s1)]))
    assert all(int((o["st"] == 0).sum()) == B for o in O[:wj])
    both.close(); pi.close(); pm.close()
for k, v in rows.items():
    v = sorted(v)
    res[k] = {"us_per_step_median": v[len(v) // 2] / S, "qp_per_s_median": 2 * B * S / (v[len(v) // 2] * 1e-6), "us_total_all": [round(x, 1) for x in v]}
print(json.dumps(res))
