#!/usr/bin/env python3
"""Single-batch latency form, the two-pass idea, priced at its BEST case before any kernel is written for it:

    one launch  (a)  ik4_kernel, qpOASES form, B = 4096: every wave solves its four robots completely; the launch ends with the wave
                     that walks the longest active set (14 us = 0.19 of the HBM roofline)
    two passes  (b)  the same 4096 robots WITHOUT bounds (the osqp form's kernel path: equality solve only - what a first pass that
                     merely FLAGS the robots whose optimum violates a bound costs at least), then
                (c)  the flagged robots alone (29 % at v_max 0.5: 1189 robots), gathered into a dense batch of their own, solved completely
                     (what a compacted second pass costs at least: its gather is not even counted)

(b) + (c) back to back on one stream against (a), cold inputs (a rotation of input sets > 1 GiB), HIP events around 40 repetitions.
    python tools/two_pass_timing.py            ->  one JSON line
"""
import json, os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca

B, VMAX = 4096, 0.5
dev = torch.device("cuda", 0)
ib = wca.synth.synth_ik_batch(B, seed=4321)
keys = ("J_left", "J_right", "J_neck", "J_com", "q", "state")
ikq = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=VMAX, jacobian_structure=wca.IK_JAC_MIXED)
iko = wca.IkSolver(form=wca.IK_FORM_OSQP, v_max=VMAX, jacobian_structure=wca.IK_JAC_MIXED)
ref = ikq.solve_host(*[ib[k] for k in keys])
need = np.flatnonzero((ref["active_lower"] | ref["active_upper"]) != 0)          # the robots a first pass would flag
nb = len(need)
sub = {k: np.ascontiguousarray(ib[k][need]) for k in keys}
K = 44                                                                          # > 1 GiB of input sets at 4096 robots
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
full_sets = [{k: up(np.roll(ib[k], s * (B // K), axis=0)) for k in keys} for s in range(K)]
sub_sets = [{k: up(np.roll(sub[k], s * max(1, nb // K), axis=0)) for k in keys} for s in range(K)]
out = dict(dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev),
           lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev))
stream = torch.cuda.Stream(dev)
sp = stream.cuda_stream


def launch(solver, d, n):
    solver.solve_device(n, *[d[k].data_ptr() for k in keys], out["dq"].data_ptr(), out["st"].data_ptr(), out["lo"].data_ptr(), out["up"].data_ptr(), 0, 0, sp)


def timed(fn, reps=40):
    for i in range(K):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(reps):
        fn(i)
    e1.record(stream)
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


one = timed(lambda i: launch(ikq, full_sets[i % K], B))
first = timed(lambda i: launch(iko, full_sets[i % K], B))
second = timed(lambda i: launch(ikq, sub_sets[i % K], nb))
both = timed(lambda i: (launch(iko, full_sets[i % K], B), launch(ikq, sub_sets[i % K], nb)))
frac = lambda us: 5240.0 * B / (us * 1e-6) / 8e12
print(json.dumps({"batch": B, "v_max": VMAX, "robots_with_active_bounds": nb, "share": nb / B,
                  "one_launch_us": one, "one_launch_hbm_frac": frac(one),
                  "first_pass_no_bounds_us": first, "second_pass_flagged_only_us": second, "two_passes_back_to_back_us": both,
                  "two_passes_hbm_frac": frac(both), "verdict": "two passes win" if both < one else "one launch wins"}))
