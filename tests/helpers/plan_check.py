"""Run by tests/test_gpu_parity.py::test_qp_plan_equals_the_single_calls in a process of its own (torch first, then libwcqp):
a plan of 7 records (wcqp_qp_plan_*: ONE launch walks through them, `ways` wavefronts per robot group, each way with its own
output buffers; ways = 0: (record, robot group) units handed out from a work queue) against the single wcqp_mpc_solve_device / wcqp_ik_solve_device calls of the same records, bit for bit."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import walking_controllers_amd as wca


def main(B, ways, R=7, horizon=50, graph=False):
    dev = torch.device("cuda", 0)
    mpc, ik = wca.MpcSolver(horizon=horizon), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4, jacobian_structure=wca.IK_JAC_MIXED)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sets = []
    for seed in range(R):
        mb, ib = wca.synth.synth_mpc_batch(B, seed=50 + seed, uprev_sigma=0.03, horizon=horizon), wca.synth.synth_ik_batch(B, seed=150 + seed)
        sets.append(({k: t(mb[k]) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")},
                     {k: t(ib[k]) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}))

    def outs():
        return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), ms=torch.full((B,), -1, dtype=torch.int32, device=dev),
                    ma=torch.zeros(B, dtype=torch.int32, device=dev), mm=torch.zeros(B, dtype=torch.float64, device=dev),
                    dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.full((B,), -1, dtype=torch.int32, device=dev),
                    lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev),
                    fe=torch.zeros(B, 12, dtype=torch.float64, device=dev), it=torch.zeros(B, dtype=torch.int32, device=dev))
    ref_o, got_o = [outs() for _ in range(R)], [outs() for _ in range(R)]
    N1 = sets[0][0]["ref"].shape[1]
    torch.cuda.synchronize()
    for (m, i), o in zip(sets, ref_o):
        mpc.solve_device(B, m["x0"].data_ptr(), m["ref"].data_ptr(), N1, m["u_prev"].data_ptr(), m["hull_A"].data_ptr(),
                         m["hull_b"].data_ptr(), m["hull_nc"].data_ptr(), o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr(), 0)
        ik.solve_device(B, i["J_left"].data_ptr(), i["J_right"].data_ptr(), i["J_neck"].data_ptr(), i["J_com"].data_ptr(), i["q"].data_ptr(),
                        i["state"].data_ptr(), o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr(), o["fe"].data_ptr(), o["it"].data_ptr(), 0)
    torch.cuda.synchronize()
    recs = (wca.capi.QpStep * R)()
    for n, ((m, i), o) in enumerate(zip(sets, got_o)):     # every record its own outputs: whatever the way, nothing is shared
        r = recs[n]
        r.x0, r.ref, r.ref_len, r.u_prev = m["x0"].data_ptr(), m["ref"].data_ptr(), N1, m["u_prev"].data_ptr()
        r.hull_A, r.hull_b, r.hull_nc = m["hull_A"].data_ptr(), m["hull_b"].data_ptr(), m["hull_nc"].data_ptr()
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr()
        r.J_left, r.J_right, r.J_neck, r.J_com = (i[k].data_ptr() for k in ("J_left", "J_right", "J_neck", "J_com"))
        r.q, r.state = i["q"].data_ptr(), i["state"].data_ptr()
        r.dq, r.ik_status, r.active_lower, r.active_upper = o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr()
        r.foot_err, r.iters = o["fe"].data_ptr(), o["it"].data_ptr()
    plan = wca.capi.QpPlan(mpc, ik, B, recs, ways=ways)
    st = torch.cuda.Stream(dev)
    plan.enqueue(st.cuda_stream)
    plan.enqueue(st.cuda_stream)            # replay: same results
    torch.cuda.synchronize()
    for n, (a, b) in enumerate(zip(ref_o, got_o)):
        for k in a:
            assert torch.equal(a[k], b[k]), (B, ways, n, k)
    assert (got_o[0]["ms"] == 0).all() and sum(int((o["ma"] != 0).sum()) for o in got_o) > 0          # hull rows really bind somewhere
    if graph:
        # wcqp_qp_plan_enqueue is "enqueue only; graph-capturable" (include/wcqp.h): captured into a hipGraph and replayed twice,
        # the outputs cleared in between (the work-queue form re-arms its ticket counters inside the launch itself)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            plan.enqueue(st.cuda_stream)
        for _ in range(2):
            for o in got_o:
                for v in o.values():
                    v.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for n, (a, b) in enumerate(zip(ref_o, got_o)):
                for k in a:
                    assert torch.equal(a[k], b[k]), ("graph", B, ways, n, k)
        del g
    plan.close()
    # MPC-only plan (no record has an IK part; the IK handle may be missing): bit for bit the single MPC calls
    mrecs = (wca.capi.QpStep * R)()
    mo = [outs() for _ in range(R)]
    for n, ((m, i), o) in enumerate(zip(sets, mo)):
        r = mrecs[n]
        r.x0, r.ref, r.ref_len, r.u_prev = m["x0"].data_ptr(), m["ref"].data_ptr(), N1, m["u_prev"].data_ptr()
        r.hull_A, r.hull_b, r.hull_nc = m["hull_A"].data_ptr(), m["hull_b"].data_ptr(), m["hull_nc"].data_ptr()
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr()
    mplan = wca.capi.QpPlan(mpc, None, B, mrecs, ways=(ways if ways != 0 else 1))
    mplan.enqueue(st.cuda_stream); mplan.enqueue(st.cuda_stream)
    torch.cuda.synchronize()
    for n, (a, b) in enumerate(zip(ref_o, mo)):
        for k in ("u0", "ms", "ma", "mm"):
            assert torch.equal(a[k], b[k]), ("mpc-only", B, ways, n, k)
        assert (b["st"] == -1).all() and (b["dq"] == 0).all()          # the IK outputs were not touched
    mplan.close()
    # IK-only plan (no record has an MPC part; the MPC handle may be missing): bit for bit the single IK calls
    irecs = (wca.capi.QpStep * R)()
    io = [outs() for _ in range(R)]
    for n, ((m, i), o) in enumerate(zip(sets, io)):
        r = irecs[n]
        r.J_left, r.J_right, r.J_neck, r.J_com = (i[k].data_ptr() for k in ("J_left", "J_right", "J_neck", "J_com"))
        r.q, r.state = i["q"].data_ptr(), i["state"].data_ptr()
        r.dq, r.ik_status, r.active_lower, r.active_upper = o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr()
        r.foot_err, r.iters = o["fe"].data_ptr(), o["it"].data_ptr()
    iplan = wca.capi.QpPlan(None, ik, B, irecs, ways=ways)
    iplan.enqueue(st.cuda_stream); iplan.enqueue(st.cuda_stream)
    torch.cuda.synchronize()
    for n, (a, b) in enumerate(zip(ref_o, io)):
        for k in ("dq", "st", "lo", "up", "fe", "it"):
            assert torch.equal(a[k], b[k]), ("ik-only", B, ways, n, k)
        assert (b["ms"] == -1).all() and (b["u0"] == 0).all()          # the MPC outputs were not touched
    iplan.close()
    # a plan needs what one launch can do: an IK handle with the general fall-back behind it is refused
    try:
        wca.capi.QpPlan(mpc, wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4, jacobian_structure=wca.IK_JAC_AUTO), B, recs)
        raise AssertionError("a plan on an AUTO-structure handle must be refused")
    except wca.WcqpError:
        pass


if __name__ == "__main__":
    for B, ways in ((1, 1), (5, 3), (777, 2), (4096, 2), (4096, 9), (4096, 16), (1, 0), (777, 0), (4096, 0), (777, -1), (4096, -1)):      # ragged batches; more ways than records; 0 = work queue; -1 = WCQP_PLAN_WAYS_AUTO
        # (4096 robots x 7 records = 7168 units for the 2048 wavefronts that are resident at once; the replay starts from the queue the first launch put back)
        main(B, ways)
    main(8192, 0, R=5)
    main(777, 3, graph=True)
    main(4096, 0, graph=True)
    main(4096, 16, R=20)                  # the geometry bench.py times by default: 16 ways, 20 records
    main(333, 0, R=3, horizon=200)
    main(333, 2, R=3, horizon=200)                                         # the shipped horizon: more than one 64-stage pass of the window
    main(64, 1, R=2, horizon=7)
    print("plan ok")
