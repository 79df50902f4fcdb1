"""Run by tests/test_gpu_parity.py::test_qp_enqueue_steps_equals_the_single_calls in a process of its own: three records
against the single calls, bit for bit - record 0 on two streams (two launches), record 1 without an MPC part, record 2 with
both calls on one stream (the one-launch form: IK and MPC workgroups in one grid)."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import walking_controllers_amd as wca


def main(B, jac):
    dev = torch.device("cuda", 0)
    mpc, ik = wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4, jacobian_structure=jac)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sets = []
    for seed in (5, 6, 7):
        mb, ib = wca.synth.synth_mpc_batch(B, seed=seed), wca.synth.synth_ik_batch(B, seed=seed + 100)
        sets.append(({k: t(mb[k]) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")},
                     {k: t(ib[k]) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}))

    def outs():
        return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), ms=torch.full((B,), -1, dtype=torch.int32, device=dev),
                    ma=torch.zeros(B, dtype=torch.int32, device=dev), mm=torch.zeros(B, dtype=torch.float64, device=dev),
                    dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.full((B,), -1, dtype=torch.int32, device=dev),
                    lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev),
                    fe=torch.zeros(B, 12, dtype=torch.float64, device=dev), it=torch.zeros(B, dtype=torch.int32, device=dev))
    ref_o, got_o = [outs() for _ in range(3)], [outs() for _ in range(3)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    N1 = sets[0][0]["ref"].shape[1]
    for n, ((m, i), o) in enumerate(zip(sets, ref_o)):
        if n != 1:
            mpc.solve_device(B, m["x0"].data_ptr(), m["ref"].data_ptr(), N1, m["u_prev"].data_ptr(), m["hull_A"].data_ptr(),
                             m["hull_b"].data_ptr(), m["hull_nc"].data_ptr(), o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr(), 0)
        ik.solve_device(B, i["J_left"].data_ptr(), i["J_right"].data_ptr(), i["J_neck"].data_ptr(), i["J_com"].data_ptr(), i["q"].data_ptr(),
                        i["state"].data_ptr(), o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr(), o["fe"].data_ptr(), o["it"].data_ptr(), 0)
    torch.cuda.synchronize()
    recs = (wca.capi.QpStep * 3)()
    for n, ((m, i), o) in enumerate(zip(sets, got_o)):
        r = recs[n]
        if n != 1:
            r.x0, r.ref, r.ref_len, r.u_prev = m["x0"].data_ptr(), m["ref"].data_ptr(), N1, m["u_prev"].data_ptr()
            r.hull_A, r.hull_b, r.hull_nc = m["hull_A"].data_ptr(), m["hull_b"].data_ptr(), m["hull_nc"].data_ptr()
            r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr()
            r.mpc_stream = streams[1].cuda_stream if n == 0 else streams[0].cuda_stream     # record 2: both calls on one stream = ONE launch
        r.J_left, r.J_right, r.J_neck, r.J_com = (i[k].data_ptr() for k in ("J_left", "J_right", "J_neck", "J_com"))
        r.q, r.state = i["q"].data_ptr(), i["state"].data_ptr()
        r.dq, r.ik_status, r.active_lower, r.active_upper = o["dq"].data_ptr(), o["st"].data_ptr(), o["lo"].data_ptr(), o["up"].data_ptr()
        r.foot_err, r.iters, r.ik_stream = o["fe"].data_ptr(), o["it"].data_ptr(), streams[n % 2].cuda_stream
    assert wca.capi.qp_enqueue_steps(mpc, ik, B, recs) == 3
    torch.cuda.synchronize()
    for a, b in zip(ref_o, got_o):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert (got_o[1]["ms"] == -1).all() and (got_o[0]["ms"] == 0).all() and (got_o[2]["st"] == 0).sum() >= 0.99 * B - 1
    # a bad record stops the walk and reports how far it got
    recs[1].dq = None
    try:
        wca.capi.qp_enqueue_steps(mpc, ik, B, recs)
        raise AssertionError("a record without dq must be refused")
    except wca.WcqpError:
        pass


if __name__ == "__main__":
    for B in (1, 5, 777):                      # ragged: not a multiple of the 4 robots per wave
        for jac in (wca.IK_JAC_MIXED, wca.IK_JAC_AUTO):
            main(B, jac)
    print("enqueue_steps ok")

