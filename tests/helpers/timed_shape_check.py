"""Run by tests/test_gpu_parity.py::test_timed_launch_shape_against_goldens in a process of its own (torch first, then
libwcqp): 12 records at the BASELINE batch over three streams sharing one handle pair, in one wcqp_qp_enqueue_steps call,
every record against the golden vectors."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import walking_controllers_amd as wca

MARGIN = 1e-7


def main():
    B, K, P, R = 4096, 4, 3, 12
    dev = torch.device("cuda", 0)
    mb, ib = wca.synth.synth_mpc_batch(B, seed=1234), wca.synth.synth_ik_batch(B, seed=4321)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base = {k: t(mb[k]) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")}
    base.update({k: t(ib[k]) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")})
    sets = [base] + [{k: torch.roll(v, shifts=j * (B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
    outs = [dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), ms=torch.full((B,), -1, dtype=torch.int32, device=dev),
                 ma=torch.zeros(B, dtype=torch.int32, device=dev), mm=torch.zeros(B, dtype=torch.float64, device=dev),
                 dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), st=torch.full((B,), -1, dtype=torch.int32, device=dev),
                 lo=torch.zeros(B, dtype=torch.int32, device=dev), up=torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(R)]
    streams = [torch.cuda.Stream(dev) for _ in range(P)]
    mpc, ik = wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5, jacobian_structure=wca.IK_JAC_MIXED)
    recs = (wca.capi.QpStep * R)()
    for n in range(R):
        d, o, sp, r = sets[n % K], outs[n], streams[n % P].cuda_stream, recs[n]
        r.x0, r.ref, r.ref_len, r.u_prev = d["x0"].data_ptr(), d["ref"].data_ptr(), d["ref"].shape[1], d["u_prev"].data_ptr()
        r.hull_A, r.hull_b, r.hull_nc = d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr()
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin, r.mpc_stream = o["u0"].data_ptr(), o["ms"].data_ptr(), o["ma"].data_ptr(), o["mm"].data_ptr(), sp
        r.J_left, r.J_right, r.J_neck, r.J_com = (d[k].data_ptr() for k in ("J_left", "J_right", "J_neck", "J_com"))
        r.q, r.state, r.dq, r.ik_status = d["q"].data_ptr(), d["state"].data_ptr(), o["dq"].data_ptr(), o["st"].data_ptr()
        r.active_lower, r.active_upper, r.ik_stream = o["lo"].data_ptr(), o["up"].data_ptr(), sp
    torch.cuda.synchronize()
    assert wca.capi.qp_enqueue_steps(mpc, ik, B, recs) == R
    torch.cuda.synchronize()
    gd = os.path.join(ROOT, "tests", "golden")
    gm = np.load(os.path.join(gd, "mpc_cfg2_b4096.npz"), allow_pickle=False)
    gi = np.load(os.path.join(gd, "ik_qpoases_v050_b4096.npz"), allow_pickle=False)
    sure_m = (gm["mu_min_active"] > MARGIN) & (gm["slack_min_inactive"] > MARGIN)
    sure_i = (gi["mu_min_active"] > MARGIN) & (gi["slack_min_inactive"] > MARGIN) & (gi["status"] == 0)
    o0 = outs[0]["dq"].cpu().numpy()
    for n in range(R):
        inst = (np.arange(B) - (n % K) * (B // K)) % B          # output row -> instance of the unrotated batch
        o = {k: v.cpu().numpy() for k, v in outs[n].items()}
        assert (o["ms"] == 0).all() and np.abs(o["u0"] - gm["u0"][inst]).max() <= 1e-9, n
        assert np.array_equal(o["ma"].astype(np.uint32)[sure_m[inst]], gm["active"][inst][sure_m[inst]]), n
        m = inst < int(gi["count"])
        gidx = inst[m]
        assert np.array_equal(o["st"][m], gi["status"][gidx]), n
        assert np.abs(o["dq"][m] - gi["dq"][gidx]).max() <= 1e-9, n
        s_ = sure_i[gidx]
        assert np.array_equal(o["lo"][m].astype(np.uint32)[s_], gi["active_lower"][gidx][s_]), n
        assert np.array_equal(o["up"][m].astype(np.uint32)[s_], gi["active_upper"][gidx][s_]), n
        # the instances the IK golden does not hold: bit-identical to the same instances of the unrotated record 0
        assert np.array_equal(o["dq"], o0[inst]), n


if __name__ == "__main__":
    main()
    print("timed shape ok")
