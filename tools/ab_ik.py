"""A/B of the two IK kernels in one process, interleaved rounds (cdna guide rule 24)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import walking_controllers_amd as wca
dev = torch.device("cuda", 0)
res = {}
for B in (4096, 65536):
    ib = wca.synth.synth_ik_batch(B, seed=4321)
    d = {k: torch.from_numpy(ib[k]).to(dev) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")}
    outs = {}
    for vmax in (0.5, 100.0, 0.3):
        solvers = {a: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=a) for a in (wca.IK_ALG_SWEEP, wca.IK_ALG_NULLSPACE, wca.IK_ALG_NULLSPACE_MFMA)}
        bufs = {a: (torch.zeros(B, 23, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev)) for a in solvers}
        sp = torch.cuda.current_stream().cuda_stream
        def run(a):
            dq, st, lo, up, it = bufs[a]
            solvers[a].solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(), d["J_com"].data_ptr(),
                                    d["q"].data_ptr(), d["state"].data_ptr(), dq.data_ptr(), st.data_ptr(), lo.data_ptr(), up.data_ptr(), 0, it.data_ptr(), sp)
        times = {a: [] for a in solvers}
        for a in solvers:
            for _ in range(3): run(a)
        torch.cuda.synchronize()
        for rnd in range(7):
            for a in solvers:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run(a)
                e1.record(); torch.cuda.synchronize()
                times[a].append(e0.elapsed_time(e1) / 10)
        a1, a2, a3 = wca.IK_ALG_SWEEP, wca.IK_ALG_NULLSPACE, wca.IK_ALG_NULLSPACE_MFMA
        ok = (bufs[a1][1] == 0) & (bufs[a2][1] == 0) & (bufs[a3][1] == 0)
        diff = float((bufs[a1][0] - bufs[a2][0])[ok].abs().max())
        diff3 = float((bufs[a3][0] - bufs[a2][0])[ok].abs().max())
        same_sets = bool(((bufs[a1][2] == bufs[a2][2]) & (bufs[a1][3] == bufs[a2][3]) & (bufs[a3][2] == bufs[a2][2]) & (bufs[a3][3] == bufs[a2][3]))[ok].all())
        res[f"B{B}_vmax{vmax}"] = {"sweep_ms_median": float(np.median(times[a1])), "sweep_ms_min": float(np.min(times[a1])),
                                   "nullspace_ms_median": float(np.median(times[a2])), "nullspace_ms_min": float(np.min(times[a2])),
                                   "nullspace_mfma_ms_median": float(np.median(times[a3])), "nullspace_mfma_ms_min": float(np.min(times[a3])),
                                   "max_abs_diff": diff, "max_abs_diff_mfma_vs_valu": diff3, "same_active_sets": same_sets, "all_solved": int(ok.sum()),
                                   "status_mismatch": int((bufs[a1][1] != bufs[a2][1]).sum() + (bufs[a3][1] != bufs[a2][1]).sum())}
print(json.dumps(res, indent=1))
