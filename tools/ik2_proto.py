"""
Numpy prototype of the null-space formulation planned for the second IK kernel (design aid):
  1. Gauss-Jordan on [A | b] with column pivoting  -> basic set B (meq vars), x_B = b' - F x_N
  2. reduced Hessian  Hr = Z'HZ = D_N + F' D_B F + (N Z)' W (N Z)   (nN = n - meq = 14 free vars)
  3. sweep inverse of Hr, x_N = -Hinv g_r, x_B from step 1
  4. Goldfarb-Idnani over the joint bounds with full-space columns tau_p = Z Hinv Z' e_p
Every step is written the way the kernel will do it (per-"lane" columns, compact indices).
"""
import sys, importlib.util
import numpy as np

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from oracle import qp_spec as qs  # noqa: E402
from ik_proto import sweep_inverse  # noqa: E402


def ik2_solve(D, N_rows, W, g, A, b, lb, ub, max_iter=100, tol=1e-12):
    """H = diag(D) + N' W N (never formed).  Returns nu, lower, upper, status, iters."""
    n = A.shape[1]; meq = A.shape[0]; nN = n - meq
    a = np.hstack([A, b[:, None]]).copy()        # column j = lane j, last column = rhs lane
    basic_row = -np.ones(n, int)                 # for each variable: the row it is basic in
    piv_col = np.zeros(meq, int)
    for r in range(meq):
        cand = np.where(basic_row < 0, np.abs(a[r, :n]), -1.0)
        p = int(np.argmax(cand))
        if cand[p] < 1e-12:
            return None, [], [], 4, 0
        basic_row[p] = r; piv_col[r] = p
        colp = a[:, p].copy()
        t = a[r, :] / colp[r]
        a[r, :] = t
        for rr in range(meq):
            if rr != r:
                a[rr, :] -= colp[rr] * t
    nonb = [j for j in range(n) if basic_row[j] < 0]           # compact order = lane order
    F = a[:, nonb]                                             # meq x nN
    bp = a[:, n]
    dB = D[piv_col]; gB = g[piv_col]; nB = N_rows[:, piv_col]  # per row r
    # reduced cost rows (lane j nonbasic): nz_j = n_j - nB F_j ; rhs lane: -N x_p
    nz = N_rows[:, nonb] - nB @ F
    nz_rhs = -nB @ bp
    wnz = W @ nz
    Hr = np.diag(D[nonb]) + F.T @ (dB[:, None] * F) + nz.T @ wnz
    h_rhs = F.T @ (dB * bp) + wnz.T @ nz_rhs                   # "column 14": the b'-dependent part
    g_r = g[nonb] - F.T @ gB - h_rhs
    Hinv = sweep_inverse(Hr)
    xN = -Hinv @ g_r
    nu = np.zeros(n)
    nu[nonb] = xN
    nu[piv_col] = bp - F @ xN

    def tau(p):                                                # full-space column Z Hinv Z' e_p
        if basic_row[p] < 0:
            t = Hinv[:, nonb.index(p)]
        else:
            t = -Hinv @ F[basic_row[p], :]
        out = np.zeros(n)
        out[nonb] = t
        out[piv_col] = -F @ t
        return out

    W_, sg, mu, T = [], [], [], []
    status, it = 0, 0
    while True:
        viol = np.maximum(nu - ub, lb - nu); viol[:6] = -np.inf
        for w in W_:
            viol[w] = -np.inf
        p = int(np.argmax(viol)); s = viol[p]
        if not (s > tol):
            break
        if it >= max_iter:
            status = 1; break
        it += 1
        sig = 1.0 if nu[p] - ub[p] >= lb[p] - nu[p] else -1.0
        tp = sig * tau(p); mu_p = 0.0; ppp = sig * tp[p]
        while True:
            k = len(W_)
            if k:
                R = np.array([[sg[a_] * T[b_][W_[a_]] for b_ in range(k)] for a_ in range(k)])
                c = np.array([sg[a_] * tp[W_[a_]] for a_ in range(k)])
                r = np.linalg.solve(R, c)
                z = tp - sum(r[a_] * T[a_] for a_ in range(k))
            else:
                r = np.zeros(0); z = tp
            nzv = sig * z[p]
            t2 = s / nzv if (k < nN and nzv > 1e-10 * ppp) else np.inf
            t1, jd = np.inf, -1
            for a_ in range(k):
                if r[a_] > 0 and mu[a_] / r[a_] < t1:
                    t1, jd = mu[a_] / r[a_], a_
            t = min(t1, t2)
            if not (t < np.inf):
                status = 2; break
            nu = nu - t * z
            for a_ in range(k):
                mu[a_] -= t * r[a_]
            mu_p += t; s -= t * nzv
            if t2 <= t1:
                W_.append(p); sg.append(sig); mu.append(mu_p); T.append(tp); break
            W_.pop(jd); sg.pop(jd); mu.pop(jd); T.pop(jd); it += 1
        if status:
            break
    lower = sorted(w - 6 for w, s_ in zip(W_, sg) if s_ < 0)
    upper = sorted(w - 6 for w, s_ in zip(W_, sg) if s_ > 0)
    return nu, lower, upper, status, it


def run_case(p, x, form):
    n = p.dof + 6
    D = np.zeros(n); D[6:] = p.joint_reg_weights
    if p.use_com_as_constraint:
        N_rows, W = x.J_neck, p.neck_weight
    else:
        N_rows = np.vstack([x.J_com, x.J_neck])
        W = np.block([[p.com_weight, np.zeros((3, 3))], [np.zeros((3, 3)), p.neck_weight]])
    g = qs.ik_gradient(p, x, form)
    A = qs.ik_task_matrix(p, x); b = qs.ik_task_rhs(p, x, form)
    big = np.finfo(float).max
    if form == "qpoases":
        lb = np.concatenate([-big * np.ones(6), -p.v_max]); ub = -lb
    else:
        lb = -big * np.ones(n); ub = big * np.ones(n)
    return ik2_solve(D, N_rows, W, g, A, b, lb, ub)


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("synth", "/root/repo/walking-controllers_amd/synth.py")
    synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
    B = 128
    ib = synth.synth_ik_batch(B)
    for use_com in (True, False):
        for form, vm in (("qpoases", 1.0), ("qpoases", 0.5), ("qpoases", 0.35), ("qpoases", 0.25), ("osqp", 0.3)):
            p = qs.IKParams(v_max=vm * np.ones(23), use_com_as_constraint=use_com)
            err, mis, itmax, st, nf = 0.0, 0, 0, 0, 0
            for i in range(B):
                x = qs.ik_inputs_from_batch(ib, i)
                nu, lo, up, status, it = run_case(p, x, form)
                st += status != 0; itmax = max(itmax, it)
                try:
                    r = qs.ik_exact(p, x, form)
                except qs.QPOracleError:
                    nf += 1; continue
                if status == 0:
                    err = max(err, np.abs(nu - r["nu"]).max())
                    mis += (lo != r["lower"]) or (up != r["upper"])
            print(f"use_com {use_com} {form} vmax {vm}: max err {err:.2e} set mismatches {mis} maxit {itmax} status!=0 {st} oracle-fail {nf}")
    # iCub-shaped degenerate case: the stance foot Jacobian touches the base columns only
    p = qs.IKParams(v_max=0.5 * np.ones(23))
    err = 0
    for i in range(32):
        x = qs.ik_inputs_from_batch(ib, i)
        x.J_left = x.J_left.copy(); x.J_left[:, 6:] = 0.0
        nu, lo, up, status, it = run_case(p, x, "qpoases")
        r = qs.ik_exact(p, x, "qpoases")
        err = max(err, np.abs(nu - r["nu"]).max()); assert status == 0
    print("stance-foot-only-base Jacobian: max err %.2e" % err)
