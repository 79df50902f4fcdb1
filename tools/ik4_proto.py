"""
numpy prototype of the fourth IK kernel (csrc/ik4.hip): closed-form elimination of the six base unknowns through
the left-foot rows + range-space solve of the remaining 23-variable QP.  Checked here against oracle/qp_spec.py
before any HIP is written; the kernel follows this file step for step.

Structure used (iDynTree MIXED free-floating frame Jacobians, which is what WalkingFK hands
to the IK, WM/src/WalkingForwardKinematics.cpp:33,436-454): base blocks
    J_left = [I B_L; 0 I | .],  J_right = [I B_R; 0 I | .],  J_com = [I B_C | .],  J_neck(ang) = [0 I | .]
so  v_base = X_L^-1 (b_L - J_Lq x)  in closed form and, with x = joint velocities (23),
    min 1/2 x' Lam x + gq' x + 1/2 |Nt x - t|^2   s.t.  A x = b,  lo <= x <= hi           (Lam > 0 diagonal)
    A = [J_Rq - X_R X_L^-1 J_Lq ; J_Cq - X_C X_L^-1 J_Lq]  (9 x 23),  Nt = L'(J_nq - J_Lq,ang),  W = L L'.
Range space: C = [Nt; A] (12 x 23), D = Lam^-1, M = C D C' + diag(I3, 0) (SPD), M y = -(C D gq + [t; b]),
x = -D (gq + C' y).  Projected inverse Hessian P = D - D C' M^-1 C D gives the dual active set its columns.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import walking_controllers_amd as wca  # noqa: E402
from oracle import qp_spec as qs  # noqa: E402


def transform(p: qs.IKParams, x: qs.IKInputs, form: str):
    """-> (C 12x23, d 12, gq 23, D 23) or None when the base blocks do not have the MIXED pattern."""
    I3, Z3 = np.eye(3), np.zeros((3, 3))
    JL, JR, JC, JN = x.J_left, x.J_right, x.J_com, x.J_neck
    ok = (np.array_equal(JL[0:3, 0:3], I3) and np.array_equal(JL[3:6, 0:3], Z3) and np.array_equal(JL[3:6, 3:6], I3)
          and np.array_equal(JR[0:3, 0:3], I3) and np.array_equal(JR[3:6, 0:3], Z3) and np.array_equal(JR[3:6, 3:6], I3)
          and np.array_equal(JC[:, 0:3], I3) and np.array_equal(JN[:, 0:3], Z3) and np.array_equal(JN[:, 3:6], I3))
    if not ok or not p.use_com_as_constraint or (p.joint_reg_weights <= 0).any():
        return None
    BL, BR, BC = JL[0:3, 3:6], JR[0:3, 3:6], JC[:, 3:6]
    b = qs.ik_task_rhs(p, x, form)
    kappa = p.k_att_foot if form == "osqp" else 1.0
    e = kappa * (-p.k_neck * qs.rot_error(x.R_neck, x.Rd_neck))          # neck target (qp.cpp:164,175)
    # columns of [J_L; J_R; J_C; J_N | rhs]: joints + the rhs column [b_L; b_R; b_C; e]
    cols = np.hstack([np.vstack([JL[:, 6:], JR[:, 6:], JC[:, 6:], JN[:, 6:]]),
                      np.concatenate([b, e])[:, None]])                # 18 x 24
    lin, ang = cols[0:3], cols[3:6]
    AR = np.vstack([cols[6:9] - lin - (BR - BL) @ ang, cols[9:12] - ang])
    AC = cols[12:15] - lin - (BC - BL) @ ang
    N = cols[15:18] - ang
    L = np.linalg.cholesky(p.neck_weight)
    Ct = np.vstack([L.T @ N, AR, AC])                                  # 12 x 24
    gq = -(p.joint_reg_weights * p.joint_reg_gains) * (p.q_reg - x.q)
    return Ct[:, :23], Ct[:, 23], gq, 1.0 / p.joint_reg_weights


def solve(C, d, gq, D, lo, hi, tol=1e-12, max_iter=100, w0=None):
    """Range-space equality solve + Goldfarb-Idnani on columns of P.  Returns x, status, lower, upper, iters.
    w0 = (vars, signs): hot start - those bounds are tried as equalities first (D_p -> 0), accepted when every
    other bound holds and their multipliers are positive."""
    n = 23
    M = (C * D) @ C.T + np.diag([1.0] * 3 + [0.0] * 9)
    try:
        Lc = np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        return None, 4, [], [], 0
    Minv = np.linalg.inv(M)
    y = -Minv @ ((C * D) @ gq + d)
    x = -D * (gq + C.T @ y)
    if w0 is not None and len(w0[0]):
        W0, S0 = list(w0[0]), list(w0[1])
        Dh = D.copy(); Dh[W0] = 0.0
        v = np.zeros(n); v[W0] = [hi[p] if s > 0 else lo[p] for p, s in zip(W0, S0)]
        Mh = (C * Dh) @ C.T + np.diag([1.0] * 3 + [0.0] * 9)
        try:
            np.linalg.cholesky(Mh)
            yh = -np.linalg.solve(Mh, (C * Dh) @ gq + d - C @ v)
            xh = -Dh * (gq + C.T @ yh) + v
            mu = -(gq + C.T @ yh + v / D)          # multiplier of the fixed variables: Lam x + gq + C'y + mu = 0
            mu_s = np.array([mu[p] * s for p, s in zip(W0, S0)])
            if (mu_s > 0).all() and (xh <= hi + tol).all() and (xh >= lo - tol).all():
                return xh, 0, [p for p, s in zip(W0, S0) if s < 0], [p for p, s in zip(W0, S0) if s > 0], 0
        except np.linalg.LinAlgError:
            pass
    P = np.diag(D) - (D[:, None] * C.T) @ Minv @ (C * D)
    W, sg, mu, T = [], [], [], []
    it = 0
    status = 0
    while True:
        viol = np.maximum(x - hi, lo - x)
        viol[W] = -np.inf
        p = int(np.argmax(viol))
        if viol[p] <= tol:
            break
        if it >= max_iter:
            status = 1
            break
        it += 1
        sig = 1.0 if x[p] - hi[p] >= lo[p] - x[p] else -1.0
        s = viol[p]
        tp = sig * P[:, p]
        ppp = sig * tp[p]
        mu_p = 0.0
        done = False
        while True:
            k = len(W)
            if k:
                R = np.array([[sg[a] * T[b][W[a]] for b in range(k)] for a in range(k)])
                c = np.array([sg[a] * tp[W[a]] for a in range(k)])
                r = np.linalg.solve(R, c)
                z = tp - sum(r[a] * T[a] for a in range(k))
            else:
                r = np.zeros(0); z = tp
            nz = sig * z[p]
            t2 = s / nz if (k < n - 9 and nz > 1e-10 * ppp) else np.inf
            t1, jd = np.inf, -1
            for a in range(k):
                if r[a] > 0 and mu[a] / r[a] < t1:
                    t1, jd = mu[a] / r[a], a
            t = min(t1, t2)
            if not np.isfinite(t):
                status = 2; done = True
                break
            x = x - t * z
            for a in range(k):
                mu[a] -= t * r[a]
            mu_p += t
            s -= t * nz
            if t2 <= t1:
                W.append(p); sg.append(sig); mu.append(mu_p); T.append(tp)
                break
            W.pop(jd); sg.pop(jd); mu.pop(jd); T.pop(jd)
            it += 1
        if done:
            break
    lower = [w for w, s_ in zip(W, sg) if s_ < 0]
    upper = [w for w, s_ in zip(W, sg) if s_ > 0]
    if status == 0:
        for w, s_ in zip(W, sg):
            x[w] = hi[w] if s_ > 0 else lo[w]
    return x, status, lower, upper, it


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for form, vmax in (("qpoases", 0.5), ("qpoases", 0.22), ("osqp", 0.3)):
        b = wca.synth.synth_ik_batch(B, seed=99)
        p = qs.IKParams(v_max=vmax * np.ones(23))
        worst = 0.0; nact = 0; ninf = 0; hot_hits = 0; hot_tries = 0
        for i in range(B):
            xin = qs.ik_inputs_from_batch(b, i)
            tr = transform(p, xin, form)
            assert tr is not None
            C, d, gq, D = tr
            big = np.inf if form == "osqp" else vmax
            lo, hi = -big * np.ones(23), big * np.ones(23)
            x, st, lower, upper, it = solve(C, d, gq, D, lo, hi)
            try:
                r = qs.ik_exact(p, xin, form)
            except qs.QPInfeasible:
                assert st != 0, (i, st)
                ninf += 1
                continue
            assert st == 0, (i, st)
            worst = max(worst, np.abs(x - r["dq"]).max())
            if r["mu_min_active"] > 1e-7 and r["slack_min_inactive"] > 1e-7:
                assert sorted(lower) == r["lower"] and sorted(upper) == r["upper"], (i, lower, upper, r["lower"], r["upper"])
            nact += len(lower) + len(upper)
            # hot start from the exact active set must be accepted and give the same point
            if lower or upper:
                hot_tries += 1
                xh, sth, lh, uh, ith = solve(C, d, gq, D, lo, hi, w0=(lower + upper, [-1.0] * len(lower) + [1.0] * len(upper)))
                if ith == 0 and sth == 0:
                    hot_hits += 1
                assert np.abs(xh - x).max() < 1e-9, (i, np.abs(xh - x).max())
        print(f"{form} v_max={vmax}: max |dq - exact| = {worst:.2e}, active bounds {nact}, infeasible {ninf}, hot-start hits {hot_hits}/{hot_tries}")


if __name__ == "__main__":
    main()
