"""
CPU tests (-m "not gpu"): the oracle against hand-computable cases, the committed golden
vectors, and its own C restatement.  The reference has no tests or fixtures for this path
("parity unpinned"), so the anchors are the formulas of SURVEY Appendix A
checked on tiny cases and the KKT certificate of the exact optimum.
"""
import os

import numpy as np
import pytest


def test_mpc_blocks_tiny_case(qs):
    """N = 2, isotropic weights: P, A_eq, gradient sub-matrix entry by entry
    (WalkingDCMModelPredictiveController.cpp:23-168)."""
    p = qs.MPCParams(horizon=2, Q=3.0 * np.eye(2), R=5.0 * np.eye(2))
    c = qs.mpc_constants(p)
    assert (c.n, c.n_x, c.n_u) == (10, 6, 4)
    th = qs.mpc_theta(2)
    assert np.array_equal(th, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [-1, 0, 1, 0], [0, -1, 0, 1.0]]))
    # input cost = |u0 - u_-1|^2_R + |u1 - u0|^2_R  ->  diag 2r, last block r, off-diagonal -r
    Pu = c.P[6:, 6:]
    assert np.array_equal(Pu, 5.0 * np.array([[2, 0, -1, 0], [0, 2, 0, -1], [-1, 0, 1, 0], [0, -1, 0, 1.0]]))
    assert np.array_equal(c.P[:6, :6], 3.0 * np.eye(6)) and not c.P[:6, 6:].any()
    assert np.array_equal(c.grad_sub, -5.0 * np.array([[1, 0], [0, 1], [0, 0], [0, 0.0]]))
    a = np.exp(np.sqrt(9.81 / 0.53) * 0.01)
    assert c.a == pytest.approx(a, rel=1e-15) and c.b == pytest.approx(1 - a, rel=1e-15)
    # rows 2..3: -x1 + a x0 + b u0 = 0
    row = c.A_eq[2]
    assert row[0] == c.a and row[2] == -1.0 and row[6] == c.b and np.count_nonzero(row) == 3
    assert np.count_nonzero(c.A_eq) == 6 * 2 + 2          # nnz = 6N + 2


def test_mpc_sizes_at_baseline_horizon(qs):
    c = qs.mpc_constants(qs.MPCParams())
    assert (c.n, c.n_x) == (202, 102)
    assert np.count_nonzero(c.P) == 398 and np.count_nonzero(c.A_eq) == 302
    assert c.a == pytest.approx(1.043961, abs=1e-6)
    assert np.linalg.eigvalsh(c.P).min() == pytest.approx(7500.0, rel=1e-9)


def test_mpc_gradient_shift_matches_full_rebuild(qs):
    """MPCSolver::setGradient: shift-by-one + last stage == full rebuild when the deque
    advanced exactly one stage (MPCSolver.cpp:216-239), and short deques are padded."""
    c = qs.mpc_constants(qs.MPCParams(horizon=6))
    rng = np.random.default_rng(0)
    ref = rng.normal(size=(9, 2))
    up = rng.normal(size=2)
    q0 = qs.mpc_gradient(c, ref, up)
    q1_shift = qs.mpc_gradient(c, ref[1:], up, q_prev=q0, reset=False)
    q1_full = qs.mpc_gradient(c, ref[1:], up)
    assert np.array_equal(q1_shift, q1_full)
    short = qs.mpc_gradient(c, ref[:3], up)
    assert np.array_equal(short[4:6], short[12:14]) and np.array_equal(short[4:6], -c.Q @ ref[2])
    assert np.array_equal(q0[c.n_x:c.n_x + 2], -c.R @ up) and not q0[c.n_x + 2:].any()


def test_mpc_assemble_layout(qs):
    c = qs.mpc_constants(qs.MPCParams(horizon=3))
    hA = np.array([[1.0, 0], [0, 1], [-1, 0], [0, -1]])
    hb = np.array([0.05, 0.025, 0.02, 0.025])
    P, q, A, l, u = qs.mpc_assemble(c, [0.01, -0.02], np.zeros((4, 2)), [0, 0], hA, hb)
    assert A.shape == (c.n_x + 4, c.n) and np.array_equal(A[c.n_x:, c.n_x:c.n_x + 2], hA)
    assert not A[c.n_x:, :c.n_x].any() and not A[c.n_x:, c.n_x + 2:].any()      # only u0 is hull-constrained
    assert np.array_equal(l[:2], [-0.01, 0.02]) and np.array_equal(u[:2], l[:2])
    assert (l[c.n_x:] == -1e30).all() and np.array_equal(u[c.n_x:], hb)


def test_config1_single_mpc_on_cpu(qs):
    """BASELINE configs[0]: one MPC QP, N = 50, single support at identity, plumbing on CPU."""
    c = qs.mpc_constants(qs.MPCParams())
    hA = np.array([[1.0, 0], [0, 1], [-1, 0], [0, -1]])
    hb = np.array([0.05, 0.025, 0.02, 0.025])
    ref = np.stack([0.002 * np.arange(51), np.zeros(51)], 1)
    r = qs.mpc_exact(c, [0.01, -0.005], ref, [0.0, 0.0], hA, hb)
    z = r["z"]
    assert np.abs(c.A_eq @ z - np.concatenate([[-0.01, 0.005], np.zeros(100)])).max() < 1e-12
    assert r["margin"] >= -1e-12 and r["u0"].shape == (2,)
    # a previous ZMP far ahead of the foot drags u0 onto the hull (R >> Q): the active row
    # carries a positive multiplier
    r2 = qs.mpc_exact(c, [0.01, -0.005], ref, [0.2, 0.0], hA, hb)
    assert r2["active"] == [0] and r2["u0"][0] == pytest.approx(0.05, abs=1e-12) and r2["mu"][0] > 0


def test_rot_error_small_angle(qs):
    """unskew(0.5 (R Rd' - Rd R')) ~ rotation vector for small angles (Utils.cpp:22-27)."""
    w = np.array([0.01, -0.02, 0.015])
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + K + 0.5 * K @ K
    e = qs.rot_error(R, np.eye(3))
    assert np.abs(e - w).max() < 1e-5
    assert np.abs(qs.rot_error(R, R)).max() < 1e-15


def test_ik_forms_differ_as_documented(qs, wca):
    """Appendix B-13/14/15: osqp form has m = 38 with 23 zero rows, kappa = k_attFoot on the
    neck term and the zero-twist rule; qpOASES form has variable bounds and kappa = 1."""
    b = wca.synth.synth_ik_batch(4, seed=1)
    p = qs.IKParams()
    x = qs.ik_inputs_from_batch(b, 0)
    P, q, A, l, u = qs.ik_assemble_osqp(p, x)
    assert A.shape == (38, 29) and not A[15:].any() and np.array_equal(l[:15], u[:15])
    assert np.array_equal(l[15:], -np.ones(23))
    H, g, Aq, lb, ub, lbA, ubA = qs.ik_assemble_qpoases(p, x)
    assert np.array_equal(H, P) and Aq.shape == (15, 29) and np.array_equal(lbA, ubA)
    assert np.linalg.matrix_rank(H) == 26                     # rank 26/29
    assert (ub[:6] == np.finfo(float).max).all() and np.array_equal(ub[6:], np.ones(23))
    # gradient: neck term differs by exactly k_attFoot (= 2)
    g_reg = np.zeros(29)
    g_reg[6:] = -p.joint_reg_weights * p.joint_reg_gains * (p.q_reg - x.q)
    assert np.allclose(q - g_reg, 2.0 * (g - g_reg), rtol=0, atol=1e-14)
    # zero-twist rule: one foot of every synthetic robot is in stance (twist == 0)
    stance_left = not x.twist_left.any()
    rows = slice(0, 6) if stance_left else slice(6, 12)
    assert not l[rows].any() and lbA[rows].any()


@pytest.mark.parametrize("name", ["mpc_cfg2_b4096.npz", "mpc_stress_b1024.npz"])
def test_oracle_reproduces_mpc_golden(qs, wca, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    kw = {"uprev_sigma": 0.04} if "stress" in name else {}
    n = 96
    b = wca.synth.synth_mpc_batch(n, seed=int(g["seed"]), **kw)
    assert np.array_equal(b["x0"][:4], g["in_x0"]) and np.array_equal(b["hull_A"][:4], g["in_hull_A"])
    c = qs.mpc_constants(qs.MPCParams())
    for i in range(n):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert np.abs(r["u0"] - g["u0"][i]).max() < 1e-13
        assert sum(1 << e for e in r["active"]) == int(g["active"][i])


@pytest.mark.parametrize("name", ["ik_qpoases_v050_b1024.npz", "ik_qpoases_v050_b4096.npz", "ik_qpoases_v030_b512.npz", "ik_osqp_b512.npz"])
def test_oracle_reproduces_ik_golden(qs, wca, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    n = int(g["count"])
    b = wca.synth.synth_ik_batch(n, seed=int(g["seed"]))
    assert np.array_equal(b["q"][:2], g["in_q"]) and np.array_equal(b["J_com"][:2], g["in_J_com"])
    p = qs.IKParams(v_max=float(g["v_max"]) * np.ones(23))
    for i in range(0, n, max(1, n // 96)):              # rows from the whole file, not only its head
        r = qs.ik_exact(p, qs.ik_inputs_from_batch(b, i), str(g["form"]))
        assert np.abs(r["dq"] - g["dq"][i]).max() < 1e-12
        assert sum(1 << j for j in r["lower"]) == int(g["active_lower"][i])
        assert sum(1 << j for j in r["upper"]) == int(g["active_upper"][i])


def test_exact_oracle_flags_infeasible(qs, wca):
    b = wca.synth.synth_ik_batch(1, seed=2)
    with pytest.raises(qs.QPInfeasible):
        qs.ik_exact(qs.IKParams(v_max=1e-3 * np.ones(23)), qs.ik_inputs_from_batch(b, 0), "qpoases")


# ---- the C restatement (OSQP algorithm, dense active set) against the exact optimum -------
def test_c_osqp_restatement_reaches_the_optimum_within_its_eps(qs, wca):
    """Secondary parity (SURVEY §7 'hard parts'): the OSQP-default-settings restatement agrees
    with the exact optimum within OSQP's own eps = 1e-3, and to 1e-7 when run tight."""
    from oracle import c_oracle as co
    mp = qs.MPCParams()
    c = qs.mpc_constants(mp)
    b = wca.synth.synth_mpc_batch(24, seed=77, uprev_sigma=0.04)
    u0, iters, status = co.mpc_batch_osqp(mp, b, nthreads=2)
    assert (status == 0).all() and iters.max() <= 4000
    ex = np.array([qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i],
                                int(b["hull_nc"][i]))["u0"] for i in range(24)])
    assert np.abs(u0 - ex).max() < 1e-3
    P, q, A, l, u = qs.mpc_assemble(c, b["x0"][0], b["ref"][0], b["u_prev"][0],
                                    b["hull_A"][0][:b["hull_nc"][0]], b["hull_b"][0][:b["hull_nc"][0]])
    x, it, rc = co.osqp_dense(P, q, A, l, u, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000)
    assert rc == 0 and np.abs(x[102:104] - ex[0]).max() < 1e-7


@pytest.mark.parametrize("form,vmax", [("qpoases", 0.5), ("qpoases", 0.3), ("osqp", 1.0)])
def test_c_ik_restatement(qs, wca, form, vmax):
    from oracle import c_oracle as co
    b = wca.synth.synth_ik_batch(48, seed=4321)
    ip = qs.IKParams(v_max=vmax * np.ones(23))
    dq, status, lo, up, iters = co.ik_batch(ip, b, form, nthreads=2)
    assert (status == 0).all()
    tol = 1e-3 if form == "osqp" else 1e-12
    for i in range(48):
        r = qs.ik_exact(ip, qs.ik_inputs_from_batch(b, i), form)
        assert np.abs(dq[i] - r["dq"]).max() < tol
        if form == "qpoases":
            assert int(lo[i]) == sum(1 << j for j in r["lower"]) and int(up[i]) == sum(1 << j for j in r["upper"])


def test_c_oracle_is_clean_under_asan_ubsan():
    """Sanitizers run on the CPU build only (GPU ASan is not available on this pool): the C
    restatement that parity and the CPU baseline rest on must itself be memory- and UB-clean."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1", OMP_NUM_THREADS="2")
    r = subprocess.run([os.path.join(root, "oracle", "_build", "selftest_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "mpc fail 0" in r.stdout and "ik form 0 fail 0" in r.stdout and "ik form 1 fail 0" in r.stdout, r.stdout


@pytest.mark.parametrize("seed,i,vmax", [(31010, 142, 0.2), (31010, 165, 0.2), (31012, 181, 0.15)])
def test_exact_oracle_says_infeasible_when_its_walk_ends_on_a_singular_set(qs, wca, seed, i, vmax):
    """Found by tools/fuzz_vs_oracle.py: on these infeasible instances the oracle's primal walk ends on a working set whose KKT
    system is singular and returns a least-squares point that fails the certificate.  That used to surface as QPOracleError
    ('could not certify') although the kernels answer WCQP_STATUS_INFEASIBLE; the certificate failure now falls back to the
    phase-1 LP + Goldfarb-Idnani path, which says what the instance is."""
    b = wca.synth.synth_ik_batch(256, seed=seed)
    with pytest.raises(qs.QPInfeasible):
        qs.ik_exact(qs.IKParams(v_max=vmax * np.ones(23)), qs.ik_inputs_from_batch(b, i), "qpoases")


def test_same_algorithm_cpu_legs_match_the_goldens(qs, wca, golden_dir):
    """oracle/wc_oracle.c part (4): the device kernels' own direct methods in plain C (condensed MPC + 2-D projection; base-eliminated
    range-space IK + dual active set) - what bench.py times as cpu_baseline.same_algorithm_qps - against the exact optimum."""
    from oracle import c_oracle as co
    mp = qs.MPCParams()
    gains = co.mpc_condensed_gains(mp)
    for name, kw in (("mpc_cfg2_b4096.npz", {}), ("mpc_stress_b1024.npz", {"uprev_sigma": 0.04})):
        g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
        b = wca.synth.synth_mpc_batch(int(g["count"]), seed=int(g["seed"]), **kw)
        u0, act, st = co.mpc_batch_condensed(mp, b, gains, nthreads=2)
        assert (st == 0).all() and np.abs(u0 - g["u0"]).max() <= 1e-12
        cc = (g["mu_min_active"] > 1e-7) & (g["slack_min_inactive"] > 1e-7)
        assert np.array_equal(act[cc], g["active"][cc])
    for name in ("ik_qpoases_v050_b4096.npz", "ik_qpoases_v030_b512.npz", "ik_osqp_b512.npz"):
        g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
        b = wca.synth.synth_ik_batch(int(g["count"]), seed=int(g["seed"]))
        p = qs.IKParams(v_max=float(g["v_max"]) * np.ones(23))
        dq, st, lo, up, it = co.ik_batch_range_space(p, b, str(g["form"]), nthreads=2)
        ok = g["status"] == 0
        assert np.array_equal(st == 0, ok) and (st[~ok] == 2).all()
        assert np.abs(dq[ok] - g["dq"][ok]).max() <= 1e-10
        cc = (g["mu_min_active"] > 1e-7) & (g["slack_min_inactive"] > 1e-7) & ok
        assert np.array_equal(lo[cc], g["active_lower"][cc]) and np.array_equal(up[cc], g["active_upper"][cc])


# ---- the solver restatements against the third-party libraries' PUBLISHED example problems --------
# The reference holds no fixture, and the libraries its two QPs are handed to (OSQP through osqp-eigen:
# WM/src/MPCSolver.cpp:68-80, WM/src/WalkingQPInverseKinematics_osqp.cpp:27-40; qpOASES: WM/src/WalkingQPInverseKinematics_qpOASES.cpp:
# 303-335) are not vendored.  What those libraries publish with a stated answer are their own first examples; the oracle's solvers
# (the exact active-set walk with its KKT certificate, the C restatement of the OSQP iteration) must reproduce them.  This pins the
# SOLVERS the goldens come from to an answer that is not this repository's own; it says nothing about the reference's assembly.
def test_solvers_on_the_osqp_documentation_demo(qs):
    """OSQP's setup-and-solve demo (the problem every language binding's README solves):
        min 1/2 x'[[4,1],[1,2]]x + [1,1]'x   s.t.  1 <= x1 + x2 <= 1,  0 <= x1 <= 0.7,  0 <= x2 <= 0.7
    published optimum x = (0.3, 0.7), objective 1.88."""
    from oracle import c_oracle as co
    P = np.array([[4.0, 1.0], [1.0, 2.0]]); q = np.array([1.0, 1.0])
    A = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]); l = np.array([1.0, 0.0, 0.0]); u = np.array([1.0, 0.7, 0.7])
    x, it, rc = co.osqp_dense(P, q, A, l, u)                                  # library defaults: eps 1e-3 on the RESIDUALS, 25 iterations
    assert rc == 0 and np.abs(x - [0.3, 0.7]).max() < 3e-3 and abs(0.5 * x @ P @ x + q @ x - 1.88) < 1e-3
    x, it, rc = co.osqp_dense(P, q, A, l, u, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000)
    assert rc == 0 and np.abs(x - [0.3, 0.7]).max() < 1e-8
    assert abs(0.5 * x @ P @ x + q @ x - 1.88) < 1e-8
    Ain = np.vstack([A[1:], -A[1:]]); bin_ = np.concatenate([u[1:], -l[1:]])
    xe, lam, mu, act = qs.qp_exact(P, q, A[:1], l[:1], Ain, bin_)
    assert np.abs(xe - [0.3, 0.7]).max() < 1e-13 and act == [1]               # x2 at its upper bound


def test_solvers_on_the_qpoases_manual_example(qs):
    """qpOASES' example1 (examples/example1.cpp, the manual's first QP):
        H = diag(1, 0.5), g = (1.5, 1), 0.5 <= x1 <= 5, -2 <= x2 <= 2, -1 <= x1 + x2 <= 2
    published: x = (0.5, -1.5), objective -6.25e-02; and its hot-started second QP g = (1, 1.5), 0 <= x1 <= 5, -1 <= x2 <= 2,
    -2 <= x1 + x2 <= 1: x = (0, -1), objective -1.25."""
    from oracle import c_oracle as co
    H = np.diag([1.0, 0.5]); A = np.array([[1.0, 1.0]])
    I2 = np.eye(2)
    for g, lb, ub, lbA, ubA, xs, obj in ((np.array([1.5, 1.0]), [0.5, -2.0], [5.0, 2.0], -1.0, 2.0, [0.5, -1.5], -6.25e-2),
                                         (np.array([1.0, 1.5]), [0.0, -1.0], [5.0, 2.0], -2.0, 1.0, [0.0, -1.0], -1.25)):
        Ain = np.vstack([I2, -I2, A, -A]); bin_ = np.array([ub[0], ub[1], -lb[0], -lb[1], ubA, -lbA])
        x, lam, mu, act = qs.qp_exact(H, g, np.zeros((0, 2)), np.zeros(0), Ain, bin_)
        assert np.abs(x - xs).max() < 1e-13 and abs(0.5 * x @ H @ x + g @ x - obj) < 1e-13
        xo, it, rc = co.osqp_dense(H, g, np.vstack([I2, A]), np.array([lb[0], lb[1], lbA]), np.array([ub[0], ub[1], ubA]),
                                   eps_abs=1e-10, eps_rel=1e-10, max_iter=20000)
        assert rc == 0 and np.abs(xo - xs).max() < 1e-7


def _slsqp(P, q, Aeq, beq, Ain, bin_, lo=None, hi=None, maxiter=400):
    """The QP through scipy's SLSQP (Kraft's Fortran code: a third implementation, none of this repository's), objective scaled to O(1)."""
    from scipy.optimize import minimize
    n = P.shape[0]
    sc = 1.0 / np.abs(P).max()
    cons = [dict(type="eq", fun=lambda v: Aeq @ v - beq, jac=lambda v: Aeq)]
    if len(bin_):
        cons.append(dict(type="ineq", fun=lambda v: bin_ - Ain @ v, jac=lambda v: -Ain))
    bounds = None if lo is None else [(None if not np.isfinite(a) else a, None if not np.isfinite(b) else b) for a, b in zip(lo, hi)]
    return minimize(lambda v: sc * (0.5 * v @ P @ v + q @ v), np.zeros(n), jac=lambda v: sc * (P @ v + q), method="SLSQP",
                    constraints=cons, bounds=bounds, options=dict(ftol=1e-20, maxiter=maxiter)).x


def test_exact_oracle_against_scipy_slsqp_on_assembled_qps(qs, wca):
    """An INDEPENDENT solver on the QPs the oracle assembles from the reference's formulas: scipy.optimize's SLSQP reaches the optimum
    `qp_exact` certifies - MPC (202 variables, 102 dynamics rows, support-polygon rows; iCubGenova04's Q / R among the cases) to 1e-9 in u0,
    IK (29 variables, 15 task rows, joint-velocity bounds with several of them active) to 1e-8 in dq.  Pins the oracle's SOLVER on real
    instances (the published examples above are two-variable problems); the assembly is still this repository's reading of the source."""
    import robots as rb
    for robot, seed in (("iCubGazeboV2_5", 77), ("iCubGenova04", 78)):
        mp = rb.mpc_params(qs, robot)
        c = qs.mpc_constants(mp)
        b = wca.synth.synth_mpc_batch(3, seed=seed, uprev_sigma=0.06, x0_sigma=0.03)
        for i in range(3):
            nc = int(b["hull_nc"][i])
            ex = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], nc)
            P, q, A, l, u = qs.mpc_assemble(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i][:nc], b["hull_b"][i][:nc])
            eq = np.isclose(l, u)
            x = _slsqp(P, q, A[eq], l[eq], A[~eq], u[~eq])
            assert np.abs(x[102:104] - ex["u0"]).max() < 1e-9, (robot, i)
    b = wca.synth.synth_ik_batch(4, seed=4321)
    ip = qs.IKParams(v_max=0.3 * np.ones(23))
    n_active = 0
    for i in range(4):
        xin = qs.ik_inputs_from_batch(b, i)
        r = qs.ik_exact(ip, xin, "qpoases")
        H, g, A, lb, ub, lbA, ubA = qs.ik_assemble_qpoases(ip, xin)
        lo = np.where(np.abs(lb) > 1e300, -np.inf, lb); hi = np.where(np.abs(ub) > 1e300, np.inf, ub)
        x = _slsqp(H, g, A, lbA, np.zeros((0, 29)), np.zeros(0), lo, hi)
        assert np.abs(x[6:] - r["dq"]).max() < 1e-8, i
        n_active += len(r["lower"]) + len(r["upper"])
    assert n_active >= 6          # the comparison is on instances whose bounds bind
