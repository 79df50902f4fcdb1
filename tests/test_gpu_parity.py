"""
GPU parity tests: the HIP path, called through the C ABI (libwcqp.so), against
  * the committed golden vectors (exact fp64 optimum from oracle/qp_spec.py), and
  * the oracle run here on the same seeded inputs at sizes it finishes in seconds, and
  * size-independent KKT properties at BASELINE.json's full batch sizes.

Tolerances (fp64 path; north_star: "solutions within 1e-6 of reference, active-set
indices bit-exact"):
  |x_gpu - x_exact|_inf <= 1e-9 (three orders tighter than required);
  active sets compared bit-for-bit on instances whose strict-complementarity margin
  (smallest active multiplier / smallest inactive slack) exceeds 1e-7; ties are counted
  and must stay rare.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SOL_TOL = 1e-9
MARGIN = 1e-7
# IK kernels: 0 = default dispatch (base elimination + range space, csrc/ik4.hip, general kernel behind it),
# 5 = the same asked for explicitly, 4 = general 16-lane null-space kernel (csrc/ik3.hip: the fall-back),
# 3 = 32-lane null-space kernel with the MFMA Gram tile (csrc/ik2.hip: kept as the independent cross-check)
ALGS = [0, 5, 4, 3]
ALG_IDS = ["default", "base_elim", "nullspace_16l", "nullspace_mfma"]


def _load(golden_dir, name):
    import os
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _clear_cut(g):
    return (g["mu_min_active"] > MARGIN) & (g["slack_min_inactive"] > MARGIN)


# ------------------------------------------------------------------------------- MPC ---
@pytest.mark.parametrize("name,kw", [("mpc_cfg2_b4096.npz", {}),
                                     ("mpc_stress_b1024.npz", {"uprev_sigma": 0.04})])
def test_mpc_matches_golden(wca, golden_dir, name, kw):
    g = _load(golden_dir, name)
    B, seed = int(g["count"]), int(g["seed"])
    b = wca.synth.synth_mpc_batch(B, seed=seed, **kw)
    assert np.array_equal(b["ref"][:4], g["in_ref"]) and np.array_equal(b["hull_b"][:4], g["in_hull_b"])
    out = wca.MpcSolver().solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    assert (out["status"] == wca.STATUS_SOLVED).all()
    assert np.abs(out["u0"] - g["u0"]).max() <= SOL_TOL
    cc = _clear_cut(g)
    assert cc.mean() > 0.98
    assert np.array_equal(out["active"][cc], g["active"][cc])          # bit-exact active sets
    assert np.abs(out["margin"] - g["margin"]).max() <= 1e-9


def test_mpc_against_oracle_live(wca, qs):
    """Fresh seeds, wider disturbances, every contact configuration."""
    c = qs.mpc_constants(qs.MPCParams())
    B = 192
    b = wca.synth.synth_mpc_batch(B, seed=2024, uprev_sigma=0.06, x0_sigma=0.03)
    out = wca.MpcSolver().solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    nact = 0
    for i in range(B):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active"][i]) == sum(1 << e for e in r["active"])
        nact += len(r["active"])
    assert nact > B // 2          # the hull really binds in this variant


def test_mpc_short_reference_is_padded(wca, qs):
    """ref_len < N+1: the tail repeats the last sample (MPCSolver.cpp:200-214)."""
    c = qs.mpc_constants(qs.MPCParams())
    b = wca.synth.synth_mpc_batch(16, seed=5, uprev_sigma=0.03)
    short = np.ascontiguousarray(b["ref"][:, :7, :])
    out = wca.MpcSolver().solve_host(b["x0"], short, b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    for i in range(16):
        r = qs.mpc_exact(c, b["x0"][i], short[i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL


def test_mpc_edge_cases(wca, qs):
    c = qs.mpc_constants(qs.MPCParams())
    m = wca.MpcSolver()
    b = wca.synth.synth_mpc_batch(5, seed=9, uprev_sigma=0.08)       # odd batch: ragged last wave
    # (a) no hull rows at all -> unconstrained optimum
    nc0 = np.zeros(5, np.int32)
    out = m.solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], nc0)
    for i in range(5):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], np.zeros((0, 2)), np.zeros(0))
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL
        assert out["active"][i] == 0 and np.isinf(out["margin"][i])
    # (b) empty polygon -> INFEASIBLE, never a silent answer
    A = b["hull_A"].copy(); bb = b["hull_b"].copy()
    A[:, 0] = [1.0, 0.0]; bb[:, 0] = -1.0
    A[:, 1] = [-1.0, 0.0]; bb[:, 1] = -1.0           # x <= -1 and x >= 1
    out = m.solve_host(b["x0"], b["ref"], b["u_prev"], A, bb, b["hull_nc"])
    assert (out["status"] == wca.STATUS_INFEASIBLE).all()
    # (c) batch of one == per-robot call
    one = m.solve_host(b["x0"][:1], b["ref"][:1], b["u_prev"][:1], b["hull_A"][:1], b["hull_b"][:1], b["hull_nc"][:1])
    full = m.solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    assert np.array_equal(one["u0"][0], full["u0"][0])


def test_mpc_shipped_horizon_200(wca, qs):
    """The shipped configuration: controllerHorizon 2 s at 10 ms -> N = 200, n = 802
    (CFG/controllerParams.ini:1); the kernel loops over 64-stage passes."""
    N = 200
    c = qs.mpc_constants(qs.MPCParams(horizon=N))
    b = wca.synth.synth_mpc_batch(20, seed=41, horizon=N, uprev_sigma=0.04)
    out = wca.MpcSolver(horizon=N).solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    for i in range(20):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert out["status"][i] == 0 and np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active"][i]) == sum(1 << e for e in r["active"])


def test_mpc_coupled_weights(wca, qs):
    """Non-diagonal (but symmetric) Q and R couple the two axes: Sigma0 is a full 2x2 metric."""
    Q = np.array([[7000.0, 1500.0], [1500.0, 5000.0]]); R = np.array([[9e6, -2e6], [-2e6, 6e6]])
    c = qs.mpc_constants(qs.MPCParams(Q=Q, R=R))
    b = wca.synth.synth_mpc_batch(48, seed=43, uprev_sigma=0.05)
    out = wca.MpcSolver(Q=Q, R=R).solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    for i in range(48):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active"][i]) == sum(1 << e for e in r["active"])


def test_mpc_properties_at_full_size(wca):
    """BASELINE config 4 per-GPU size (8192): feasibility, idempotence, permutation invariance."""
    B = 8192
    b = wca.synth.synth_mpc_batch(B, seed=31, uprev_sigma=0.05)
    m = wca.MpcSolver()
    out = m.solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    assert (out["status"] == wca.STATUS_SOLVED).all()
    res = np.einsum("bkc,bc->bk", b["hull_A"], out["u0"]) - b["hull_b"]
    assert res.max() <= 1e-10                               # inside the polygon
    act = out["active"]
    for e in range(8):                                      # active rows are tight
        on = (act >> e) & 1 == 1
        assert np.abs(res[on, e]).max(initial=0.0) <= 1e-12
    perm = np.random.default_rng(0).permutation(B)
    out2 = m.solve_host(b["x0"][perm], b["ref"][perm], b["u_prev"][perm], b["hull_A"][perm], b["hull_b"][perm], b["hull_nc"][perm])
    assert np.array_equal(out2["u0"], out["u0"][perm])      # batch-permutation invariance, bitwise
    assert np.array_equal(out2["active"], act[perm])


def test_hull_rows_from_foot_poses(wca, qs):
    """§8f-3: device hull builder against the oracle's builder (oracle/hull_spec.py) and the
    C++ host mirror's convention; then straight into the MPC kernel."""
    from oracle import hull_spec as hs
    rng = np.random.default_rng(7)
    B = 300
    rect = np.array([0.05, 0.025, 0.05, -0.025, -0.02, -0.025, -0.02, 0.025])     # foot_size corners
    def pose(xy, yaw):
        T = np.zeros((B, 12)); T[:, 0:2] = xy; c, s = np.cos(yaw), np.sin(yaw)
        T[:, 3] = c; T[:, 4] = -s; T[:, 6] = s; T[:, 7] = c; T[:, 11] = 1.0
        return T
    lxy = rng.normal(0, 0.02, (B, 2)) + [0.0, 0.08]; rxy = rng.normal(0, 0.03, (B, 2)) + [0.03, -0.08]
    lyaw, ryaw = rng.uniform(-0.4, 0.4, B), rng.uniform(-0.4, 0.4, B)
    contact = rng.integers(0, 4, B).astype(np.uint8)
    A, b, nc = wca.hull_from_feet_host(rect, pose(lxy, lyaw), pose(rxy, ryaw), contact)
    for i in range(B):
        pts = []
        if contact[i] & 1: pts.append(hs.foot_corners(lxy[i], lyaw[i]))
        if contact[i] & 2: pts.append(hs.foot_corners(rxy[i], ryaw[i]))
        if not pts:
            assert nc[i] == 0
            continue
        Ar, br, ncr = hs.hull_rows(np.vstack(pts))
        assert nc[i] == ncr
        assert np.abs(A[i] - Ar).max() < 1e-12 and np.abs(b[i, :ncr] - br[:ncr]).max() < 1e-12 and (b[i, ncr:] == 1e30).all()
    # rows feed the MPC kernel unchanged
    m = wca.synth.synth_mpc_batch(B, seed=3, uprev_sigma=0.05)
    out = wca.MpcSolver().solve_host(m["x0"], m["ref"], m["u_prev"], A, b, nc)
    c = qs.mpc_constants(qs.MPCParams())
    for i in range(0, B, 7):
        r = qs.mpc_exact(c, m["x0"][i], m["ref"][i], m["u_prev"][i], A[i], b[i], int(nc[i]))
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL


# -------------------------------------------------------------------------------- IK ---
def _ik_solver(wca, form, vmax, algorithm=0):
    return wca.IkSolver(form=wca.IK_FORM_QPOASES if form == "qpoases" else wca.IK_FORM_OSQP, v_max=vmax,
                        algorithm=algorithm)


@pytest.mark.parametrize("algorithm", ALGS, ids=ALG_IDS)
@pytest.mark.parametrize("name", ["ik_qpoases_v050_b1024.npz", "ik_qpoases_v050_b4096.npz", "ik_qpoases_v030_b512.npz", "ik_osqp_b512.npz"])
def test_ik_matches_golden(wca, golden_dir, name, algorithm):
    g = _load(golden_dir, name)
    B, seed, form, vmax = int(g["count"]), int(g["seed"]), str(g["form"]), float(g["v_max"])
    b = wca.synth.synth_ik_batch(B, seed=seed)
    assert np.array_equal(b["J_left"][:2], g["in_J_left"]) and np.array_equal(b["state"][:2], g["in_state"])
    out = _ik_solver(wca, form, vmax, algorithm).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    ok = g["status"] == 0
    assert (out["status"][ok] == wca.STATUS_SOLVED).all()
    assert (out["status"][~ok] != wca.STATUS_SOLVED).all()
    assert np.abs(out["dq"][ok] - g["dq"][ok]).max() <= SOL_TOL
    assert np.abs(out["foot_err"][ok] - g["foot_err"][ok]).max() <= 1e-8
    cc = _clear_cut(g) & ok
    assert cc.mean() > 0.97
    assert np.array_equal(out["active_lower"][cc], g["active_lower"][cc])   # bit-exact active sets
    assert np.array_equal(out["active_upper"][cc], g["active_upper"][cc])


@pytest.mark.parametrize("algorithm", ALGS, ids=ALG_IDS)
@pytest.mark.parametrize("form,vmax", [("qpoases", 0.4), ("qpoases", 0.22), ("osqp", 0.3)])
def test_ik_against_oracle_live(wca, qs, form, vmax, algorithm):
    B = 160
    b = wca.synth.synth_ik_batch(B, seed=99)
    p = qs.IKParams(v_max=vmax * np.ones(23))
    out = _ik_solver(wca, form, vmax, algorithm).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    checked = 0
    n_infeasible = 0
    for i in range(B):
        x = qs.ik_inputs_from_batch(b, i)
        try:
            r = qs.ik_exact(p, x, form)
        except qs.QPInfeasible:
            # the dual active set says INFEASIBLE exactly when the oracle's phase-1 LP does: no NUMERIC / MAX_ITER here
            assert out["status"][i] == wca.STATUS_INFEASIBLE
            n_infeasible += 1
            continue
        assert out["status"][i] == wca.STATUS_SOLVED
        assert np.abs(out["dq"][i] - r["dq"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active_lower"][i]) == sum(1 << j for j in r["lower"])
            assert int(out["active_upper"][i]) == sum(1 << j for j in r["upper"])
        checked += 1
    assert checked > B * 0.8
    if form == "qpoases" and vmax < 0.3:
        assert n_infeasible >= 1                 # the tight variant really contains infeasible instances


def test_ik_osqp_form_quirks(wca, qs):
    """osqp back-end: limits never bind, extra k_attFoot on the neck term, zero-twist rule
    (SURVEY Appendix B-13/14/15): the two forms must DIFFER on the same inputs."""
    b = wca.synth.synth_ik_batch(32, seed=3)
    a = _ik_solver(wca, "osqp", 0.2).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    q = _ik_solver(wca, "qpoases", 0.2).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    assert (a["active_lower"] == 0).all() and (a["active_upper"] == 0).all()
    assert np.abs(a["dq"]).max() > 0.2 + 1e-3            # limits are NOT enforced in this form
    ok = q["status"] == wca.STATUS_SOLVED
    assert np.abs(q["dq"][ok]).max() <= 0.2 + 1e-12
    assert np.abs(a["dq"] - q["dq"]).max() > 1e-3


@pytest.mark.parametrize("algorithm", ALGS, ids=ALG_IDS)
def test_ik_com_as_cost_variant(wca, qs, algorithm):
    """useCoMAsConstraint = 0: 12 equality rows, CoM task moves into the cost."""
    B = 48
    b = wca.synth.synth_ik_batch(B, seed=17)
    p = qs.IKParams(use_com_as_constraint=False, v_max=0.4 * np.ones(23))
    s = wca.IkSolver(form=wca.IK_FORM_QPOASES, use_com_as_constraint=False, v_max=0.4, algorithm=algorithm)
    out = s.solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    for i in range(B):
        r = qs.ik_exact(p, qs.ik_inputs_from_batch(b, i), "qpoases")
        assert out["status"][i] == wca.STATUS_SOLVED
        assert np.abs(out["dq"][i] - r["dq"]).max() <= SOL_TOL


def test_ik_stance_foot_touches_only_the_base(wca, qs):
    """iCub-shaped structure: the reference's floating base IS the stance-foot
    link, so that foot's Jacobian has zero joint columns.  The column-pivoted elimination must
    pick base columns for those rows."""
    B = 64
    b = wca.synth.synth_ik_batch(B, seed=23)
    JL = b["J_left"].copy(); JL[:, :, 6:] = 0.0
    p = qs.IKParams(v_max=5.0 * np.ones(23))
    for alg in (0, 4, 3):
        out = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=5.0, algorithm=alg).solve_host(JL, b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
        for i in range(B):
            x = qs.ik_inputs_from_batch(dict(b, J_left=JL), i)
            r = qs.ik_exact(p, x, "qpoases")
            assert out["status"][i] == 0 and np.abs(out["dq"][i] - r["dq"]).max() <= SOL_TOL


def _to_body_fixed(b, rows, rng):
    """Re-expresses the base velocity of the listed instances in a rotated base frame: nu_base = blkdiag(R, R) nu',
    i.e. every Jacobian's base block is right-multiplied by blkdiag(R, R).  Same joint-space problem, but the base
    blocks are no longer [I B; 0 I]."""
    out = {k: v.copy() for k, v in b.items()}
    for i in rows:
        w = rng.normal(0, 0.7, 3)
        th = np.linalg.norm(w); k = w / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        X = np.zeros((6, 6)); X[:3, :3] = R; X[3:, 3:] = R
        for name in ("J_left", "J_right", "J_neck", "J_com"):
            out[name][i][:, :6] = out[name][i][:, :6] @ X
    return out


def test_ik_jacobian_structure_fallback(wca, qs):
    """The default kernel relies on MIXED-representation base blocks and checks them per instance.  Instances without
    the pattern must (a) be re-solved by the general kernel under WCQP_IK_JAC_AUTO, with the oracle's optimum,
    (b) come back WCQP_STATUS_STRUCTURE under WCQP_IK_JAC_MIXED while their neighbours are solved,
    (c) agree with the all-general path (WCQP_IK_JAC_GENERAL) everywhere."""
    B, vmax = 203, 0.4
    rng = np.random.default_rng(5)
    b = wca.synth.synth_ik_batch(B, seed=61)
    odd = sorted(rng.choice(B, 37, replace=False).tolist() + [B - 1])
    bb = _to_body_fixed(b, odd, rng)
    args = (bb["J_left"], bb["J_right"], bb["J_neck"], bb["J_com"], bb["q"], bb["state"])
    p = qs.IKParams(v_max=vmax * np.ones(23))
    auto = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_AUTO).solve_host(*args)
    gen = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_GENERAL).solve_host(*args)
    mix = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_MIXED).solve_host(*args)
    is_odd = np.zeros(B, bool); is_odd[odd] = True
    assert (mix["status"][is_odd] == wca.STATUS_STRUCTURE).all() and (mix["dq"][is_odd] == 0).all()
    assert (mix["status"][~is_odd] != wca.STATUS_STRUCTURE).all()
    assert np.array_equal(mix["dq"][~is_odd], auto["dq"][~is_odd])
    assert (auto["status"] == gen["status"]).all() and (auto["status"] != wca.STATUS_STRUCTURE).all()
    assert np.array_equal(auto["dq"][is_odd], gen["dq"][is_odd])              # same kernel solved them
    ok = gen["status"] == 0
    assert np.abs(auto["dq"][ok] - gen["dq"][ok]).max() <= 1e-10
    assert (auto["active_lower"][ok] == gen["active_lower"][ok]).all() and (auto["active_upper"][ok] == gen["active_upper"][ok]).all()
    for i in odd[:12]:
        r = qs.ik_exact(p, qs.ik_inputs_from_batch(bb, i), "qpoases")
        assert auto["status"][i] == 0 and np.abs(auto["dq"][i] - r["dq"]).max() <= SOL_TOL
    # a rotated base frame changes the base velocity's coordinates, not the joint velocities
    plain = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    both = ok & (plain["status"] == 0)
    assert np.abs(plain["dq"][both] - auto["dq"][both]).max() <= 1e-9


def test_ik_mixed_pattern_tolerance(wca, qs):
    """A real producer forms the MIXED base blocks through rotation products (iDynTree, WM/src/WalkingForwardKinematics.cpp:33,
    436-454): R R' is I only to rounding.  Identity / zero entries perturbed by +-1 ulp (and up to 3e-14) must stay on the
    base-elimination kernel - SOLVED under WCQP_IK_JAC_MIXED, within 1e-9 of the oracle's optimum of the PERTURBED inputs, same
    active sets - while a 1e-10 perturbation is past WCQP_IK_MIXED_TOL: STRUCTURE under MIXED, the general kernel under AUTO."""
    B, vmax = 96, 0.4
    rng = np.random.default_rng(17)
    b = wca.synth.synth_ik_batch(B, seed=77)
    p = qs.IKParams(v_max=vmax * np.ones(23))

    def perturbed(eps_of):
        bb = {k: np.array(v, copy=True) for k, v in b.items()}
        for i in range(B):
            for name, rows in (("J_left", 6), ("J_right", 6), ("J_com", 3), ("J_neck", 3)):
                blk = bb[name][i][:, :6]
                for r in range(rows):
                    for c in range(6):
                        # the pattern entries: identity diagonals and the zero blocks (the B blocks, columns 3..5 of linear rows, are data)
                        lin_row = name != "J_neck" and r < 3
                        if lin_row and c >= 3:
                            continue
                        blk[r, c] += eps_of(blk[r, c], rng)
        return bb

    ulp = lambda v, g: g.choice([-1.0, 0.0, 1.0]) * (np.spacing(1.0) if v != 0.0 else 1e-17)
    tiny = lambda v, g: g.uniform(-3e-14, 3e-14)          # (the tolerance is on the SUM of a base column's deviations: 18 entries)
    for eps_of in (ulp, tiny):
        bb = perturbed(eps_of)
        assert any(not np.array_equal(bb[k], b[k]) for k in ("J_left", "J_right", "J_com", "J_neck"))
        args = (bb["J_left"], bb["J_right"], bb["J_neck"], bb["J_com"], bb["q"], bb["state"])
        mix = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_MIXED).solve_host(*args)
        assert (mix["status"] != wca.STATUS_STRUCTURE).all()
        n_checked = 0
        for i in range(0, B, 3):
            try:
                r = qs.ik_exact(p, qs.ik_inputs_from_batch(bb, i), "qpoases")
            except qs.QPInfeasible:
                assert mix["status"][i] == wca.STATUS_INFEASIBLE
                continue
            assert mix["status"][i] == 0 and np.abs(mix["dq"][i] - r["dq"]).max() <= SOL_TOL
            if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
                assert int(mix["active_lower"][i]) == sum(1 << j for j in r["lower"]) and int(mix["active_upper"][i]) == sum(1 << j for j in r["upper"])
            n_checked += 1
        assert n_checked >= 28
    far = perturbed(lambda v, g: 1e-10)
    args = (far["J_left"], far["J_right"], far["J_neck"], far["J_com"], far["q"], far["state"])
    mix = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_MIXED).solve_host(*args)
    assert (mix["status"] == wca.STATUS_STRUCTURE).all()
    auto = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_AUTO).solve_host(*args)
    gen = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, jacobian_structure=wca.IK_JAC_GENERAL).solve_host(*args)
    assert (auto["status"] != wca.STATUS_STRUCTURE).all() and np.array_equal(auto["dq"], gen["dq"])      # the general kernel solved them
    for i in range(0, B, 8):
        if auto["status"][i] == 0:
            r = qs.ik_exact(p, qs.ik_inputs_from_batch(far, i), "qpoases")
            assert np.abs(auto["dq"][i] - r["dq"]).max() <= SOL_TOL


def test_ik_properties_at_full_size(wca):
    """BASELINE config 3 size (4096) + ragged batch: KKT-type properties computed here in
    numpy from the GPU output alone, plus permutation invariance."""
    B = 4097
    vmax = 0.5
    b = wca.synth.synth_ik_batch(B, seed=8)
    s = _ik_solver(wca, "qpoases", vmax)
    out = s.solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    ok = out["status"] == wca.STATUS_SOLVED
    assert ok.mean() > 0.999
    # every instance the kernel did not solve must be infeasible for the exact oracle too
    from oracle import qp_spec as qs
    p_or = qs.IKParams(v_max=vmax * np.ones(23))
    for i in np.flatnonzero(~ok):
        assert out["status"][i] == wca.STATUS_INFEASIBLE
        with pytest.raises(qs.QPInfeasible):
            qs.ik_exact(p_or, qs.ik_inputs_from_batch(b, int(i)), "qpoases")
    dq = out["dq"]
    assert np.abs(dq[ok]).max() <= vmax + 1e-12                          # bounds hold
    lo, up = out["active_lower"], out["active_upper"]
    for j in range(23):                                                  # active bounds are tight
        assert np.abs(dq[ok & ((up >> j) & 1 == 1), j] - vmax).max(initial=0.0) <= 1e-12
        assert np.abs(dq[ok & ((lo >> j) & 1 == 1), j] + vmax).max(initial=0.0) <= 1e-12
    assert (lo & up == 0).all()
    # equality rows: J nu = b  <=>  reported foot errors vanish (they are b - J nu)
    assert np.abs(out["foot_err"][ok]).max() <= 1e-9
    perm = np.random.default_rng(1).permutation(B)
    out2 = s.solve_host(b["J_left"][perm], b["J_right"][perm], b["J_neck"][perm], b["J_com"][perm],
                        b["q"][perm], b["state"][perm])
    assert np.array_equal(out2["dq"], dq[perm])                          # bitwise, whatever the lane half
    assert np.array_equal(out2["active_upper"], up[perm])


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 3, 5, 7, 4095])
def test_ik_16lane_kernel_on_ragged_batches(wca, batch):
    """The 16-lane kernel packs 4 instances per wave: batches that leave 1..3 of a wave's rows
    without an instance must give, row for row, what the 32-lane kernel gives (same optimum,
    same active sets, same foot errors), and must not touch memory past the batch."""
    b = wca.synth.synth_ik_batch(batch, seed=77)
    args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    ref = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4, algorithm=3).solve_host(*args)
    for alg in (0, 4):
        out = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4, algorithm=alg).solve_host(*args)
        assert (out["status"] == ref["status"]).all()
        ok = ref["status"] == 0
        assert np.abs(out["dq"][ok] - ref["dq"][ok]).max(initial=0.0) <= 1e-10
        assert (out["active_lower"][ok] == ref["active_lower"][ok]).all() and (out["active_upper"][ok] == ref["active_upper"][ok]).all()
        assert np.abs(out["foot_err"][ok] - ref["foot_err"][ok]).max(initial=0.0) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("vmax", [0.2, 0.12])
def test_ik_kernels_agree_under_tight_bounds(wca, vmax):
    """Many active bounds (6..10 per instance, working-set drops, infeasible instances): the three
    kernels walk the same dual active set, so status, active sets and solutions must coincide
    (`tools/stress_ik.py` runs the same comparison on 300 k instances)."""
    B = 4000
    b = wca.synth.synth_ik_batch(B, seed=303)
    args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    outs = {a: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=a).solve_host(*args) for a in (0, 4, 3)}
    ref = outs[3]
    assert 0.3 * B < (ref["status"] == 0).sum() < B          # a real mix of solved and infeasible instances
    for a in (0, 4):
        o = outs[a]
        assert (o["status"] == ref["status"]).all()
        ok = ref["status"] == 0
        assert (o["active_lower"][ok] == ref["active_lower"][ok]).all() and (o["active_upper"][ok] == ref["active_upper"][ok]).all()
        assert np.abs(o["dq"][ok] - ref["dq"][ok]).max() <= 1e-9


# ------------------------------------------------------------------ status paths ---
def test_mpc_outside_hull_status(wca, qs):
    """WalkingController::solve fails when computeMargin(u0) < -convex_hull_tolerance (cpp:513-517).  With a NEGATIVE
    tolerance every solution closer than |tol| to the boundary - in particular every one ON it - must come back
    WCQP_STATUS_OUTSIDE_HULL with the (rejected) optimum still in u0 (Appendix B-8); the others stay SOLVED."""
    b = wca.synth.synth_mpc_batch(400, seed=77, uprev_sigma=0.06, x0_sigma=0.03)
    tol = -0.004
    strict = wca.MpcSolver(convex_hull_tolerance=tol).solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    plain = wca.MpcSolver().solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    assert (plain["status"] == wca.STATUS_SOLVED).all()
    want_out = plain["margin"] < -tol
    assert want_out.sum() > 20 and (~want_out).sum() > 20
    assert (strict["status"][want_out] == wca.STATUS_OUTSIDE_HULL).all()
    assert (strict["status"][~want_out] == wca.STATUS_SOLVED).all()
    assert np.array_equal(strict["u0"], plain["u0"]) and (plain["active"][plain["margin"] < 1e-12] != 0).all()


@pytest.mark.parametrize("algorithm", [0, 4, 3], ids=["default", "nullspace_16l", "nullspace_mfma"])
def test_ik_max_iter_status(wca, algorithm):
    """nWSR analogue (qp.cpp:312): with a budget of one working-set change, an instance that needs one bound is still
    solved, one that needs more comes back WCQP_STATUS_MAX_ITER - never SOLVED with a violated bound."""
    B, vmax = 600, 0.3
    b = wca.synth.synth_ik_batch(B, seed=515)
    args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    full = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=algorithm).solve_host(*args)
    one = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=algorithm, max_iter=1).solve_host(*args)
    needs_more = (full["iters"] > 1) & (full["status"] == wca.STATUS_SOLVED)
    needs_le1 = (full["iters"] <= 1) & (full["status"] == wca.STATUS_SOLVED)
    assert needs_more.sum() > 20 and needs_le1.sum() > 20
    assert (one["status"][needs_more] == wca.STATUS_MAX_ITER).all()
    assert (one["status"][needs_le1] == wca.STATUS_SOLVED).all()
    assert np.array_equal(one["dq"][needs_le1], full["dq"][needs_le1])


def test_ik_posture_update_reaches_the_solve(wca, qs):
    """wcqp_ik_set_posture = WalkingQPIK::setDesiredJointPosition: the new posture enters the gradient of the next solve
    (osqp.cpp:185, qp.cpp:166)."""
    B = 40
    b = wca.synth.synth_ik_batch(B, seed=9)
    args = (b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    s = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=2.0)
    before = s.solve_host(*args)
    reg = np.deg2rad(wca.synth.WALK_POSTURE_DEG)
    s.set_posture(reg)
    after = s.solve_host(*args)
    assert np.abs(after["dq"] - before["dq"]).max() > 1e-2
    p = qs.IKParams(v_max=2.0 * np.ones(23), joint_reg_deg=wca.synth.WALK_POSTURE_DEG.copy())
    for i in range(0, B, 5):
        r = qs.ik_exact(p, qs.ik_inputs_from_batch(b, i), "qpoases")
        assert after["status"][i] == 0 and np.abs(after["dq"][i] - r["dq"]).max() <= SOL_TOL


def test_qp_enqueue_steps_equals_the_single_calls():
    """wcqp_qp_enqueue_steps makes the same calls as wcqp_mpc_solve_device + wcqp_ik_solve_device per record
    (tests/helpers/enqueue_steps_check.py: three records, two streams, one record without an MPC part, bit for bit).
    In a process of its own: torch brings its own HIP runtime and has to initialise before libwcqp's does."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "enqueue_steps_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "enqueue_steps ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_qp_plan_equals_the_single_calls():
    """wcqp_qp_plan_*: a plan's ONE launch (wavefronts walking through the records, the MPC on the IK's lanes, several
    wavefronts per robot group) gives what the single calls give for every record, bit for bit - ragged batches, the BASELINE
    batch, more ways than records, replay (tests/helpers/plan_check.py, a process of its own)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "plan_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "plan ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_timed_launch_shape_against_goldens():
    """The launch shape bench.py times: BASELINE batch 4096, the bench's own inputs (MPC seed 1234 = golden
    mpc_cfg2_b4096, IK seed 4321 at v_max 0.5 = golden ik_qpoases_v050_b1024), 12 records handed over in ONE
    wcqp_qp_enqueue_steps call, spread over three streams that share ONE wcqp_mpc_t / wcqp_ik_t pair (three batches in
    flight, each record the one-launch step qp_pair_kernel), inputs rotated per record like the bench's input sets; every
    record's outputs against the goldens (tests/helpers/timed_shape_check.py; a process of its own because torch brings its
    own HIP runtime and has to initialise before libwcqp's does)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "timed_shape_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "timed shape ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
