"""
Parity on ALL THREE parameter sets the reference ships (tests/robots.py holds their values: iCubGazeboV2_5, iCubGenova04,
icubGazeboSim - the last one is the only robot that ships `use_mpc 1`, i.e. the one configuration in which the reference
runs BOTH hot-path QPs).  Same bars as tests/test_gpu_parity.py: |x_gpu - x_exact| <= 1e-9, active sets bit for bit where the
strict-complementarity margin exceeds 1e-7, oracle-infeasible <=> WCQP_STATUS_INFEASIBLE.
"""
import os
import subprocess

import numpy as np
import pytest

import robots

SOL_TOL = 1e-9
MARGIN = 1e-7
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "_build", "host_mirror_driver")


# ------------------------------------------------------------------------------------------------- CPU (no GPU needed)
@pytest.mark.parametrize("horizon", [50, 200])
@pytest.mark.parametrize("robot", robots.NAMES)
def test_condensing_on_every_shipped_parameter_set(wca, qs, robot, horizon):
    """The host-side condensing (rows of K^-1, K = [P A_eq'; A_eq 0] of dimension 6N + 4: 1204 at the shipped N = 200) against
    two independent solves of the equality-only QP: the oracle's KKT solve as it stands, and the SAME optimum from the KKT
    with the cost scaled by 1/R (the optimum does not depend on a cost scale; that system is 11 orders better conditioned).
    Measured here (and asserted loosely): cond(K) 5.5e14 / 6.9e14 (iCubGazeboV2_5, icubGazeboSim, N = 50 / 200), 2.6e15 /
    3.4e16 for iCubGenova04's Q = 750, R = 9e7 - and still |u0 - u0_kkt| <= 2e-13 at N = 200, because the conditioning is
    the cost's SCALE (R against the unit rows of A_eq), not a loss of rank: cond of the scaled system is 1e4 .. 5e5."""
    rng = np.random.default_rng(3)
    m = robots.mpc_solver(wca, robot, horizon)
    Gr, Gx, Gu, S0 = m.condensed()
    c = qs.mpc_constants(robots.mpc_params(qs, robot, horizon))
    worst = 0.0
    for _ in range(3):
        x0, up, ref = 0.05 * rng.normal(size=2), 0.05 * rng.normal(size=2), 0.05 * rng.normal(size=(horizon + 1, 2))
        P, q, A, l, u = qs.mpc_assemble(c, x0, ref, up, np.zeros((0, 2)), np.zeros(0))
        u0 = np.einsum("iab,ib->a", Gr, ref) + Gx @ x0 + Gu @ up
        z, _ = qs._kkt_solve(P, q, A, u)
        s = 1.0 / robots.ROBOTS[robot]["R"]
        Ks = np.block([[s * P, A.T], [A, np.zeros((c.n_x, c.n_x))]])
        zs = np.linalg.solve(Ks, np.concatenate([-s * q, u]))
        worst = max(worst, np.abs(u0 - z[c.n_x:c.n_x + 2]).max(), np.abs(u0 - zs[c.n_x:c.n_x + 2]).max())
    assert worst <= 1e-11, worst
    K = np.block([[P, A.T], [A, np.zeros((c.n_x, c.n_x))]])
    assert np.linalg.cond(Ks) < 1e7 < 1e13 < np.linalg.cond(K)
    Si = np.linalg.inv(Ks)[c.n_x:c.n_x + 2, c.n_x:c.n_x + 2] * s          # (K^-1)_uu = s (Ks^-1)_uu
    assert np.abs(S0 - Si).max() <= 1e-9 * np.abs(S0).max()


def _driver(tmp_path, robot, *args, controller_horizon=0.5):
    if not os.path.exists(DRIVER):
        pytest.skip("driver not built (run __graft_entry__.build())")
    m, k = tmp_path / "mpc.ini", tmp_path / "ik.ini"
    m.write_text(robots.mpc_ini(robot, controller_horizon)); k.write_text(robots.ik_ini(robot))
    argv = [DRIVER] + [a.replace("@mpc", str(m)).replace("@ik", str(k)) for a in args]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    recs, cur = [], {}
    for ln in r.stdout.splitlines():
        key, _, val = ln.partition(":")
        nums = np.array([float(x) for x in val.split()])
        if key == "tick" and cur:
            recs.append(cur); cur = {}
        cur[key] = nums
    recs.append(cur)
    return recs


@pytest.mark.parametrize("robot", robots.NAMES)
def test_every_shipped_config_parses_through_the_host_mirror(tmp_path, robot):
    """WalkingController::initialize / WalkingQPIK::initialize of the C++ mirror on each robot's controllerParams.ini /
    qpInverseKinematics.ini values (the shipped controllerHorizon 2 s -> N = 200 included)."""
    for ch in (0.5, 2.0):
        (r,) = _driver(tmp_path, robot, "parse", "@mpc", "@ik", controller_horizon=ch)
        assert r["mpc_init"][0] == 1 and r["ik_init"][0] == 1


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("horizon", [50, 200])
@pytest.mark.parametrize("robot", robots.NAMES)
def test_mpc_against_oracle_on_every_robot(wca, qs, robot, horizon):
    """test_mpc_against_oracle_live / test_mpc_shipped_horizon_200 on each robot's Q, R, CoM height."""
    c = qs.mpc_constants(robots.mpc_params(qs, robot, horizon))
    B = 96 if horizon == 50 else 24
    b = wca.synth.synth_mpc_batch(B, seed=2024 + horizon, horizon=horizon, uprev_sigma=0.06, x0_sigma=0.03)
    out = robots.mpc_solver(wca, robot, horizon).solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    nact = 0
    for i in range(B):
        r = qs.mpc_exact(c, b["x0"][i], b["ref"][i], b["u_prev"][i], b["hull_A"][i], b["hull_b"][i], int(b["hull_nc"][i]))
        assert out["status"][i] in (wca.STATUS_SOLVED, wca.STATUS_OUTSIDE_HULL)
        assert np.abs(out["u0"][i] - r["u0"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active"][i]) == sum(1 << e for e in r["active"])
        nact += len(r["active"])
    assert nact > B // 4          # the hull really binds


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", [0, 4], ids=["default", "nullspace_16l"])
@pytest.mark.parametrize("form,vmax", [("qpoases", 0.4), ("qpoases", 0.22), ("osqp", 0.3)])
@pytest.mark.parametrize("robot", robots.NAMES)
def test_ik_against_oracle_on_every_robot(wca, qs, robot, form, vmax, algorithm):
    """test_ik_against_oracle_live on each robot's weights, gains, posture and additional_rotation; the default kernel and its
    fall-back."""
    B = 96
    rb = robots.ROBOTS[robot]
    b = wca.synth.synth_ik_batch(B, seed=99, additional_rotation=rb["additional_rotation"], posture_deg=rb["reg_deg"])
    p = robots.ik_params(qs, robot, vmax)
    out = robots.ik_solver(wca, robot, form, vmax, algorithm=algorithm).solve_host(b["J_left"], b["J_right"], b["J_neck"], b["J_com"], b["q"], b["state"])
    checked = n_bounds = 0
    for i in range(B):
        try:
            r = qs.ik_exact(p, qs.ik_inputs_from_batch(b, i), form)
        except qs.QPInfeasible:
            assert out["status"][i] == wca.STATUS_INFEASIBLE
            continue
        assert out["status"][i] == wca.STATUS_SOLVED
        assert np.abs(out["dq"][i] - r["dq"]).max() <= SOL_TOL
        if r["mu_min_active"] > MARGIN and r["slack_min_inactive"] > MARGIN:
            assert int(out["active_lower"][i]) == sum(1 << j for j in r["lower"])
            assert int(out["active_upper"][i]) == sum(1 << j for j in r["upper"])
        checked += 1
        n_bounds += len(r["lower"]) + len(r["upper"])
    assert checked > B * 0.7
    assert (n_bounds > 0) == (form == "qpoases")


@pytest.mark.gpu
@pytest.mark.parametrize("robot", robots.NAMES)
def test_tick_pipeline_on_every_robot(wca, qs, robot):
    """The closed-loop tick (constant Jacobians; the whole run in one launch) with each robot's controller parameters - MPC
    weights, CoM height, ZMP-CoM gains (zmpControllerParams.ini:7-8), IK weights / gains / neck rotation - against
    oracle/tick_spec.py at 1e-9 over 150 ticks (more than one contact change per robot).  The velocity limit is chosen per robot
    so that bounds bind; with iCubGenova04's foot gains (k_posFoot 7, k_attFoot 5) the synthetic constant-Jacobian robots mostly
    reach an infeasible IK within the run (10 of 12 in the oracle): the device must then stop the SAME robots on the SAME ticks -
    this case doubles as the failure-path test at another parameter set."""
    from oracle import tick_spec as ts
    B, T = 12, 150
    vmax = {"iCubGazeboV2_5": 0.45, "iCubGenova04": 1.2, "icubGazeboSim": 0.3}[robot]
    rb = robots.ROBOTS[robot]
    p = ts.TickParams(com_height=rb["com_height"], k_com=rb["k_com"], k_zmp=rb["k_zmp"])
    d = wca.synth.synth_tick_batch(B, T, com_height=rb["com_height"], additional_rotation=rb["additional_rotation"])
    ref = ts.run_ticks(p, d, T, robots.ik_params(qs, robot, vmax), mpc_params=robots.mpc_params(qs, robot, 50))
    pipe = wca.TickPipeline(B, T, robots.mpc_solver(wca, robot, 50), robots.ik_solver(wca, robot, "qpoases", vmax), log_ticks=T,
                            k_com=rb["k_com"], k_zmp=rb["k_zmp"])
    pipe.upload(d)
    pipe.run(T)
    out = pipe.download()
    assert out["tick"] == T
    assert np.array_equal(out["mpc_fail"], ref["mpc_fail"]) and np.array_equal(out["ik_fail"], ref["ik_fail"])
    ok = ref["ik_fail"] == 0
    assert ok.sum() >= (2 if robot == "iCubGenova04" else B)
    assert ((ref["active_lower_log"] | ref["active_upper_log"]) != 0).any()          # bounds really bind
    assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9
    assert np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8                        # stopped robots: dq = 0 on both sides
    assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9
    assert np.abs(out["dcm"] - ref["dcm"]).max() <= 1e-9 and np.abs(out["com"] - ref["com"]).max() <= 1e-9
    assert np.array_equal(out["active_lower"][ok], ref["active_lower"][ok]) and np.array_equal(out["active_upper"][ok], ref["active_upper"][ok])


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iCubGenova04", "icubGazeboSim"])
def test_walking_controller_tick_sequence_on_the_other_robots(tmp_path, qs, robot):
    """tests/test_host_mirror.py::test_walking_controller_tick_sequence through wc::WalkingController with the other two robots'
    controllerParams.ini values (14 ticks: contact changes, a trajectory reset, a stale shift, a short deque)."""
    recs = _driver(tmp_path, robot, "mpc", "@mpc")
    assert len(recs) == 14
    c = qs.mpc_constants(robots.mpc_params(qs, robot, 50))
    u_prev = np.zeros(2)
    q_prev, feet_prev = None, None
    for r in recs:
        tick, lc, rc, reset, solved, got, status, active, did_reset = r["tick"].astype(int)
        fresh = (lc, rc) != feet_prev or did_reset == 1
        feet_prev = (lc, rc)
        q = qs.mpc_gradient(c, r["deque"].reshape(-1, 2), u_prev, q_prev=None if fresh else q_prev, reset=bool(reset))
        q_prev = q
        ref_window = -np.linalg.solve(c.Q, q[:c.n_x].reshape(-1, 2).T).T
        ex = qs.mpc_exact(c, r["x0"], ref_window, u_prev, r["hull_A"].reshape(-1, 2), r["hull_b"])
        assert solved == 1 and got == 1 and status == 0
        assert np.abs(r["u0"] - ex["u0"]).max() <= 1e-9
        assert int(active) == sum(1 << e for e in ex["active"])
        u_prev = r["u0"]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["qpoases", "osqp"])
@pytest.mark.parametrize("robot", ["iCubGenova04", "icubGazeboSim"])
def test_walking_qpik_tick_sequence_on_the_other_robots(tmp_path, qs, robot, form):
    """tests/test_host_mirror.py::test_walking_qpik_tick_sequence through wc::WalkingQPIK_{qpOASES,osqp} with the other two robots'
    qpInverseKinematics.ini values (setDesiredNeckOrientation applies THEIR additional_rotation)."""
    recs = _driver(tmp_path, robot, "ik", "@ik", form)
    assert len(recs) == 6
    p = robots.ik_params(qs, robot, 0.35)
    for r in recs:
        tick, solved, got, got_twice, status, lo, up = r["tick"].astype(int)
        if "q_reg" in r:
            p.joint_reg_deg = np.rad2deg(r["q_reg"])
        x = qs.IKInputs(
            J_left=r["J_left"].reshape(6, 29), J_right=r["J_right"].reshape(6, 29),
            J_neck=r["J_neck6"].reshape(6, 29)[3:], J_com=r["J_com"].reshape(3, 29), q=r["q"],
            p_left=r["p_left"], R_left=r["R_left"].reshape(3, 3), p_right=r["p_right"], R_right=r["R_right"].reshape(3, 3),
            pd_left=r["pd_left"], Rd_left=r["Rd_left"].reshape(3, 3), pd_right=r["pd_right"], Rd_right=r["Rd_right"].reshape(3, 3),
            R_neck=r["R_neck"].reshape(3, 3), Rd_neck=r["neck_des_arg"].reshape(3, 3) @ p.additional_rotation, com=r["com"],
            com_des=r["com_des"], com_vel_des=r["com_vel"], twist_left=r["twist_left"], twist_right=r["twist_right"])
        ex = qs.ik_exact(p, x, form)
        assert solved == 1 and got == 1 and status == 0
        assert np.abs(r["dq"] - ex["dq"]).max() <= 1e-9
        if ex["mu_min_active"] > 1e-7 and ex["slack_min_inactive"] > 1e-7:
            assert int(lo) == sum(1 << j for j in ex["lower"]) and int(up) == sum(1 << j for j in ex["upper"])
