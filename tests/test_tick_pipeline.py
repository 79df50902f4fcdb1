"""
Device-resident tick pipeline (BASELINE configs 4/5; SURVEY §8f-1 tick harness, §8f-2 glue)
against its CPU restatement oracle/tick_spec.py, which drives the exact solvers through the
reference's per-tick call order (WalkingModule.cpp:578-745).
"""
import numpy as np
import pytest


def test_contact_schedule_and_reference_are_consistent(wca):
    """CPU: the synthetic DCM reference satisfies xi_{t+1} = a xi_t + b zmp_t with the ZMP
    reference inside the support polygon of the scheduled contact pair."""
    from oracle import tick_spec as ts
    p = ts.TickParams()
    d = wca.synth.synth_tick_batch(6, 400)
    a = np.exp(np.sqrt(p.gravity / p.com_height) * p.dT)
    xi, z = d["ref_traj"], d["zmp_ref"]
    assert np.abs(xi[:, 1:] - (a * xi[:, :-1] + (1 - a) * z[:, :-1])).max() < 1e-12
    for t in range(0, 400, 7):
        code = ts.contact_code(t, d["phase0"], p)
        for i in range(6):
            k = int(code[i]); nc = int(d["hull_tab_nc"][i, k])
            assert (d["hull_tab_A"][i, k, :nc] @ z[i, t] <= d["hull_tab_b"][i, k, :nc] + 1e-12).all()
    full = wca.synth.synth_tick_batch(6, 50)
    part = wca.synth.synth_tick_batch(3, 50, first=3)
    assert np.array_equal(part["ref_traj"], full["ref_traj"][3:]) and np.array_equal(part["phase0"], full["phase0"][3:])


@pytest.mark.gpu
@pytest.mark.parametrize("ik_algorithm", [0, 4, 3], ids=["fused_1_launch", "2_launches", "4_launches"])
def test_tick_pipeline_matches_cpu_restatement(wca, qs, ik_algorithm):
    """Default IK algorithm (base elimination): MPC, glue, IK and post step run in ONE launch per tick; the general
    16-lane kernel keeps the MPC launch (2 per tick); an explicit 32-lane algorithm keeps the stand-alone glue /
    post kernels as well (4 per tick)."""
    from oracle import tick_spec as ts
    B, T = 24, 150            # > one contact change per instance (double support lasts 110 ticks)
    p = ts.TickParams()
    d = wca.synth.synth_tick_batch(B, T)
    vmax = 0.45
    ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax * np.ones(23)))
    assert ref["mpc_fail"].sum() == 0 and ref["ik_fail"].sum() == 0
    for use_graph in (False, True):
        # plain launches / hipGraph replay of one-tick launches; the fused kernel's default (the whole run in ONE launch, a wave
        # walking through the ticks on its own) is the third form below
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=ik_algorithm), log_ticks=T,
                                ticks_per_launch=1)
        pipe.upload(d)
        pipe.run(T, use_graph=use_graph)
        out = pipe.download()
        assert out["tick"] == T and out["mpc_fail"].sum() == 0 and out["ik_fail"].sum() == 0
        # closed loop over 150 ticks: rounding differences are damped (|lambda| = 0.974), not amplified
        assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9
        assert np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8
        assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9
        assert np.abs(out["dcm"] - ref["dcm"]).max() <= 1e-9 and np.abs(out["com"] - ref["com"]).max() <= 1e-9
        # the active joint-velocity bounds the last tick ended on (what the next tick's hot start begins from): bit-exact
        assert np.array_equal(out["active_lower"], ref["active_lower"]) and np.array_equal(out["active_upper"], ref["active_upper"])
        if not use_graph:
            eager = out
    assert np.array_equal(out["u0_log"], eager["u0_log"]) and np.array_equal(out["dq_log"], eager["dq_log"])   # graph == eager, bitwise
    assert np.abs(ref["dq_log"]).max() == pytest.approx(vmax, abs=1e-12)        # velocity limits really bind
    # ... and stopped at a tick where they do, the pipeline hands back exactly the oracle's active sets
    tb = int(np.nonzero((ref["active_lower_log"] | ref["active_upper_log"]).any(axis=1))[0][-1]) + 1
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=ik_algorithm))
    pipe.upload(d)
    pipe.run(tb)
    part = pipe.download()
    assert np.array_equal(part["active_lower"], ref["active_lower_log"][tb - 1]) and np.array_equal(part["active_upper"], ref["active_upper_log"][tb - 1])
    assert (part["active_lower"] | part["active_upper"]).any()
    # several ticks per launch (0: all of a run() call; 7: launches of 7, 7, ... and a remainder), split over two run() calls
    for k in (0, 7):
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=ik_algorithm), log_ticks=T,
                                ticks_per_launch=k)
        pipe.upload(d)
        pipe.run(61); pipe.run(T - 61)
        multi = pipe.download()
        assert multi["tick"] == T
        for key in ("u0_log", "dq_log", "q_des", "dcm", "com", "ik_fail", "mpc_fail", "hot_try", "hot_hit", "active_lower", "active_upper"):
            assert np.array_equal(multi[key], eager[key]), (k, key)


@pytest.mark.gpu
@pytest.mark.parametrize("kin_mode", [False, True], ids=["constant_jacobians", "fused_kinematics"])
def test_tick_pipeline_in_the_osqp_form(wca, qs, kin_mode):
    """The tick with the reference's osqp back-end semantics (WalkingQPIK_osqp: joint-limit rows that never bind, the extra
    k_attFoot on the neck term, the zero-twist rule for the foot in contact - SURVEY Appendix B-13/14/15): the stance foot's
    twist IS exactly zero in the tick, so the zero-twist branch is the one that runs.  Against oracle/tick_spec.py at 1e-9."""
    from oracle import tick_spec as ts
    B, T = 8, 90
    p = ts.TickParams()
    if kin_mode:
        kin, d = _walk_scenario(wca, B, T)
        extra = dict(joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
        ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=np.ones(23), joint_reg_deg=wca.synth.WALK_POSTURE_DEG.copy()), ik_form="osqp",
                           kin_model=wca.synth.icub_like_model(), foot_rect=wca.synth.FOOT_RECT)
    else:
        kin, d, extra = None, wca.synth.synth_tick_batch(B, T), {}
        ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=np.ones(23)), ik_form="osqp")
    assert ref["ik_fail"].sum() == 0 and ref["mpc_fail"].sum() == 0
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_OSQP, v_max=1.0, **extra), log_ticks=T, kin=kin)
    pipe.upload(d); pipe.run(T)
    out = pipe.download()
    assert out["tick"] == T and out["ik_fail"].sum() == 0 and out["mpc_fail"].sum() == 0
    assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9 and np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8
    assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9
    assert np.abs(ref["dq_log"]).max() > 0.05


@pytest.mark.gpu
def test_tick_pipeline_long_run_is_stable_and_shard_invariant(wca):
    """1000 ticks (config 5 length) on 512 robots: no solver failure, DCM stays on its reference,
    and two half-batches reproduce the full batch bit for bit."""
    B, T = 512, 1000
    d = wca.synth.synth_tick_batch(B, T)

    def run(data, first, count):
        pipe = wca.TickPipeline(count, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5), first=first)
        pipe.upload(data)
        pipe.run(T, use_graph=True)
        return pipe.download()
    full = run(d, 0, B)
    assert full["tick"] == T and full["mpc_fail"].sum() == 0
    assert (full["ik_fail"] > 0).sum() <= 0.002 * B + 1      # robots that stopped (a failed IK stops a robot for good)
    # an IK the device did not solve must be one the exact oracle cannot solve either:
    # replay every failing robot (at most three) through oracle/tick_spec.py and compare the failure counts
    from oracle import tick_spec as ts, qp_spec as qs
    for i in np.flatnonzero(full["ik_fail"])[:3]:
        one = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        one["first"] = int(i)
        ref = ts.run_ticks(ts.TickParams(), one, T, qs.IKParams(v_max=0.5 * np.ones(23)))
        assert ref["ik_fail"][0] == full["ik_fail"][i]
    assert np.abs(full["dcm"] - d["ref_traj"][:, T]).max() < 0.05
    halves = [run(wca.synth.synth_tick_batch(B // 2, T, first=f), f, B // 2) for f in (0, B // 2)]
    assert np.array_equal(np.concatenate([h["q_des"] for h in halves]), full["q_des"])
    assert np.array_equal(np.concatenate([h["dcm"] for h in halves]), full["dcm"])


@pytest.mark.gpu
@pytest.mark.parametrize("ticks_per_launch", [0, 1, 3], ids=["whole_call_per_launch", "one_tick_per_launch", "three_per_launch"])
def test_tick_run_refuses_to_run_past_the_trajectories(wca, ticks_per_launch):
    """The per-instance trajectories hold max_ticks + N + 1 stages: enqueueing more ticks than that (in one call or
    over several) must be refused, not read the neighbour's trajectory; odd tick counts exercise the
    phase parity of the two-copy tick index with and without graph replays."""
    B, T = 8, 30
    d = wca.synth.synth_tick_batch(B, T)
    ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5)
    ref = wca.TickPipeline(B, T, wca.MpcSolver(), ik(), log_ticks=T)
    ref.upload(d); ref.run(T, use_graph=False)
    want = ref.download()
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), ik(), log_ticks=T, ticks_per_launch=ticks_per_launch)
    pipe.upload(d)
    with pytest.raises(wca.WcqpError):
        pipe.run(T + 1, use_graph=False)
    pipe.run(1, use_graph=True)          # a call of ONE tick: the MPC of its tick alone, then the tick without an MPC ahead
    pipe.run(2, use_graph=True)          # plain (fewer than a graph's 8 ticks), leaves an odd tick index
    pipe.run(19, use_graph=True)         # one plain tick to an even index, two graph replays, two plain ticks
    pipe.run(8, use_graph=False)
    with pytest.raises(wca.WcqpError):
        pipe.run(1, use_graph=True)
    got = pipe.download()
    assert got["tick"] == T
    assert np.array_equal(got["u0_log"], want["u0_log"]) and np.array_equal(got["dq_log"], want["dq_log"])
    pipe.upload(d)                       # upload rewinds
    pipe.run(T, use_graph=True)
    assert np.array_equal(pipe.download()["dq_log"], want["dq_log"])


@pytest.mark.gpu
@pytest.mark.parametrize("ticks_per_launch", [0, 1], ids=["whole_call_per_launch", "one_tick_per_launch+graph"])
def test_trajectory_merge_on_the_device(wca, qs, ticks_per_launch):
    """SURVEY 8f-1, trajectory merge (WM/src/WalkingModule.cpp:500-535, 1263-1308): a newly planned DCM trajectory is spliced in
    20 ticks ahead of the running tick while the pipeline keeps its state - twice in 330 ticks (a merge every 150 ticks), between
    wcqp_tick_run calls.  Same trajectories as oracle/tick_spec.py with the same merges (1e-9), and really different from the
    run without them; captured graphs stay valid (the trajectory buffer does not move)."""
    from oracle import tick_spec as ts
    B, T, vmax = 8, 330, 0.45
    p = ts.TickParams()
    d = wca.synth.synth_tick_batch(B, T)
    N = p.horizon
    merges = {}
    for t_m in (130, 280):
        frm = t_m + 20
        n = T + N + 1 - frm
        # the new plan: the old one bent away smoothly by up to 1.5 cm (per-instance direction), continuous at the merge point
        ramp = 1.0 - np.exp(-np.arange(n) / 40.0)
        dirn = np.stack([np.cos(0.7 * np.arange(B) + t_m), np.sin(0.7 * np.arange(B) + t_m)], 1)
        base = merges[130][1][:, 150:] if t_m == 280 else d["ref_traj"][:, frm:]
        merges[t_m] = (frm, np.ascontiguousarray(base[:, :n] + 0.015 * ramp[None, :, None] * dirn[:, None, :]))
    ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax)
    ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax * np.ones(23)), splices=merges)
    plain = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax * np.ones(23)))
    assert np.abs(ref["u0_log"] - plain["u0_log"]).max() > 1e-3          # the merges matter
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), ik(), log_ticks=T, ticks_per_launch=ticks_per_launch)
    pipe.upload(d)
    done = 0
    for t_m in (130, 280):
        pipe.run(t_m - done, use_graph=True); done = t_m
        pipe.splice_reference(*merges[t_m])
    pipe.run(T - done, use_graph=True)
    out = pipe.download()
    assert out["tick"] == T and out["mpc_fail"].sum() == 0 and np.array_equal(out["ik_fail"], ref["ik_fail"])
    assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9 and np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8
    assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9 and np.abs(out["dcm"] - ref["dcm"]).max() <= 1e-9
    # a merge may not reach into the past, nor beyond the trajectories
    with pytest.raises(wca.WcqpError):
        pipe.splice_reference(T - 1, np.zeros((B, 4, 2)))
    with pytest.raises(wca.WcqpError):
        pipe.splice_reference(T + N, np.zeros((B, 4, 2)))


@pytest.mark.gpu
@pytest.mark.parametrize("kin_mode", [False, True], ids=["constant_jacobians", "fused_kinematics"])
def test_logger_rows_match_the_cpu_restatement(wca, qs, kin_mode):
    """SURVEY 8f-4, second half: the row WalkingModule hands its logger per tick (WM/src/WalkingModule.cpp:800-810, the 53 columns
    of :1231-1250) - measured / desired DCM and ZMP, CoM, desired CoM position / velocity, actual and desired foot poses as
    position + roll-pitch-yaw, foot errors - kept per robot for the first `logger_ticks` ticks by a logging build of the tick
    kernel; against oracle/tick_spec.py's rows.  The trajectories are the ones of the product kernel, bit for bit."""
    from oracle import tick_spec as ts
    B, T, L = 6, 40, 40
    p = ts.TickParams()
    if kin_mode:
        kin, d = _walk_scenario(wca, B, T)
        vmax = wca.synth.WALK_VMAX
        mk = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
        ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax.copy(), joint_reg_deg=wca.synth.WALK_POSTURE_DEG.copy()),
                           kin_model=wca.synth.icub_like_model(), foot_rect=wca.synth.FOOT_RECT, logger_ticks=L)
    else:
        kin, d = None, wca.synth.synth_tick_batch(B, T)
        mk = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.45)
        ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=0.45 * np.ones(23)), logger_ticks=L)
    outs = []
    for lt in (L, 0):
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), mk(), log_ticks=T, kin=kin, logger_ticks=lt)
        pipe.upload(d); pipe.run(17); pipe.run(T - 17)
        outs.append(pipe.download())
    logged, plain = outs
    assert np.array_equal(logged["dq_log"], plain["dq_log"]) and np.array_equal(logged["u0_log"], plain["u0_log"])
    assert "logger" not in plain and logged["logger"].shape == (L, B, 53)
    assert ref["ik_fail"].sum() == 0 and logged["ik_fail"].sum() == 0
    err = np.abs(logged["logger"] - ref["logger"])
    assert err[:, :, :41].max() <= 1e-9, np.unravel_index(err[:, :, :41].argmax(), err[:, :, :41].shape)
    assert err[:, :, 41:].max() <= 1e-9             # foot errors: residuals of equality rows, O(1e-15) on both sides
    assert np.abs(ref["logger"][:, :, 20:23]).max() > 1e-3 and np.abs(ref["logger"][:, :, 8:10] - ref["logger"][:, :, 6:8]).max() > 1e-6


def _walk_scenario(wca, B, T, first=0):
    """A coherent synthetic robot marching in place (synth_walk_batch): poses from the device kinematics at tick 0."""
    kin = wca.KinModel(wca.synth.icub_like_model())
    kb = wca.synth.synth_walk_kin_batch(B, first=first)
    poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
    return kin, wca.synth.synth_walk_batch(B, T, poses, kb, first=first)


def _kin_tick_against_oracle(wca, qs, B, T, vmax, ik_algorithm, handoff=0, graphs=(False, True), ticks_per_launch=0):
    from oracle import tick_spec as ts
    p = ts.TickParams()
    kin, d = _walk_scenario(wca, B, T)
    model = wca.synth.icub_like_model()
    vmax = np.broadcast_to(np.asarray(vmax, float), (23,)).copy()
    ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax, joint_reg_deg=wca.synth.WALK_POSTURE_DEG.copy()),
                       kin_model=model, foot_rect=wca.synth.FOOT_RECT)
    assert ref["mpc_fail"].sum() == 0
    outs = []
    for use_graph in graphs:
        ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, algorithm=ik_algorithm, joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), ik, log_ticks=T, kin=kin, kin_handoff=handoff, ticks_per_launch=ticks_per_launch)
        pipe.upload(d)
        pipe.run(T, use_graph=use_graph)
        out = pipe.download()
        assert out["tick"] == T and out["mpc_fail"].sum() == 0
        assert np.array_equal(out["ik_fail"], ref["ik_fail"])
        assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9
        assert np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8
        assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9
        assert np.abs(out["dcm"] - ref["dcm"]).max() <= 1e-9 and np.abs(out["com"] - ref["com"]).max() <= 1e-9
        assert np.array_equal(out["active_lower"], ref["active_lower"]) and np.array_equal(out["active_upper"], ref["active_upper"])
        outs.append(out)
    return d, ref, outs, vmax


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["fused_kinematics", "fused_kinematics,one_tick_per_launch", "kin_launch+compact_handoff", "kin_launch+dense_handoff", "kin+mpc+ik"])
def test_tick_pipeline_with_per_tick_kinematics_matches_cpu_restatement(wca, qs, variant):
    """SURVEY 8f-4 inside the tick: every tick evaluates the forward kinematics at the integrated joint state with the base
    anchored at the stance foot and hands fresh MIXED Jacobians and the actual poses to the IK
    (WM/src/WalkingModule.cpp:715, 396-410) - inside the solve kernel itself (default; many ticks per launch or one), through a
    kinematics launch per tick that hands over compact per-joint records or four dense Jacobians (`kin_handoff`), or through the
    general 16-lane kernel with a stand-alone MPC launch (algorithm 4);
    the support polygons come from the desired foot poses (...PredictiveController.cpp:364-435).  150 closed-loop ticks against
    oracle/tick_spec.py (kin_spec + hull_spec + the exact QP solvers) at 1e-9, graph replay == plain launches bitwise, no robot fails."""
    alg = 4 if variant == "kin+mpc+ik" else 0
    handoff = wca.KIN_HANDOFF_DENSE if "dense" in variant else (wca.KIN_HANDOFF_COMPACT if "compact" in variant else wca.KIN_HANDOFF_FUSED)
    d, ref, (eager, out), vmax = _kin_tick_against_oracle(wca, qs, 12, 150, wca.synth.WALK_VMAX, alg, handoff=handoff,
                                                          ticks_per_launch=1 if "one_tick" in variant else 0)
    assert ref["ik_fail"].sum() == 0                       # the walk scenario does not fall
    assert np.abs(ref["q_des"] - d["q0"]).max() > 0.05     # the Jacobians really change: the joints travel
    assert np.array_equal(out["u0_log"], eager["u0_log"]) and np.array_equal(out["dq_log"], eager["dq_log"])   # graph == eager, bitwise
    assert (np.abs(np.abs(ref["dq_log"]) - vmax) < 1e-12).any()        # velocity limits really bind


@pytest.mark.gpu
def test_kinematics_handoff_forms_agree(wca):
    """The compact kinematics -> IK hand-off carries exactly the non-zero entries of the four Jacobians, and the fused kinematics
    phase computes them where they are used: all three forms of the tick give the same trajectories (the kinematics kernel's
    two forms bit for bit in u0; the fused phase, 16 lanes per robot instead of 32, to rounding)."""
    B, T = 256, 64
    kin, d = _walk_scenario(wca, B, T)
    res = []
    for handoff in (wca.KIN_HANDOFF_COMPACT, wca.KIN_HANDOFF_DENSE, wca.KIN_HANDOFF_FUSED):
        ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=wca.synth.WALK_VMAX, joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), ik, log_ticks=T, kin=kin, kin_handoff=handoff)
        pipe.upload(d); pipe.run(T, use_graph=True)
        res.append(pipe.download())
    a, b, c = res
    assert a["ik_fail"].sum() == 0 and b["ik_fail"].sum() == 0 and c["ik_fail"].sum() == 0
    assert np.array_equal(a["u0_log"], b["u0_log"]) and np.array_equal(a["u0_log"], c["u0_log"])
    assert np.abs(a["dq_log"] - b["dq_log"]).max() <= 1e-12 and np.abs(a["q_des"] - b["q_des"]).max() <= 1e-12
    assert np.abs(a["dq_log"] - c["dq_log"]).max() <= 1e-10 and np.abs(a["q_des"] - c["q_des"]).max() <= 1e-11


@pytest.mark.gpu
def test_a_robot_whose_ik_fails_is_stopped_like_the_oracle(wca, qs):
    """The failure path (WM/src/WalkingModule.cpp:723-739: updateModule returns false on an unsolved QP-IK): with joint
    velocity limits too tight for the pelvis' change of sides (0.45 rad/s on every joint) some robots' IK becomes infeasible
    in double support; the device stops exactly those robots on exactly those ticks (dq = 0 from then on), as
    oracle/tick_spec.py does."""
    d, ref, outs, vmax = _kin_tick_against_oracle(wca, qs, 10, 130, 0.45, 0, graphs=(True,))
    assert (ref["ik_fail"] > 0).sum() >= 1 and (ref["ik_fail"] == 0).sum() >= 1
    stopped = np.flatnonzero(ref["ik_fail"] > 0)
    first = 130 - ref["ik_fail"][stopped]
    for i, t0 in zip(stopped, first):
        assert np.all(outs[0]["dq_log"][t0:, i] == 0.0)


@pytest.mark.gpu
def test_ik_hot_start_matches_cold_start_and_falls_back(wca):
    """IK hot start (SQProblem::hotstart, qp.cpp:312-335): every tick first tries the previous tick's
    active velocity bounds.  Same trajectories as the all-cold pipeline (the QP is strictly convex: 1e-9, identical failure
    counts); the previous set is accepted on most ticks on which it is tried, and REJECTED on some - a contact change
    rewrites the foot twists, a bound stops binding - where the kernel must fall back to the cold walk."""
    B, T, vmax = 256, 240, 0.45
    d = wca.synth.synth_tick_batch(B, T)
    outs = {}
    for hot in (True, False):
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax), log_ticks=T, ik_hot_start=hot)
        pipe.upload(d)
        pipe.run(T, use_graph=True)
        outs[hot] = pipe.download()
    hot, cold = outs[True], outs[False]
    assert cold["hot_try"].sum() == 0 and cold["hot_hit"].sum() == 0
    assert np.array_equal(hot["ik_fail"], cold["ik_fail"]) and hot["mpc_fail"].sum() == 0
    assert np.abs(hot["dq_log"] - cold["dq_log"]).max() <= 1e-9 and np.abs(hot["q_des"] - cold["q_des"]).max() <= 1e-9
    tries, hits = int(hot["hot_try"].sum()), int(hot["hot_hit"].sum())
    assert tries > 0.005 * B * T                      # bounds bind on a share of the robot-ticks here (1.4 % at v_max 0.45)
    assert hits > 0.8 * tries                         # ... and mostly stay the same from one tick to the next
    assert hits < tries                               # the fall-back really ran


@pytest.mark.gpu
@pytest.mark.parametrize("kin_mode", [True, False], ids=["kinematics", "constant_jacobians"])
def test_tick_at_the_benchmark_geometry(wca, qs, kin_mode):
    """BASELINE configs 4/5 at their per-GPU size as bench.py runs them: 8192 robots, cut into three robot
    groups, each a pipeline on a HIP stream of its own, 64 ticks.  No oracle finishes this size in seconds, so the checks are
    the size-independent properties: no QP fails, every joint velocity respects its limit and the ones the solver reports
    active sit exactly on it, the three groups reproduce one full-batch pipeline bit for bit (shard == full), and a sample
    of robots matches oracle/tick_spec.py over the first ticks."""
    B, T, L = 8192, 64, 8
    S = wca.synth
    cuts = [B * k // 3 for k in range(4)]
    if kin_mode:
        vmax = S.WALK_VMAX
        mk_ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(S.WALK_POSTURE_DEG))
        kin = wca.KinModel(S.icub_like_model())

        def data(first, cnt):
            kb = S.synth_walk_kin_batch(cnt, first=first)
            poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((cnt, 87)))["state"]
            return S.synth_walk_batch(cnt, T, poses, kb, first=first)
    else:
        vmax = 0.5 * np.ones(23)
        mk_ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5)
        kin = None
        data = lambda first, cnt: S.synth_tick_batch(cnt, T, first=first)

    def run(first, cnt, stream=0):
        d = data(first, cnt)
        pipe = wca.TickPipeline(cnt, T, wca.MpcSolver(), mk_ik(), first=first, log_ticks=L, kin=kin)
        pipe.upload(d)
        return d, pipe
    groups = [run(cuts[k], cuts[k + 1] - cuts[k]) for k in range(3)]
    streams = [wca.capi.stream_create() for _ in range(3)]
    wca.capi.stream_synchronize()
    for (d, pipe), st in zip(groups, streams):
        pipe.run(T, use_graph=True, stream=st)          # the three groups are in flight together
    outs = [pipe.download() for _, pipe in groups]
    for st in streams:
        wca.capi.stream_destroy(st)
    cat = lambda k: np.concatenate([o[k] for o in outs], axis=(1 if k.endswith("_log") else 0))
    assert all(o["tick"] == T for o in outs) and cat("mpc_fail").sum() == 0
    fails = cat("ik_fail")
    assert (fails > 0).sum() <= (0 if kin_mode else 2)         # the walk does not fall; a random constant-Jacobian robot may be infeasible
    dq = cat("dq_log")
    assert (np.abs(dq) <= vmax + 1e-12).all()
    on = np.isclose(np.abs(dq), vmax, rtol=0, atol=1e-12)
    assert on.any() and (np.abs(dq)[on] == np.broadcast_to(vmax, dq.shape)[on]).all()          # active bounds are tight, exactly
    # shard == full, bit for bit
    dfull, pfull = run(0, B)
    pfull.run(T, use_graph=True)
    full = pfull.download()
    for k in ("u0_log", "dq_log", "q_des", "dcm", "com", "ik_fail"):
        assert np.array_equal(cat(k), full[k]), k
    # a sample of robots (and every robot that failed) through the CPU restatement, first L ticks
    from oracle import tick_spec as ts
    sample = sorted(set([0, 1, cuts[1] - 1, cuts[1], cuts[2], B - 1]) | set(np.flatnonzero(fails)[:2].tolist()))
    for i in sample:
        one = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in dfull.items()}
        one["first"] = int(i)
        ipar = qs.IKParams(v_max=np.asarray(vmax, float).copy(), **({"joint_reg_deg": S.WALK_POSTURE_DEG.copy()} if kin_mode else {}))
        ref = ts.run_ticks(ts.TickParams(), one, L, ipar, **({"kin_model": S.icub_like_model(), "foot_rect": S.FOOT_RECT} if kin_mode else {}))
        assert np.abs(full["u0_log"][:, i] - ref["u0_log"][:, 0]).max() <= 1e-9, i
        assert np.abs(full["dq_log"][:, i] - ref["dq_log"][:, 0]).max() <= 1e-8, i


@pytest.mark.gpu
@pytest.mark.parametrize("kin_mode", [False, True], ids=["constant_jacobians", "fused_kinematics"])
def test_external_feedback_mode(wca, qs, kin_mode):
    """wcqp_tick_params.plant = EXTERNAL: every tick reads the caller's measured DCM / CoM / ZMP (and joints) from
    device arrays (wcqp_tick_set_feedback_device) instead of stepping the internal LIPM plant - what the reference reads from the robot every
    tick (WM/src/WalkingModule.cpp:612, 665, 373).  (a) Fed the internal plant's OWN states, tick by tick, the pipeline reproduces the
    internal run (1e-9 against oracle/tick_spec.py, whose logs are what is fed back); (b) fed other measurements - a disturbed DCM, a ZMP
    that is not the previous command, measured joints that differ from the desired ones - it follows tick_spec.run_ticks(external=...);
    (c) the call contract: one tick per run call, each behind its own feedback."""
    from oracle import tick_spec as ts
    B, T = 10, 60
    p = ts.TickParams()
    if kin_mode:
        kin, d = _walk_scenario(wca, B, T)
        vmax = wca.synth.WALK_VMAX.copy()
        ipar = qs.IKParams(v_max=vmax, joint_reg_deg=wca.synth.WALK_POSTURE_DEG.copy())
        okw = dict(kin_model=wca.synth.icub_like_model(), foot_rect=wca.synth.FOOT_RECT)
        mk_ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
    else:
        kin, d = None, wca.synth.synth_tick_batch(B, T)
        vmax = 0.45 * np.ones(23)
        ipar, okw = qs.IKParams(v_max=vmax), {}
        mk_ik = lambda: wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.45)
    internal = ts.run_ticks(p, d, T, ipar, **okw)

    def run_external(ext):
        pipe = wca.TickPipeline(B, T, wca.MpcSolver(), mk_ik(), log_ticks=T, kin=kin, external_feedback=True)
        pipe.upload(d)
        with pytest.raises(wca.WcqpError):
            pipe.run(1)                                   # no feedback yet
        for t in range(T):
            # (host arrays: wcqp_tick_set_feedback_host stages them and enqueues the same copy kernel wcqp_tick_set_feedback_device does -
            # the device entry point itself is driven by bench.py's `tick.external_feedback` pass, from torch tensors)
            pipe.set_feedback_host(ext["dcm"][t], ext["com"][t], ext["zmp"][t], ext["q"][t] if ext.get("q") is not None else None)
            if t == 3:
                with pytest.raises(wca.WcqpError):
                    pipe.run(2)                           # one tick per call
            pipe.run(1)
        with pytest.raises(wca.WcqpError):
            pipe.run(1)                                   # the feedback of tick T - 1 was consumed
        return pipe.download()
    # (a) the internal plant's own states fed back
    same = run_external(dict(dcm=internal["dcm_log"], com=internal["com_log"], zmp=internal["zmp_log"]))
    assert same["tick"] == T and np.array_equal(same["ik_fail"], internal["ik_fail"]) and same["mpc_fail"].sum() == 0
    assert np.abs(same["u0_log"] - internal["u0_log"]).max() <= 1e-9 and np.abs(same["dq_log"] - internal["dq_log"]).max() <= 1e-8
    assert np.abs(same["q_des"] - internal["q_des"]).max() <= 1e-9
    # ... which is also what the pipeline with the INTERNAL plant does, whatever the launch form
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), mk_ik(), log_ticks=T, kin=kin)
    pipe.upload(d); pipe.run(T)
    own = pipe.download()
    assert np.abs(same["u0_log"] - own["u0_log"]).max() <= 1e-9 and np.abs(same["dq_log"] - own["dq_log"]).max() <= 1e-8
    # (b) measurements of somebody else's plant
    rng = np.random.default_rng(4)
    ext = dict(dcm=internal["dcm_log"] + 1e-3 * rng.normal(size=(T, B, 2)), com=internal["com_log"] + 5e-4 * rng.normal(size=(T, B, 2)),
               zmp=internal["zmp_log"] + 2e-3 * rng.normal(size=(T, B, 2)), q=internal["q_log"] + 0.01 * rng.normal(size=(T, B, 23)))
    ref = ts.run_ticks(p, d, T, ipar, external=ext, **okw)
    out = run_external(ext)
    assert np.array_equal(out["ik_fail"], ref["ik_fail"]) and np.array_equal(out["mpc_fail"], ref["mpc_fail"])
    assert np.abs(out["u0_log"] - ref["u0_log"]).max() <= 1e-9 and np.abs(out["dq_log"] - ref["dq_log"]).max() <= 1e-8
    assert np.abs(out["q_des"] - ref["q_des"]).max() <= 1e-9
    assert np.abs(ref["u0_log"] - internal["u0_log"]).max() > 1e-4 and np.abs(ref["dq_log"] - internal["dq_log"]).max() > 1e-3      # it really is another run
    # the internal-plant handle refuses feedback
    with pytest.raises(wca.WcqpError):
        pipe.set_feedback_device(1, 1, 1)
