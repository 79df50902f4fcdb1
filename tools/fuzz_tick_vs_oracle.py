#!/usr/bin/env python3
"""Closed-loop fuzz: the tick pipeline (fused kinematics, and constant Jacobians) against oracle/tick_spec.py on robots and horizons the
GPU suite does not have time for: several groups of robots (different `first` offsets = different synthetic robots), 300 ticks each.
Same tolerances as tests/test_tick_pipeline.py (u0 1e-9, dq 1e-8, q_des 1e-9; identical failure bookkeeping).  Prints one JSON line per case.
   python tools/fuzz_tick_vs_oracle.py [groups] [robots per group] [ticks] [first group]     (first group: robots no earlier run has seen; round 3 ran groups 1..4)"""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import walking_controllers_amd as wca
from oracle import qp_spec as qs, tick_spec as ts
sys.path.insert(0, os.path.join(R, "tests"))
import robots as rb
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
G0 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
S = wca.synth
p = ts.TickParams()
worst = {"u0": 0.0, "dq": 0.0, "q_des": 0.0}
t0 = time.time()
for g in range(G):
    first = 1000 * (g + G0)
    # ---- fused kinematics on the walk scenario
    kin = wca.KinModel(S.icub_like_model())
    kb = S.synth_walk_kin_batch(B, first=first)
    poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
    d = S.synth_walk_batch(B, T, poses, kb, first=first)
    vmax = np.broadcast_to(np.asarray(S.WALK_VMAX, float), (23,)).copy()
    ref = ts.run_ticks(p, d, T, qs.IKParams(v_max=vmax, joint_reg_deg=S.WALK_POSTURE_DEG.copy()), kin_model=S.icub_like_model(), foot_rect=S.FOOT_RECT)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(S.WALK_POSTURE_DEG))
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(), ik, first=first, log_ticks=T, kin=kin)
    pipe.upload(d); pipe.run(T)
    out = pipe.download()
    e = {"u0": float(np.abs(out["u0_log"] - ref["u0_log"]).max()), "dq": float(np.abs(out["dq_log"] - ref["dq_log"]).max()), "q_des": float(np.abs(out["q_des"] - ref["q_des"]).max())}
    assert np.array_equal(out["ik_fail"], ref["ik_fail"]) and out["mpc_fail"].sum() == ref["mpc_fail"].sum() == 0
    assert e["u0"] <= 1e-9 and e["dq"] <= 1e-8 and e["q_des"] <= 1e-9, ("kin", g, e)
    assert np.array_equal(out["active_lower"], ref["active_lower"]) and np.array_equal(out["active_upper"], ref["active_upper"])
    print(json.dumps(dict(case="tick, fused kinematics", first=first, robots=B, ticks=T, max_abs_err=e, robots_failed=int((ref["ik_fail"] > 0).sum()),
                          ticks_with_a_bound_at_its_limit=int((np.abs(np.abs(ref["dq_log"]) - vmax) < 1e-12).any(axis=(1, 2)).sum()))), flush=True)
    for k in worst: worst[k] = max(worst[k], e[k])
    # ---- constant Jacobians, with the controller parameters of the three robots the reference ships in turn (tests/robots.py: MPC weights,
    # CoM height, ZMP-CoM gains, IK weights / gains / neck rotation); iCubGenova04's foot gains stop most of these synthetic robots within
    # the run - in the oracle too: the failure path at scale
    robot = rb.NAMES[(g + G0) % len(rb.NAMES)]
    rr = rb.ROBOTS[robot]
    v2 = {"iCubGazeboV2_5": 0.45, "iCubGenova04": 1.2, "icubGazeboSim": 0.3}[robot]
    p2 = ts.TickParams(com_height=rr["com_height"], k_com=rr["k_com"], k_zmp=rr["k_zmp"])
    d2 = S.synth_tick_batch(B, T, first=first, com_height=rr["com_height"], additional_rotation=rr["additional_rotation"])
    ref2 = ts.run_ticks(p2, d2, T, rb.ik_params(qs, robot, v2), mpc_params=rb.mpc_params(qs, robot, 50))
    pipe2 = wca.TickPipeline(B, T, rb.mpc_solver(wca, robot, 50), rb.ik_solver(wca, robot, "qpoases", v2), first=first, log_ticks=T, k_com=rr["k_com"], k_zmp=rr["k_zmp"])
    pipe2.upload(d2); pipe2.run(T)
    o2 = pipe2.download()
    e = {"u0": float(np.abs(o2["u0_log"] - ref2["u0_log"]).max()), "dq": float(np.abs(o2["dq_log"] - ref2["dq_log"]).max()), "q_des": float(np.abs(o2["q_des"] - ref2["q_des"]).max())}
    assert np.array_equal(o2["ik_fail"], ref2["ik_fail"]), ("tables", g)
    assert e["u0"] <= 1e-9 and e["dq"] <= 1e-8 and e["q_des"] <= 1e-9, ("tables", g, e)
    print(json.dumps(dict(case="tick, constant Jacobians", robot=robot, first=first, robots=B, ticks=T, max_abs_err=e, robots_failed=int((ref2["ik_fail"] > 0).sum()))), flush=True)
    for k in worst: worst[k] = max(worst[k], e[k])
print(json.dumps(dict(summary=dict(groups=G, robots_per_group=B, ticks=T, robot_ticks=2 * G * B * T, worst=worst, seconds=round(time.time() - t0, 1)))), flush=True)
