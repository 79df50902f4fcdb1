#!/usr/bin/env python3
"""How much of the walking tick is the active-set walk?  Times the fused-kinematics tick (8192 robots, 1000 ticks, one launch) for
several upper-body velocity limits (legs at 1.5 rad/s): the looser the limit, the fewer robot-ticks carry active bounds.  Diagnostic (GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca
B, T = 8192, 1024
S = wca.synth
kin = wca.KinModel(S.icub_like_model())
kb = S.synth_walk_kin_batch(B)
poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
d = S.synth_walk_batch(B, T, poses, kb)
for up in (0.3, 0.4, 0.6, 1.5):
    vm = S.WALK_VMAX.copy(); vm[:11] = up
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vm, joint_reg_rad=np.deg2rad(S.WALK_POSTURE_DEG))
    p = wca.TickPipeline(B, T, wca.MpcSolver(), ik, kin=kin)
    p.upload(d); p.run(24); wca.capi.stream_synchronize()
    t0 = time.perf_counter(); p.run(1000); wca.capi.stream_synchronize(); dt = time.perf_counter() - t0
    o = p.download()
    nact = np.array([bin(int(a) | int(b)).count("1") for a, b in zip(o["active_lower"], o["active_upper"])])
    print(json.dumps({"upper_body_vmax": up, "active_bounds_per_robot_hist_at_last_tick": np.bincount(nact, minlength=9)[:12].tolist(), "us_per_tick": 1e6 * dt / 1000, "qp_per_s": 2 * B * 1000 / dt, "robots_failed": int((o["ik_fail"] > 0).sum()),
                      "hot_start_tried_frac": float(o["hot_try"].sum()) / (B * T), "hot_start_hit_rate": float(o["hot_hit"].sum()) / max(1, int(o["hot_try"].sum()))}), flush=True)
    p.close()
