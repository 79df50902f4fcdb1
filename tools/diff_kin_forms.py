#!/usr/bin/env python3
"""Differential run of the tick's three kinematics forms (fused / compact / dense) at the benchmark size: same robots, same ticks,
how far do the closed-loop trajectories drift apart from rounding alone?  Diagnostic (GPU box)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S = wca.synth
kin = wca.KinModel(S.icub_like_model())
kb = S.synth_walk_kin_batch(B)
poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
d = S.synth_walk_batch(B, T, poses, kb)
res = {}
for name, h in (("fused", 0), ("compact", 2), ("dense", 1)):
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=S.WALK_VMAX, joint_reg_rad=np.deg2rad(S.WALK_POSTURE_DEG))
    p = wca.TickPipeline(B, T, wca.MpcSolver(), ik, kin=kin, kin_handoff=h, log_ticks=0)
    p.upload(d); p.run(T)
    res[name] = p.download()
    p.close()
out = {"robots": B, "ticks": T}
for a, b in (("fused", "compact"), ("compact", "dense")):
    dq = np.abs(res[a]["q_des"] - res[b]["q_des"]).max(axis=1)
    out["%s_vs_%s" % (a, b)] = {"max_abs_q_des": float(dq.max()), "robots_above_1e-9": int((dq > 1e-9).sum()), "robots_above_1e-6": int((dq > 1e-6).sum()),
                                "dcm_identical": bool(np.array_equal(res[a]["dcm"], res[b]["dcm"])), "ik_fail": [int(res[a]["ik_fail"].sum()), int(res[b]["ik_fail"].sum())],
                                "hot_hit_equal": bool(np.array_equal(res[a]["hot_hit"], res[b]["hot_hit"]))}
print(json.dumps(out))
