#!/usr/bin/env python3
"""Why do robots of the walk scenario (per-tick kinematics) reach an infeasible IK?  Runs the tick pipeline in chunks,
records the first failing chunk of every robot and prints what those robots look like; then the same with one factor
of the scenario removed at a time.  Diagnostic only (GPU box)."""
import argparse
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca  # noqa: E402


def lift_swing(B, first, lift=(0.08, 0.14), lin_sigma=0.01, ang_sigma=0.02):
    """swing twist amplitudes of a foot that is LIFTED: vertical component up (the zero-net-displacement profile of
    tick_device.h brings it back down), small horizontal / angular scatter"""
    rng = wca.synth.CounterRNG(2718 ^ 0x11F7, first, B)
    tw = np.zeros((B, 6))
    tw[:, 0:2] = rng.normal(2, lin_sigma)
    tw[:, 2] = rng.uniform(1, lift[0], lift[1])[:, 0]
    tw[:, 3:6] = rng.normal(3, ang_sigma)
    return np.ascontiguousarray(tw)


def scenario(B, T, first=0, swing_scale=1.0, joint_sigma=0.015, base_rot_sigma=0.015, posture=None, neck_scale=1.0, dcm_sigma=0.002, lift=None):
    S = wca.synth
    posture = S.WALK_POSTURE_DEG if posture is None else posture
    kin = wca.KinModel(S.icub_like_model())
    kb = S.synth_kin_batch(B, seed=27182, first=first, joint_sigma=joint_sigma, posture_deg=posture, base_rot_sigma=base_rot_sigma)
    poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
    d = S.synth_walk_batch(B, T, poses, kb, first=first)
    d["swing_twist"] = np.ascontiguousarray(d["swing_twist"] * swing_scale) if lift is None else lift_swing(B, first, lift)
    return kin, d, posture


def run(tag, B, T, vmax=1.0, noise=1e-4, chunk=16, verbose=False, **kw):
    kin, d, posture = scenario(B, T, **kw)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(posture))
    pipe = wca.TickPipeline(B, T, wca.MpcSolver(horizon=50), ik, kin=kin, noise=noise)
    pipe.upload(d)
    first_fail = np.full(B, -1, np.int64)
    q_at_fail = {}
    done = 0
    q_prev = d["q0"].copy()
    while done < T:
        n = min(chunk, T - done)
        pipe.run(n, use_graph=False)
        done += n
        o = pipe.download()
        new = np.flatnonzero((o["ik_fail"] > 0) & (first_fail < 0))
        first_fail[new] = done - o["ik_fail"][new]           # tick index of the first failure (fail counts every tick since)
        for i in new[:64]:
            q_at_fail[int(i)] = o["q_des"][i].copy()
        q_prev = o["q_des"]
    nf = int((first_fail >= 0).sum())
    res = {"tag": tag, "robots": B, "ticks": T, "robots_with_ik_fail": nf, "mpc_fail": int(o["mpc_fail"].sum()),
           "frac_robot_ticks_with_previous_bounds": float(o["hot_try"].sum()) / (B * T)}
    if nf:
        ff = first_fail[first_fail >= 0]
        res["first_fail_tick_quantiles"] = [int(x) for x in np.quantile(ff, [0, 0.25, 0.5, 0.75, 1.0])]
        # where in the step cycle
        ph = (ff + d["phase0"][first_fail >= 0]) % 360
        sidx = ph % 180
        res["fail_in_double_support"] = int((sidx < 110).sum())
        res["fail_in_single_support"] = int((sidx >= 110).sum())
        res["fail_sidx_hist"] = np.histogram(sidx, bins=[0, 30, 60, 90, 110, 130, 150, 170, 180])[0].tolist()
        if verbose:
            for i, q in list(q_at_fail.items())[:6]:
                dq0 = np.rad2deg(q - d["q0"][i])
                res.setdefault("examples", []).append({"robot": i, "tick": int(first_fail[i]), "sidx": int((first_fail[i] + d["phase0"][i]) % 180),
                                                        "side": int(((first_fail[i] + d["phase0"][i]) % 360) // 180),
                                                        "q_deg": np.round(np.rad2deg(q), 1).tolist(), "moved_deg": np.round(dq0, 1).tolist()})
    # how far the joints travel (all robots): the extreme knee angles tell how close a leg gets to straight
    qd = np.rad2deg(o["q_des"])
    res["final_knee_deg_min_max"] = [float(qd[:, [14, 20]].min()), float(qd[:, [14, 20]].max())]
    print(json.dumps(res), flush=True)
    pipe.close()
    return res


def crouch(hip, knee):
    p = wca.synth.WALK_POSTURE_DEG.copy()
    p[[11, 17]] = -hip; p[[14, 20]] = knee; p[[15, 21]] = -(knee - hip)
    return p


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=1024)
    ap.add_argument("--big", type=int, default=0, help="also run the best candidate at this batch")
    args = ap.parse_args()
    B, T = args.batch, args.ticks
    for hip, knee, legs, up in ((50, 100, 1.5, 0.3), (48, 96, 1.5, 0.3), (50, 100, 1.5, 0.25)):
        vm = legs * np.ones(23); vm[:11] = up
        run("big_crouch_%d_%d_lift_0.04_0.08_vmax_legs%.2f_upper_%.2f" % (hip, knee, legs, up), args.big, T, lift=(0.04, 0.08), posture=crouch(hip, knee), vmax=vm, chunk=128)
