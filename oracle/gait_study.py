"""
ORACLE-SIDE STUDY (test infrastructure; never imported by the product path).

Which synthetic gait can the reference's DCM-MPC track in closed loop?  (VERDICT r1 item 7 / DESIGN.md tick section.)

The reference's planner (UnicyclePlanner's DCMTrajectoryGenerator, upstream; driven from
WM/src/TrajectoryGenerator.cpp:93-154 with app/robots/iCubGazeboV2_5/plannerParams.ini:34-40) builds the DCM
reference step by step: in SINGLE support the ZMP sits on the stance foot's ZMP point (foot + left/rightZMPDelta) and
the DCM follows xi' = omega (xi - zmp) backwards from the end of the step; in DOUBLE support the DCM is a cubic
polynomial that joins the boundary DCM positions and velocities of the two neighbouring single supports.  Shipped
timing: nominalDuration 0.9 s per step, switchOverSwingRatio 0.7 => 0.53 s single + 0.37 s double support.

This script closes the loop  MPC (the OSQP restatement in oracle/wc_oracle.c at library defaults, i.e. what the
reference runs; shipped weights Q = 7500, R = 9e6, hull of the current contact pair on u0) -> LIPM plant
xi+ = a xi + b u0  for a robot marching in place, for
  * the two ways of shaping the double support (cubic DCM as above / linear ZMP hand-over with the DCM integrated
    backwards through it, which is what walking-controllers_amd/synth.py did in round 1),
  * the shipped 0.9 s step and the 1.8 s step of round 1,
  * horizons N = 50 (BASELINE) and N = 200 (shipped controllerHorizon 2 s),
and prints the worst DCM tracking error, how often the ZMP sits on the hull, and whether the DCM left the support
region (= the robot falls).  Run:  python -m oracle.gait_study
"""
from __future__ import annotations

import json
import sys

import numpy as np

from . import c_oracle as co
from . import hull_spec as hs
from . import qp_spec as qs

DT, H, G = 0.01, 0.53, 9.81
OMEGA = np.sqrt(G / H)
A = np.exp(OMEGA * DT)
B = 1.0 - A


def dcm_reference(zl, zr, n_ticks, step_ticks, ds_ticks, ds_shape):
    """DCM and ZMP reference for marching in place starting with the left foot as stance; ZMP points zl, zr (2,).
    Each step = ds_ticks of double support followed by step_ticks - ds_ticks of single support on that step's
    stance foot."""
    n_steps = n_ticks // step_ticks + 4
    T = n_steps * step_ticks
    stance = [zl if k % 2 == 0 else zr for k in range(n_steps)]
    zmp = np.zeros((T, 2)); xi = np.zeros((T + 1, 2))
    if ds_shape == "linear_zmp":
        for k in range(n_steps):
            prev = stance[k - 1] if k else stance[0]
            for s in range(step_ticks):
                lam = min(1.0, s / float(ds_ticks))
                zmp[k * step_ticks + s] = prev + lam * (stance[k] - prev)
        xi[T] = zmp[T - 1]
        for t in range(T - 1, -1, -1):
            xi[t] = (xi[t + 1] - B * zmp[t]) / A
        return xi[:n_ticks + 1], zmp[:n_ticks]
    # the planner's way: single supports backwards from the last step, cubic DCM in the double supports
    ss = step_ticks - ds_ticks
    xi_ss_end = [None] * n_steps; xi_ss_start = [None] * n_steps
    xi_ss_end[-1] = stance[-1].copy()
    for k in range(n_steps - 1, -1, -1):
        if xi_ss_end[k] is None:
            # the single support of step k ends where the double support of step k + 1 starts; the planner places that
            # boundary so that the next single support's start is reached: take the next start pulled back through the
            # double support's mean ZMP (mid-point of the two feet)
            mid = 0.5 * (stance[k] + stance[k + 1])
            xi_ss_end[k] = mid + (xi_ss_start[k + 1] - mid) * np.exp(-OMEGA * DT * ds_ticks)
        xi_ss_start[k] = stance[k] + (xi_ss_end[k] - stance[k]) * np.exp(-OMEGA * DT * ss)
    for k in range(n_steps):
        t0 = k * step_ticks
        # double support: cubic between (pos, vel) at the end of the previous single support and the start of this one
        p0 = xi_ss_end[k - 1] if k else xi_ss_start[0]
        z_prev = stance[k - 1] if k else stance[0]
        v0 = OMEGA * (p0 - z_prev)
        p1 = xi_ss_start[k]
        v1 = OMEGA * (p1 - stance[k])
        Td = ds_ticks * DT
        for s in range(ds_ticks):
            tau = s * DT
            a2 = (3 * (p1 - p0) - (2 * v0 + v1) * Td) / Td ** 2
            a3 = (-2 * (p1 - p0) + (v0 + v1) * Td) / Td ** 3
            xi[t0 + s] = p0 + v0 * tau + a2 * tau ** 2 + a3 * tau ** 3
            vel = v0 + 2 * a2 * tau + 3 * a3 * tau ** 2
            zmp[t0 + s] = xi[t0 + s] - vel / OMEGA
        for s in range(ss):
            xi[t0 + ds_ticks + s] = stance[k] + (xi_ss_start[k] - stance[k]) * np.exp(OMEGA * DT * s)
            zmp[t0 + ds_ticks + s] = stance[k]
    xi[T] = xi[T - 1]
    return xi[:n_ticks + 1], zmp[:n_ticks]


def closed_loop(horizon, step_ticks, ds_ticks, ds_shape, n_steps=8):
    mp = qs.MPCParams(horizon=horizon, sampling_time=DT, com_height=H, gravity=G)
    left_xy, right_xy = np.array([0.0, 0.08]), np.array([0.0, -0.08])
    zl, zr = left_xy + np.array([0.03, -0.005]), right_xy + np.array([0.03, 0.005])
    T = n_steps * step_ticks
    xi_ref, zmp_ref = dcm_reference(zl, zr, T + horizon + 1, step_ticks, ds_ticks, ds_shape)
    L, R = hs.foot_corners(left_xy, 0.0), hs.foot_corners(right_xy, 0.0)
    hulls = {0: hs.hull_rows(L), 1: hs.hull_rows(R), 2: hs.hull_rows(np.vstack([L, R]))}
    xi = xi_ref[0].copy(); u_prev = zmp_ref[0].copy()
    worst, on_hull, fell = 0.0, 0, False
    for t in range(T):
        k, s = divmod(t, step_ticks)
        code = 2 if s < ds_ticks else (k % 2)
        hA, hb, nc = hulls[code]
        batch = dict(x0=xi[None], ref=xi_ref[None, t:t + horizon + 1], u_prev=u_prev[None], hull_A=hA[None], hull_b=hb[None],
                     hull_nc=np.array([nc], np.int32))
        u0, _, st = co.mpc_batch_osqp(mp, batch, nthreads=1)
        u = u0[0]
        margin = np.min(hb[:nc] - hA[:nc] @ u)
        on_hull += margin < 1e-3
        xi = A * xi + B * u
        u_prev = u
        worst = max(worst, float(np.abs(xi - xi_ref[t + 1]).max()))
        if np.abs(xi - xi_ref[t + 1]).max() > 0.25:
            fell = True
            break
    return dict(horizon=horizon, step_s=step_ticks * DT, double_support_s=ds_ticks * DT, double_support_shape=ds_shape,
                worst_dcm_error_m=round(worst, 4), ticks_with_zmp_on_hull=int(on_hull), ticks=t + 1, fell=bool(fell),
                zmp_ref_max_excursion_outside_feet_m=round(float(max(0.0, np.abs(zmp_ref[:, 1]).max() - 0.105)), 4))


def main():
    rows = []
    for horizon in (50, 200):
        for step_ticks, ds_ticks in ((90, 37), (180, 110)):
            for shape in ("cubic_dcm", "linear_zmp"):
                rows.append(closed_loop(horizon, step_ticks, ds_ticks, shape))
                print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main()
