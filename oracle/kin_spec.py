"""
ORACLE (test infrastructure only; never imported by the product path).

Forward kinematics and free-floating Jacobians in MIXED representation for a kinematic tree given
in the table form of `wcqp_kin_params` — what the reference obtains from
iDynTree::KinDynComputations through `WalkingFK` (WM/src/WalkingForwardKinematics.cpp):
  setInternalRobotState            :258-276   base transform + joint positions
  getLeft/RightFootToWorldTransform:354-366, getNeckOrientation :402-405, evaluateCoM :312-340
  get{Left,Right}FootJacobian, getNeckJacobian, getCoMJacobian :436-454 (MIXED, :33)
MIXED: linear velocity of the frame origin and angular velocity both expressed in the world frame;
generalised velocity nu = (v_base_origin^W, omega_base^W, dq).  For a frame f with world origin
p_f rigidly attached after joint j_f:
   base columns   [ I3  -S(p_f - p_b) ;  0  I3 ]
   joint column i [ a_i x (p_f - p_i) ;  a_i ]   if joint i lies on the path root -> j_f, else 0
with a_i the joint axis and p_i the joint origin in world.  CoM: mass-weighted mean of the link
CoM Jacobians = for joint i  (m_sub(i) / M) a_i x (c_sub(i) - p_i), sub(i) = links moved by joint i.

PARITY UNPINNED: the reference's robot model (model.urdf) and iDynTree are not in the repository;
the model tables are iCub-shaped synthetic data (walking-controllers_amd/synth.py).  The functions
below are pinned against themselves: `numeric_jacobians` differentiates the forward kinematics.
"""
from __future__ import annotations

import numpy as np


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _rot(axis, angle):
    K = _skew(np.asarray(axis, float))
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def forward(model: dict, base: np.ndarray, q: np.ndarray) -> dict:
    """World rotation / origin of every joint frame (after its own rotation), joint axes in world, link CoMs,
    the three attached frames and the total CoM.  base = [p(3), R(9 row-major)]."""
    n = model["dof"]
    pb, Rb = np.asarray(base[:3], float), np.asarray(base[3:12], float).reshape(3, 3)
    R = np.zeros((n, 3, 3)); p = np.zeros((n, 3)); a = np.zeros((n, 3))
    for j in range(n):
        par = int(model["parent"][j])
        Rp, pp = (Rb, pb) if par < 0 else (R[par], p[par])
        Rpre = Rp @ model["R0"][j]
        p[j] = pp + Rp @ model["p0"][j]
        a[j] = Rpre @ model["axis"][j]
        R[j] = Rpre @ _rot(model["axis"][j], q[j])
    c_link = p + np.einsum("jab,jb->ja", R, model["com"])
    c_root = pb + Rb @ model["root_com"]
    M = model["root_mass"] + model["mass"].sum()
    com = (model["root_mass"] * c_root + (model["mass"][:, None] * c_link).sum(0)) / M
    frames = []
    for f in range(3):
        jf = int(model["frame_joint"][f])
        frames.append((p[jf] + R[jf] @ model["frame_p"][f], R[jf] @ model["frame_R"][f]))
    return dict(R=R, p=p, a=a, c_link=c_link, com=com, M=M, frames=frames, pb=pb, Rb=Rb)


def _path(model, j):
    out = []
    while j >= 0:
        out.append(j); j = int(model["parent"][j])
    return out


def jacobians(model: dict, base: np.ndarray, q: np.ndarray) -> dict:
    """J_left, J_right (6 x (6+n)), J_neck (3 x (6+n), angular rows), J_com (3 x (6+n)) and the poses."""
    n = model["dof"]
    k = forward(model, base, q)
    Js = []
    for f in range(3):
        pf, _ = k["frames"][f]
        J = np.zeros((6, 6 + n))
        J[:3, :3] = np.eye(3); J[:3, 3:6] = -_skew(pf - k["pb"]); J[3:, 3:6] = np.eye(3)
        for i in _path(model, int(model["frame_joint"][f])):
            J[:3, 6 + i] = np.cross(k["a"][i], pf - k["p"][i])
            J[3:, 6 + i] = k["a"][i]
        Js.append(J)
    # subtree masses / first moments
    msub = model["mass"].astype(float).copy(); csub = model["mass"][:, None] * k["c_link"]
    for j in range(n - 1, -1, -1):
        par = int(model["parent"][j])
        if par >= 0:
            msub[par] += msub[j]; csub[par] += csub[j]
    Jc = np.zeros((3, 6 + n))
    Jc[:, :3] = np.eye(3); Jc[:, 3:6] = -_skew(k["com"] - k["pb"])
    for i in range(n):
        Jc[:, 6 + i] = np.cross(k["a"][i], csub[i] - msub[i] * k["p"][i]) / k["M"]
    return dict(J_left=Js[0], J_right=Js[1], J_neck=Js[2][3:], J_com=Jc,
                p_left=k["frames"][0][0], R_left=k["frames"][0][1], p_right=k["frames"][1][0], R_right=k["frames"][1][1],
                R_neck=k["frames"][2][1], com=k["com"])


def _log_rot(R):
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return w          # first order is all the finite differences need


def numeric_jacobians(model: dict, base: np.ndarray, q: np.ndarray, h: float = 1e-6) -> dict:
    """Central differences of `forward` along each generalised velocity direction (mixed representation)."""
    n = model["dof"]

    def moved(k_dir, step):
        b = np.asarray(base, float).copy(); qq = np.asarray(q, float).copy()
        if k_dir < 3:
            b[k_dir] += step
        elif k_dir < 6:
            w = np.zeros(3); w[k_dir - 3] = step
            Rb = b[3:12].reshape(3, 3)
            b[3:12] = (_rot(w / abs(step), abs(step)) @ Rb if step != 0 else Rb).reshape(9)
        else:
            qq[k_dir - 6] += step
        return forward(model, b, qq)
    k0 = forward(model, base, q)
    out = dict(J_left=np.zeros((6, 6 + n)), J_right=np.zeros((6, 6 + n)), J_neck=np.zeros((3, 6 + n)), J_com=np.zeros((3, 6 + n)))
    for d in range(6 + n):
        kp, km = moved(d, h), moved(d, -h)
        for name, f in (("J_left", 0), ("J_right", 1)):
            out[name][:3, d] = (kp["frames"][f][0] - km["frames"][f][0]) / (2 * h)
            out[name][3:, d] = (_log_rot(kp["frames"][f][1] @ k0["frames"][f][1].T) - _log_rot(km["frames"][f][1] @ k0["frames"][f][1].T)) / (2 * h)
        out["J_neck"][:, d] = (_log_rot(kp["frames"][2][1] @ k0["frames"][2][1].T) - _log_rot(km["frames"][2][1] @ k0["frames"][2][1].T)) / (2 * h)
        out["J_com"][:, d] = (kp["com"] - km["com"]) / (2 * h)
    return out
