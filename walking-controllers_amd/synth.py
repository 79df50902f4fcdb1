"""
Synthetic iCub-shaped workloads in the batch
layouts of the C ABI (`include/wcqp.h`).

Every number depends only on (seed, global instance index, slot), through a
counter-based 64-bit mixer, so any shard [first, first+count) of a batch is
bit-identical to the same rows of the full batch whatever the shard count —
this is what lets the multi-GPU path be checked for shard-count invariance.

The constants are the values shipped in the reference's
app/robots/iCubGazeboV2_5/{controllerParams,qpInverseKinematics,plannerParams}.ini
(values only; no parser is reproduced here).
"""
from __future__ import annotations

import numpy as np

HULL_ROWS = 8            # hull rows are padded to 8 per instance (n_c in 4..8)
HULL_PAD_B = 1e30        # padded rows: 0*u <= +OsqpEigen::INFTY
IK_STATE_LEN = 87        # packed pose block, see IK_STATE_OFFSETS

# offsets inside the packed per-instance IK state block (doubles)
IK_STATE_OFFSETS = dict(
    p_left=0, R_left=3, p_right=12, R_right=15,
    pd_left=24, Rd_left=27, pd_right=36, Rd_right=39,
    R_neck=48, Rd_neck=57,
    com=66, com_des=69, com_vel_des=72,
    twist_left=75, twist_right=81,
)

FOOT_X = (-0.02, 0.05)       # controllerParams.ini:7  foot_size
FOOT_Y = (-0.025, 0.025)
# the four corners (x, y) of the foot rectangle in the foot frame, in the order foot_corners() walks them
FOOT_RECT = np.array([FOOT_X[1], FOOT_Y[1], FOOT_X[1], FOOT_Y[0], FOOT_X[0], FOOT_Y[0], FOOT_X[0], FOOT_Y[1]])
NOMINAL_WIDTH = 0.16         # plannerParams.ini:27

ICUB_JOINT_REG_DEG = np.array([15, 0, 0,
                               -7, 22, 11, 30,
                               -7, 22, 11, 30,
                               5.082, 0.406, -0.131, -45.249, -26.454, -0.351,
                               5.082, 0.406, -0.131, -45.249, -26.454, -0.351], float)

# Posture the WALK scenario (synth_walk_batch, per-tick kinematics) starts from and regularises to.  The shipped
# jointRegularization (ICUB_JOINT_REG_DEG) belongs to the real iCub URDF, which is not in the repository: on the
# iCub-SHAPED tree of icub_like_model() it puts the soles 15 cm ahead of the centre of mass and tilts them by 65 degrees,
# a robot no LIPM can balance.  This is the tree's own crouch: thighs 50 deg forward, knees 100 deg, soles flat under the
# CoM.  The depth is what keeps every leg away from its straight (singular) configuration while the pelvis sways over the
# other foot: tools/walk_diag.py, 65536 robots x 1024 ticks - the straightest knee stays above 31 deg and no IK becomes
# infeasible (at 35 / 70 deg 5.8 % of the robots over-stretched a leg at some point and their IK had no solution).
WALK_POSTURE_DEG = np.array([5, 0, 0,
                             -7, 22, 11, 30,
                             -7, 22, 11, 30,
                             -50.0, 0.406, -0.131, 100.0, -50.0, -0.351,
                             -50.0, 0.406, -0.131, 100.0, -50.0, -0.351], float)
# Joint velocity limits of the walk scenario [rad/s] (the reference reads them from the robot at run time,
# RobotHelper.cpp:288-298; there is no value in the repository): legs 1.5, torso and arms 0.3.  The twelve leg joints are
# all but determined by the 15 task rows, so a leg joint on its limit soon leaves the QP without a solution, while the
# upper body is where the QP has freedom: with these limits a bound is active on ~12 % of the robot-ticks (the upper
# body's, mostly while the pelvis changes sides) and none of 65536 robots fails in 1024 ticks.
WALK_VMAX = np.array([0.3] * 11 + [1.5] * 12, float)
# Swing foot of the walk scenario: LIFTED - the vertical twist amplitude is positive (the zero-net-displacement profile of
# tick_device.h: up in the first half of the single support, down in the second; 0.04-0.08 m/s peak = 0.9-1.8 cm of lift),
# small horizontal / angular scatter.  (An isotropic random twist pushes the foot down and outwards as often as up: that
# stretches the leg.)
WALK_SWING_LIFT = (0.04, 0.08)
WALK_SWING_LIN_SIGMA = 0.01
WALK_SWING_ANG_SIGMA = 0.02

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= _M2
    x ^= x >> np.uint64(27)
    x *= _M3
    x ^= x >> np.uint64(31)
    return x


class CounterRNG:
    """Stateless counter-based generator: value(seed, instance, slot)."""

    def __init__(self, seed: int, first: int, count: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.inst = np.arange(first, first + count, dtype=np.uint64)
        self._slot = 0

    def _bits(self, k: int) -> np.ndarray:
        slots = np.arange(self._slot, self._slot + k, dtype=np.uint64)
        self._slot += k
        with np.errstate(over="ignore"):
            base = _mix(self.inst * _M1 + self.seed)[:, None]
            return _mix(base + (slots[None, :] + np.uint64(1)) * _M3)

    def uniform(self, k: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        u = (self._bits(k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, k: int, sigma: float = 1.0) -> np.ndarray:
        k2 = (k + 1) // 2
        u1 = 1.0 - self.uniform(k2)          # (0, 1]
        u2 = self.uniform(k2)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], axis=1)
        return sigma * z[:, :k]


# --------------------------------------------------------------------------------------
# support polygon -> hull rows  (our own convention, SURVEY Appendix D-4:
# CCW hull, unit outward normals, rows a.u <= b, padded to 8 with 0.u <= 1e30)
# --------------------------------------------------------------------------------------
def convex_hull_ccw(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain; returns CCW vertices without collinear points."""
    P = sorted(map(tuple, np.asarray(pts, float)))
    if len(P) <= 2:
        return np.array(P)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lo, up = [], []
    for p in P:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], p) <= 0:
            lo.pop()
        lo.append(p)
    for p in reversed(P):
        while len(up) >= 2 and cross(up[-2], up[-1], p) <= 0:
            up.pop()
        up.append(p)
    return np.array(lo[:-1] + up[:-1])


def hull_rows(pts: np.ndarray):
    """rows (A[8,2], b[8], nc) of the convex hull of 2-D points."""
    V = convex_hull_ccw(pts)
    nc = len(V)
    assert 3 <= nc <= HULL_ROWS, nc
    A = np.zeros((HULL_ROWS, 2))
    b = np.full(HULL_ROWS, HULL_PAD_B)
    for k in range(nc):
        v0, v1 = V[k], V[(k + 1) % nc]
        d = v1 - v0
        nrm = np.array([d[1], -d[0]]) / np.hypot(d[0], d[1])
        A[k] = nrm
        b[k] = nrm @ v0
    return A, b, nc


def foot_corners(pos_xy: np.ndarray, yaw: float) -> np.ndarray:
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s], [s, c]])
    loc = np.array([[FOOT_X[1], FOOT_Y[1]], [FOOT_X[1], FOOT_Y[0]],
                    [FOOT_X[0], FOOT_Y[0]], [FOOT_X[0], FOOT_Y[1]]])
    return loc @ R.T + pos_xy


def synth_mpc_batch(count: int, seed: int = 1234, horizon: int = 50, first: int = 0,
                    dT: float = 0.01, x0_sigma: float = 0.01, uprev_sigma: float = 0.005):
    """SURVEY.md §8d config 2.  Returns dict of C-contiguous arrays."""
    N = horizon
    rng = CounterRNG(seed, first, count)
    u_state = rng.uniform(1)[:, 0]
    yaw = rng.uniform(2, -0.3, 0.3)
    other_x = rng.uniform(1, -0.1, 0.1)[:, 0]
    other_y = rng.uniform(1, -0.02, 0.02)[:, 0]
    x0_n = rng.normal(2, x0_sigma)
    v = rng.uniform(2, -0.2, 0.2)
    up_n = rng.normal(2, uprev_sigma)
    ref_n = rng.normal(2 * (N + 1), 1e-4).reshape(count, N + 1, 2)

    hull_A = np.zeros((count, HULL_ROWS, 2))
    hull_b = np.zeros((count, HULL_ROWS))
    hull_nc = np.zeros(count, np.int32)
    centroid = np.zeros((count, 2))
    contact = np.zeros(count, np.int32)       # 0 = left only, 1 = right only, 2 = both
    for i in range(count):
        st = 0 if u_state[i] < 0.35 else (1 if u_state[i] < 0.70 else 2)
        contact[i] = st
        # stance foot at the origin; the other foot nominal-width away (left is +y)
        if st == 0:
            pts = foot_corners(np.zeros(2), yaw[i, 0])
        elif st == 1:
            pts = foot_corners(np.zeros(2), yaw[i, 1])
        else:
            left = foot_corners(np.zeros(2), yaw[i, 0])
            right = foot_corners(np.array([other_x[i], -NOMINAL_WIDTH + other_y[i]]), yaw[i, 1])
            pts = np.vstack([left, right])
        A, b, nc = hull_rows(pts)
        hull_A[i], hull_b[i], hull_nc[i] = A, b, nc
        centroid[i] = convex_hull_ccw(pts).mean(axis=0)
    x0 = centroid + x0_n
    stage = np.arange(N + 1, dtype=float)[None, :, None]
    ref = x0[:, None, :] + stage * v[:, None, :] * dT + ref_n
    u_prev = centroid + up_n
    return dict(x0=np.ascontiguousarray(x0), ref=np.ascontiguousarray(ref),
                u_prev=np.ascontiguousarray(u_prev), hull_A=hull_A, hull_b=hull_b,
                hull_nc=hull_nc, contact=contact, centroid=centroid)


# --------------------------------------------------------------------------------------
def _skew(p: np.ndarray) -> np.ndarray:
    z = np.zeros(p.shape[0])
    return np.stack([np.stack([z, -p[:, 2], p[:, 1]], -1),
                     np.stack([p[:, 2], z, -p[:, 0]], -1),
                     np.stack([-p[:, 1], p[:, 0], z], -1)], -2)


def _small_rot(w: np.ndarray) -> np.ndarray:
    """Rodrigues for a batch of rotation vectors (count, 3)."""
    th = np.linalg.norm(w, axis=1)
    th_safe = np.where(th > 0, th, 1.0)
    K = _skew(w / th_safe[:, None])
    I = np.eye(3)[None]
    s = np.sin(th)[:, None, None]
    c = (1 - np.cos(th))[:, None, None]
    return I + s * K + c * (K @ K)


def _rotz(a: np.ndarray) -> np.ndarray:
    c, s, z, o = np.cos(a), np.sin(a), np.zeros_like(a), np.ones_like(a)
    return np.stack([np.stack([c, -s, z], -1), np.stack([s, c, z], -1), np.stack([z, z, o], -1)], -2)


def synth_ik_batch(count: int, seed: int = 4321, dof: int = 23, first: int = 0,
                   joint_sigma: float = 0.3, com_sigma: float = 0.05, additional_rotation=None, posture_deg=None):
    """SURVEY.md §8d config 3 (dense random joint columns, exact mixed-representation
    base blocks).  Returns dict of C-contiguous arrays in ABI layout.
    additional_rotation / posture_deg: another robot's qpInverseKinematics.ini values (default: iCubGazeboV2_5's; the
    random draws do not depend on them)."""
    n = dof + 6
    rng = CounterRNG(seed, first, count)
    o = IK_STATE_OFFSETS

    def jac6():
        p = rng.normal(3, 0.3)
        J = np.zeros((count, 6, n))
        J[:, 0:3, 0:3] = np.eye(3)
        J[:, 0:3, 3:6] = -_skew(p)
        J[:, 3:6, 3:6] = np.eye(3)
        J[:, :, 6:] = rng.normal(6 * dof, joint_sigma).reshape(count, 6, dof)
        return J

    J_left, J_right, J_neck6 = jac6(), jac6(), jac6()
    pc = rng.normal(3, 0.3)
    J_com = np.zeros((count, 3, n))
    J_com[:, :, 0:3] = np.eye(3)
    J_com[:, :, 3:6] = -_skew(pc)
    J_com[:, :, 6:] = rng.normal(3 * dof, com_sigma).reshape(count, 3, dof)

    q_reg = np.deg2rad((ICUB_JOINT_REG_DEG if posture_deg is None else np.asarray(posture_deg, float)) if dof == 23 else np.zeros(dof))
    q = q_reg[None, :] + rng.normal(dof, 0.05)

    state = np.zeros((count, IK_STATE_LEN))
    yaw = rng.uniform(3, -0.3, 0.3)
    # desired feet: left at origin-ish, right nominal-width away; small yaw
    pd_left = np.concatenate([rng.uniform(2, -0.1, 0.1), np.zeros((count, 1))], 1)
    pd_right = pd_left + np.array([0.0, -NOMINAL_WIDTH, 0.0]) + \
        np.concatenate([rng.uniform(2, -0.05, 0.05), rng.uniform(1, 0.0, 0.03)], 1)
    Rd_left, Rd_right = _rotz(yaw[:, 0]), _rotz(yaw[:, 1])
    R_left = _small_rot(rng.normal(3, 0.01)) @ Rd_left
    R_right = _small_rot(rng.normal(3, 0.01)) @ Rd_right
    p_left = pd_left + rng.normal(3, 0.005)
    p_right = pd_right + rng.normal(3, 0.005)
    # setDesiredNeckOrientation right-multiplies additional_rotation (WM/src/WalkingQPInverseKinematics.cpp:143-146)
    add_rot = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]) if additional_rotation is None else np.asarray(additional_rotation, float)
    Rd_neck = _rotz(yaw[:, 2]) @ add_rot
    R_neck = _small_rot(rng.normal(3, 0.02)) @ Rd_neck
    com_des = 0.5 * (pd_left + pd_right) + np.array([0.0, 0.0, 0.53])
    com = com_des + rng.normal(3, 0.005)
    com_vel = rng.normal(3, 0.05)
    stance_left = rng.uniform(1)[:, 0] < 0.5
    tw = rng.normal(6, 0.2)
    twist_left = np.where(stance_left[:, None], 0.0, tw)
    twist_right = np.where(stance_left[:, None], tw, 0.0)

    def put(name, arr):
        a = arr.reshape(count, -1)
        state[:, o[name]:o[name] + a.shape[1]] = a

    put("p_left", p_left); put("R_left", R_left); put("p_right", p_right); put("R_right", R_right)
    put("pd_left", pd_left); put("Rd_left", Rd_left); put("pd_right", pd_right); put("Rd_right", Rd_right)
    put("R_neck", R_neck); put("Rd_neck", Rd_neck)
    put("com", com); put("com_des", com_des); put("com_vel_des", com_vel)
    put("twist_left", twist_left); put("twist_right", twist_right)

    return dict(J_left=np.ascontiguousarray(J_left), J_right=np.ascontiguousarray(J_right),
                J_neck=np.ascontiguousarray(J_neck6[:, 3:6, :]), J_com=np.ascontiguousarray(J_com),
                q=np.ascontiguousarray(q), state=state)


# --------------------------------------------------------------------------------------
def _dcm_reference(zl, zr, phase0, T, step_ticks, ds_ticks, dT, com_height, gravity):
    """ZMP reference that sits on the stance foot's ZMP point (zl / zr) in single support and moves linearly to the
    next stance foot in double support; DCM reference integrated BACKWARDS through xi_t = (xi_{t+1} - b zmp_t) / a
    (the way the reference's planner builds it, WM/src/TrajectoryGenerator.cpp:152-154).  Returns (xi[:, :T], zmp[:, :T])."""
    count = zl.shape[0]
    Tz = T + 4 * step_ticks                                   # tail so that the backward pass has settled
    tau = np.arange(Tz)[None, :]
    cyc = (tau + phase0[:, None]) % (2 * step_ticks)
    sidx = cyc % step_ticks
    side = cyc // step_ticks                                  # 0: left is the stance foot of this step
    cur = np.where(side[..., None] == 0, zl[:, None, :], zr[:, None, :])
    prev = np.where(side[..., None] == 0, zr[:, None, :], zl[:, None, :])
    lam = np.clip(sidx / float(ds_ticks), 0.0, 1.0)[..., None]
    zmp = prev + lam * (cur - prev)
    a = np.exp(np.sqrt(gravity / com_height) * dT); b = 1.0 - a
    xi = np.zeros((count, Tz, 2))
    xi[:, -1] = zmp[:, -1]
    for t in range(Tz - 2, -1, -1):
        xi[:, t] = (xi[:, t + 1] - b * zmp[:, t]) / a
    return np.ascontiguousarray(xi[:, :T]), np.ascontiguousarray(zmp[:, :T])


def synth_tick_batch(count: int, n_ticks: int, seed: int = 2718, horizon: int = 50, first: int = 0,
                     step_ticks: int = 180, ds_ticks: int = 110, dT: float = 0.01, com_height: float = 0.53,
                     gravity: float = 9.81, additional_rotation=None):
    """Inputs of the device-resident tick pipeline (BASELINE configs 4/5): per instance a long
    DCM reference trajectory (the deque the reference consumes one stage per tick), the
    support-polygon rows of the three contact pairs (left, right, both), a random phase
    offset of the step cycle, constant Jacobians/poses for the IK, initial states.

    The DCM reference is dynamically consistent, the way the reference's planner builds it
    (WM/src/TrajectoryGenerator.cpp:152-154): a ZMP reference that sits on the stance foot
    (+ leftZMPDelta / rightZMPDelta, plannerParams.ini:39-40) in single support and moves
    linearly to the next stance foot in double support, integrated BACKWARDS through
    xi_t = (xi_{t+1} - b zmp_t) / a."""
    rng = CounterRNG(seed ^ 0x5EED, first, count)
    T = n_ticks + horizon + 1
    yaw = rng.uniform(2, -0.2, 0.2)
    dxy = np.concatenate([rng.uniform(1, -0.05, 0.05), rng.uniform(1, -0.01, 0.01)], 1)
    phase0 = (rng.uniform(1)[:, 0] * (2 * step_ticks)).astype(np.int32)
    dcm_n = rng.normal(2, 0.002)
    swing = rng.normal(6, 0.2)
    left_xy = np.zeros((count, 2)); left_xy[:, 1] = 0.5 * NOMINAL_WIDTH
    right_xy = dxy.copy(); right_xy[:, 1] -= 0.5 * NOMINAL_WIDTH
    hull_tab_A = np.zeros((count, 3, HULL_ROWS, 2)); hull_tab_b = np.zeros((count, 3, HULL_ROWS))
    hull_tab_nc = np.zeros((count, 3), np.int32)
    for i in range(count):
        L = foot_corners(left_xy[i], yaw[i, 0]); R = foot_corners(right_xy[i], yaw[i, 1])
        for k, pts in enumerate((L, R, np.vstack([L, R]))):
            A, b, nc = hull_rows(pts)
            hull_tab_A[i, k], hull_tab_b[i, k], hull_tab_nc[i, k] = A, b, nc

    def rot2(a, v):
        return np.stack([np.cos(a) * v[0] - np.sin(a) * v[1], np.sin(a) * v[0] + np.cos(a) * v[1]], -1)
    zl = left_xy + rot2(yaw[:, 0], (0.03, -0.005))            # leftZMPDelta
    zr = right_xy + rot2(yaw[:, 1], (0.03, 0.005))            # rightZMPDelta
    ref, zmp = _dcm_reference(zl, zr, phase0, T, step_ticks, ds_ticks, dT, com_height, gravity)
    ik = synth_ik_batch(count, seed=seed + 1, first=first, additional_rotation=additional_rotation)
    dcm0 = ref[:, 0, :] + dcm_n
    return dict(first=first, ref_traj=ref, zmp_ref=zmp, hull_tab_A=hull_tab_A,
                hull_tab_b=hull_tab_b, hull_tab_nc=hull_tab_nc, phase0=phase0, J_left=ik["J_left"],
                J_right=ik["J_right"], J_neck=ik["J_neck"], J_com=ik["J_com"], state0=ik["state"],
                swing_twist=np.ascontiguousarray(swing), q0=ik["q"], dcm0=np.ascontiguousarray(dcm0),
                com0=np.ascontiguousarray(dcm0.copy()), u_init=np.ascontiguousarray(zmp[:, 0].copy()))


# ------------------------------------------------------------------------------------------------
# kinematics.  The reference reads its robot model from an external URDF
# (`model.urdf`, WM/src/WalkingModule.cpp:107) that is not part of the repository, so the tree below
# is iCub-SHAPED, not iCub: same joint list and order (CFG/robotControl.ini:3-7), same chain
# topology (torso 3 -> arms 4 + 4; legs 6 + 6 from the root link), plausible link lengths, masses
# and axes.  A table derived from the real URDF has the same form and goes through the same ABI.
JOINT_NAMES = ("torso_pitch", "torso_roll", "torso_yaw",
               "l_shoulder_pitch", "l_shoulder_roll", "l_shoulder_yaw", "l_elbow",
               "r_shoulder_pitch", "r_shoulder_roll", "r_shoulder_yaw", "r_elbow",
               "l_hip_pitch", "l_hip_roll", "l_hip_yaw", "l_knee", "l_ankle_pitch", "l_ankle_roll",
               "r_hip_pitch", "r_hip_roll", "r_hip_yaw", "r_knee", "r_ankle_pitch", "r_ankle_roll")


def _rot_axis(axis, angle):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def icub_like_model() -> dict:
    """Kinematic tree in the table form of `wcqp_kin_params`: joint j has parent joint `parent[j]` (-1 = root
    link, always < j), a fixed transform (R0, p0) from the parent joint frame to its own frame at q = 0, a unit
    axis in its own frame, and carries one link (mass, centre of mass in the joint frame).  Three frames are
    attached: left sole, right sole, neck."""
    X, Y, Z = (1.0, 0, 0), (0, 1.0, 0), (0, 0, 1.0)
    parent = [-1, 0, 1, 2, 3, 4, 5, 2, 7, 8, 9, -1, 11, 12, 13, 14, 15, -1, 17, 18, 19, 20, 21]
    axis = [Y, X, Z, Y, X, Z, Y, Y, X, Z, Y, Y, X, Z, Y, Y, X, Y, X, Z, Y, Y, X]
    p0 = [(0, 0, 0.12), (0, 0, 0.03), (0, 0, 0.03),
          (0.0, 0.11, 0.14), (0, 0.02, 0), (0, 0, -0.07), (0.015, 0, -0.08),
          (0.0, -0.11, 0.14), (0, -0.02, 0), (0, 0, -0.07), (0.015, 0, -0.08),
          (0, 0.068, -0.04), (0, 0, -0.01), (0, 0, -0.03), (0, 0, -0.22), (0, 0, -0.21), (0, 0, -0.02),
          (0, -0.068, -0.04), (0, 0, -0.01), (0, 0, -0.03), (0, 0, -0.22), (0, 0, -0.21), (0, 0, -0.02)]
    R0 = [np.eye(3) for _ in range(23)]
    R0[3] = _rot_axis(X, np.deg2rad(15.0)); R0[7] = _rot_axis(X, np.deg2rad(-15.0))      # shoulders tilted outwards
    R0[13] = _rot_axis(Z, np.deg2rad(2.0)); R0[19] = _rot_axis(Z, np.deg2rad(-2.0))
    mass = [1.2, 0.9, 4.5, 1.1, 0.4, 0.9, 0.8, 1.1, 0.4, 0.9, 0.8,
            1.4, 0.6, 1.9, 1.3, 0.5, 0.7, 1.4, 0.6, 1.9, 1.3, 0.5, 0.7]
    com = [(0, 0, 0.01), (0, 0, 0.01), (0.005, 0, 0.10), (0, 0, -0.02), (0, 0, 0), (0, 0, -0.04), (0.01, 0, -0.06),
           (0, 0, -0.02), (0, 0, 0), (0, 0, -0.04), (0.01, 0, -0.06),
           (0, 0, -0.01), (0, 0, 0), (0, 0, -0.10), (0, 0, -0.10), (0, 0, -0.01), (0.02, 0, -0.03),
           (0, 0, -0.01), (0, 0, 0), (0, 0, -0.10), (0, 0, -0.10), (0, 0, -0.01), (0.02, 0, -0.03)]
    return dict(dof=23, parent=np.array(parent, np.int32), axis=np.array(axis, float), p0=np.array(p0, float),
                R0=np.stack(R0), mass=np.array(mass, float), com=np.array(com, float),
                root_mass=5.8, root_com=np.array([-0.01, 0.0, 0.02]),
                frame_joint=np.array([16, 22, 2], np.int32),          # left sole, right sole, neck
                frame_p=np.array([(0.02, 0, -0.045), (0.02, 0, -0.045), (0, 0, 0.20)], float),
                frame_R=np.stack([np.eye(3), np.eye(3), _rot_axis(Y, np.deg2rad(5.0))]))


def synth_kin_batch(count: int, seed: int = 31415, first: int = 0, joint_sigma: float = 0.25, posture_deg=None,
                    base_rot_sigma: float = 0.15) -> dict:
    """Base poses ([B][12]: position, row-major rotation) and joint angles [B][23] around the posture the
    reference regularises to (ICUB_JOINT_REG_DEG), shard-invariant like the other generators.  (A walk starts close
    to that posture - joint_sigma ~0.04 - or the regularisation term alone asks for rad/s of joint velocity.)"""
    rng = CounterRNG(seed, first, count)
    q = np.deg2rad(ICUB_JOINT_REG_DEG if posture_deg is None else posture_deg)[None, :] + rng.normal(23, joint_sigma)
    pos = rng.normal(3, 1.0) * np.array([0.05, 0.05, 0.02]) + np.array([0.0, 0.0, 0.55])
    w = rng.normal(3, base_rot_sigma)
    R = np.stack([_rot_axis(wi / (np.linalg.norm(wi) + 1e-300), np.linalg.norm(wi)) for wi in w])
    base = np.concatenate([pos, R.reshape(count, 9)], axis=1)
    return dict(q=np.ascontiguousarray(q), base=np.ascontiguousarray(base))


def synth_walk_kin_batch(count: int, first: int = 0) -> dict:
    """Base poses / joint angles the walk scenario starts from: robots standing upright on the tree's own crouch
    (WALK_POSTURE_DEG), a degree of scatter."""
    return synth_kin_batch(count, seed=27182, first=first, joint_sigma=0.015, posture_deg=WALK_POSTURE_DEG, base_rot_sigma=0.015)


def synth_walk_batch(count: int, n_ticks: int, poses: np.ndarray, kin_batch: dict, seed: int = 2718, horizon: int = 50,
                     first: int = 0, step_ticks: int = 180, ds_ticks: int = 110, dT: float = 0.01, com_height: float = 0.53,
                     gravity: float = 9.81):
    """Inputs of the tick pipeline WITH per-tick kinematics (`TickPipeline(kin=...)`): a coherent synthetic robot
    marching in place.  `kin_batch` = synth_walk_kin_batch(count, first=first) (base poses, joint angles), `poses` [count][87] =
    the pose block the kinematics produce for it at tick 0 (`KinModel.jacobians_host(base, q, state=zeros)["state"]` on
    the device, or the oracle's twin in the tests): the desired foot poses ARE the initial ones (the feet stay planted;
    the pipeline anchors the base at the stance foot, so it moves as the stance leg's joints do), the desired neck orientation is the initial one up to a small rotation, the ZMP
    reference alternates between the two feet's ZMP points (plannerParams.ini:39-40) and the DCM reference follows from it."""
    rng = CounterRNG(seed ^ 0xA11CE, first, count)
    T = n_ticks + horizon + 1
    o = IK_STATE_OFFSETS
    # the walk starts from standing in the middle of a double-support phase (either foot next), where the DCM reference
    # passes between the feet - close to where the robot's own CoM is; contact pairs then change every 70-110 ticks
    u_ph = rng.uniform(2)
    phase0 = ((u_ph[:, 0] < 0.5) * step_ticks + (0.35 + 0.2 * u_ph[:, 1]) * ds_ticks).astype(np.int32)
    dcm_n = rng.normal(2, 0.002)
    # amplitudes of the swing foot's twist profile (zero net displacement: tick_device.h): the foot is lifted
    swing = np.concatenate([rng.normal(2, WALK_SWING_LIN_SIGMA), rng.uniform(1, *WALK_SWING_LIFT), rng.normal(3, WALK_SWING_ANG_SIGMA)], axis=1)
    neck_w = rng.normal(3, 0.02)
    state0 = np.array(poses, dtype=np.float64, copy=True)
    for src, dst, k in (("p_left", "pd_left", 3), ("R_left", "Rd_left", 9), ("p_right", "pd_right", 3), ("R_right", "Rd_right", 9)):
        state0[:, o[dst]:o[dst] + k] = state0[:, o[src]:o[src] + k]
    Rn = state0[:, o["R_neck"]:o["R_neck"] + 9].reshape(count, 3, 3)
    state0[:, o["Rd_neck"]:o["Rd_neck"] + 9] = (_small_rot(neck_w) @ Rn).reshape(count, 9)

    def zmp_point(p_name, R_name, delta):
        p = state0[:, o[p_name]:o[p_name] + 2]
        R = state0[:, o[R_name]:o[R_name] + 9].reshape(count, 3, 3)
        return p + np.stack([R[:, 0, 0] * delta[0] + R[:, 0, 1] * delta[1], R[:, 1, 0] * delta[0] + R[:, 1, 1] * delta[1]], -1)
    zl = zmp_point("p_left", "R_left", (0.03, -0.005))        # leftZMPDelta
    zr = zmp_point("p_right", "R_right", (0.03, 0.005))       # rightZMPDelta
    ref, zmp = _dcm_reference(zl, zr, phase0, T, step_ticks, ds_ticks, dT, com_height, gravity)
    dcm0 = ref[:, 0, :] + dcm_n
    return dict(first=first, ref_traj=ref, zmp_ref=zmp, phase0=phase0, state0=np.ascontiguousarray(state0),
                swing_twist=np.ascontiguousarray(swing), q0=np.ascontiguousarray(kin_batch["q"]),
                dcm0=np.ascontiguousarray(dcm0),
                # the plant's CoM starts where the robot's own (kinematic) CoM is
                com0=np.ascontiguousarray(state0[:, o["com"]:o["com"] + 2].copy()), u_init=np.ascontiguousarray(zmp[:, 0].copy()))
