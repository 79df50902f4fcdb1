"""
ctypes binding of the C ABI declared in include/wcqp.h.

This is host-side plumbing only: every solve goes through libwcqp.so's HIP
kernels.  There is no CPU fallback — if the library is missing the import
fails, and if no GPU is present the solve entry points return WCQP_E_HIP and
`check()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WCQP_LIB_PATH") or os.path.join(_HERE, "libwcqp.so")   # override: diagnostic builds only

WCQP_OK = 0
STATUS_SOLVED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_OUTSIDE_HULL, STATUS_NUMERIC, STATUS_STRUCTURE = range(6)
IK_FORM_QPOASES, IK_FORM_OSQP = 0, 1
IK_ALG_DEFAULT, IK_ALG_SWEEP, IK_ALG_NULLSPACE, IK_ALG_NULLSPACE_MFMA, IK_ALG_NULLSPACE_16L, IK_ALG_BASE_ELIM = 0, 1, 2, 3, 4, 5
IK_JAC_AUTO, IK_JAC_MIXED, IK_JAC_GENERAL = 0, 1, 2
KIN_HANDOFF_FUSED, KIN_HANDOFF_DENSE, KIN_HANDOFF_COMPACT = 0, 1, 2
HULL_ROWS = 8
MAX_DOF = 32
IK_STATE_LEN = 87

# every symbol include/wcqp.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "wcqp_strerror", "wcqp_version", "wcqp_device_count", "wcqp_stream_create", "wcqp_stream_destroy", "wcqp_stream_synchronize",
    "wcqp_mpc_create", "wcqp_mpc_destroy", "wcqp_mpc_get_condensed", "wcqp_mpc_get_matrices",
    "wcqp_mpc_solve_device", "wcqp_mpc_solve_host",
    "wcqp_ik_create", "wcqp_ik_destroy", "wcqp_ik_set_posture", "wcqp_ik_solve_device", "wcqp_ik_solve_host",
    "wcqp_hull_from_feet_device", "wcqp_hull_from_feet_host",
    "wcqp_kin_create", "wcqp_kin_destroy", "wcqp_kin_jacobians_device", "wcqp_kin_jacobians_host",
    "wcqp_tick_create", "wcqp_tick_destroy", "wcqp_tick_upload", "wcqp_tick_run", "wcqp_tick_download", "wcqp_tick_splice_reference",
    "wcqp_tick_set_feedback_device", "wcqp_tick_set_feedback_host",
    "wcqp_qp_enqueue_steps", "wcqp_qp_plan_create", "wcqp_qp_plan_enqueue", "wcqp_qp_plan_destroy",
    "wcqp_slab_layout_for", "wcqp_qp_step_from_slabs",
)


class WcqpError(RuntimeError):
    pass


class MpcParams(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("sampling_time", C.c_double), ("com_height", C.c_double),
                ("gravity", C.c_double), ("Q", C.c_double * 4), ("R", C.c_double * 4),
                ("convex_hull_tolerance", C.c_double), ("feas_tol", C.c_double)]


class IkParams(C.Structure):
    _fields_ = [("dof", C.c_int32), ("use_com_as_constraint", C.c_int32), ("form", C.c_int32),
                ("max_iter", C.c_int32),
                ("com_weight", C.c_double * 9), ("neck_weight", C.c_double * 9),
                ("joint_reg_weights", C.c_double * MAX_DOF), ("joint_reg_gains", C.c_double * MAX_DOF),
                ("joint_reg_rad", C.c_double * MAX_DOF),
                ("v_min", C.c_double * MAX_DOF), ("v_max", C.c_double * MAX_DOF),
                ("k_pos_com", C.c_double), ("k_pos_foot", C.c_double),
                ("k_att_foot", C.c_double), ("k_neck", C.c_double),
                ("rho", C.c_double), ("tol", C.c_double), ("algorithm", C.c_int32),
                ("jacobian_structure", C.c_int32)]


class QpStep(C.Structure):
    """wcqp_qp_step: the arguments of one wcqp_mpc_solve_device + one wcqp_ik_solve_device call (raw device addresses)."""
    _fields_ = [("x0", C.c_void_p), ("ref", C.c_void_p), ("ref_len", C.c_int32), ("u_prev", C.c_void_p),
                ("hull_A", C.c_void_p), ("hull_b", C.c_void_p), ("hull_nc", C.c_void_p),
                ("u0", C.c_void_p), ("mpc_status", C.c_void_p), ("mpc_active", C.c_void_p), ("mpc_margin", C.c_void_p),
                ("mpc_stream", C.c_void_p),
                ("J_left", C.c_void_p), ("J_right", C.c_void_p), ("J_neck", C.c_void_p), ("J_com", C.c_void_p),
                ("q", C.c_void_p), ("state", C.c_void_p),
                ("dq", C.c_void_p), ("ik_status", C.c_void_p), ("active_lower", C.c_void_p), ("active_upper", C.c_void_p),
                ("foot_err", C.c_void_p), ("iters", C.c_void_p), ("ik_stream", C.c_void_p)]


SLAB_IN = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc", "J_left", "J_right", "J_neck", "J_com", "q", "state")
SLAB_OUT = ("u0", "mpc_margin", "dq", "mpc_status", "mpc_active", "ik_status", "active_lower", "active_upper", "iters")


class SlabLayout(C.Structure):
    """wcqp_slab_layout: where the arrays of one rank's block of robots sit inside its input / output slab (include/wcqp.h)."""
    _fields_ = [("batch", C.c_int32), ("ref_len", C.c_int32), ("in_offset", C.c_int64 * len(SLAB_IN)), ("in_bytes", C.c_int64),
                ("out_offset", C.c_int64 * len(SLAB_OUT)), ("out_bytes", C.c_int64)]

    @classmethod
    def make(cls, batch, ref_len):
        L = cls()
        check(lib().wcqp_slab_layout_for(int(batch), int(ref_len), C.byref(L)), "wcqp_slab_layout_for")
        return L

    def in_shapes(self):
        """name -> (byte offset, numpy dtype, shape) of the input slab's arrays."""
        B, n = self.batch, self.ref_len
        shp = dict(x0=(B, 2), ref=(B, n, 2), u_prev=(B, 2), hull_A=(B, HULL_ROWS, 2), hull_b=(B, HULL_ROWS), hull_nc=(B,), J_left=(B, 6, 29),
                   J_right=(B, 6, 29), J_neck=(B, 3, 29), J_com=(B, 3, 29), q=(B, 23), state=(B, IK_STATE_LEN))
        return {k: (int(self.in_offset[i]), np.int32 if k == "hull_nc" else np.float64, shp[k]) for i, k in enumerate(SLAB_IN)}

    def out_shapes(self):
        B = self.batch
        f64 = dict(u0=(B, 2), mpc_margin=(B,), dq=(B, 23))
        return {k: (int(self.out_offset[i]), np.float64 if k in f64 else (np.int32 if k in ("mpc_status", "ik_status", "iters") else np.uint32), f64.get(k, (B,)))
                for i, k in enumerate(SLAB_OUT)}

    def step(self, in_slab: int, out_slab: int) -> "QpStep":
        """A step record whose pointers point INTO the two slabs (raw device addresses)."""
        r = QpStep()
        check(lib().wcqp_qp_step_from_slabs(C.byref(self), C.c_void_p(in_slab), C.c_void_p(out_slab), C.byref(r)), "wcqp_qp_step_from_slabs")
        return r


def qp_enqueue_steps(mpc, ik, batch, steps):
    """wcqp_qp_enqueue_steps: `steps` is a ctypes array of QpStep (build it once, replay it often); mpc / ik are the
    MpcSolver / IkSolver handles (either may be None when no record uses it)."""
    done = C.c_int32(0)
    check(lib().wcqp_qp_enqueue_steps(mpc._h if mpc is not None else None, ik._h if ik is not None else None, int(batch),
                                      len(steps), steps, C.byref(done)), "wcqp_qp_enqueue_steps")
    return done.value


PLAN_WAYS_AUTO = -1


class QpPlan:
    """wcqp_qp_plan_*: the records of qp_enqueue_steps uploaded once, replayed as ONE launch that walks through them; `ways`
    wavefronts share a robot group (way w takes records w, w + ways, ...: records of different ways need their own outputs);
    ways = 0: a work queue over (record, robot group) units - every record needs outputs of its own.
    Records without an IK part (J_left = None in all of them; ik may be None): an MPC-only plan; without an MPC part (x0 = None; mpc may
    be None): an IK-only plan."""

    def __init__(self, mpc, ik, batch, steps, ways=1):
        self._h = C.c_void_p()
        self._keep = (mpc, ik, steps)
        check(lib().wcqp_qp_plan_create(mpc._h if mpc is not None else None, ik._h if ik is not None else None, int(batch), len(steps), steps, int(ways), C.byref(self._h)), "wcqp_qp_plan_create")

    def enqueue(self, stream=0):
        check(lib().wcqp_qp_plan_enqueue(self._h, stream or None), "wcqp_qp_plan_enqueue")

    def close(self):
        if self._h:
            lib().wcqp_qp_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KinParams(C.Structure):
    _fields_ = [("dof", C.c_int32), ("parent", C.c_int32 * 32),
                ("R0", (C.c_double * 9) * 32), ("p0", (C.c_double * 3) * 32), ("axis", (C.c_double * 3) * 32),
                ("mass", C.c_double * 32), ("com", (C.c_double * 3) * 32),
                ("root_mass", C.c_double), ("root_com", C.c_double * 3),
                ("frame_joint", C.c_int32 * 3), ("frame_R", (C.c_double * 9) * 3), ("frame_p", (C.c_double * 3) * 3)]


class TickParams(C.Structure):
    _fields_ = [("batch", C.c_int32), ("first", C.c_int32), ("max_ticks", C.c_int32), ("log_ticks", C.c_int32),
                ("step_ticks", C.c_int32), ("ds_ticks", C.c_int32),
                ("k_com", C.c_double), ("k_zmp", C.c_double), ("noise", C.c_double), ("seed", C.c_uint64),
                ("mpc", MpcParams), ("ik", IkParams),
                ("ik_cold_start_only", C.c_int32), ("use_kinematics", C.c_int32), ("kin", KinParams), ("foot_rect", C.c_double * 8),
                ("kin_handoff", C.c_int32), ("ticks_per_launch", C.c_int32), ("logger_ticks", C.c_int32), ("plant", C.c_int32)]


class TickInputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ref_traj", "hull_tab_A", "hull_tab_b", "hull_tab_nc", "phase0",
                                          "J_left", "J_right", "J_neck", "J_com", "state0", "swing_twist",
                                          "q0", "dcm0", "com0", "u_init")]


class TickOutputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("u0_log", "dq_log", "q_des", "dcm", "com", "mpc_fail", "ik_fail", "hot_try", "hot_hit", "tick", "logger", "active_lower", "active_upper")]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WcqpError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                            "(make -C walking-controllers_amd/csrc); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        dp, ip, up, vp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.wcqp_strerror.restype = C.c_char_p
        L.wcqp_strerror.argtypes = [C.c_int]
        L.wcqp_stream_create.argtypes = [C.POINTER(C.c_void_p)]
        L.wcqp_stream_destroy.argtypes = [C.c_void_p]
        L.wcqp_stream_synchronize.argtypes = [C.c_void_p]
        L.wcqp_mpc_create.argtypes = [C.POINTER(MpcParams), C.POINTER(C.c_void_p)]
        L.wcqp_mpc_destroy.argtypes = [C.c_void_p]
        L.wcqp_mpc_get_condensed.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.wcqp_mpc_get_matrices.argtypes = [C.c_void_p, dp, dp, dp]
        mpc_args = [C.c_void_p, C.c_int32, dp, dp, C.c_int32, dp, dp, dp, ip, dp, ip, up, dp]
        L.wcqp_mpc_solve_device.argtypes = mpc_args + [vp]
        L.wcqp_mpc_solve_host.argtypes = mpc_args
        L.wcqp_ik_create.argtypes = [C.POINTER(IkParams), C.POINTER(C.c_void_p)]
        L.wcqp_ik_destroy.argtypes = [C.c_void_p]
        L.wcqp_ik_set_posture.argtypes = [C.c_void_p, C.c_void_p]
        ik_args = [C.c_void_p, C.c_int32, dp, dp, dp, dp, dp, dp, dp, ip, up, up, dp, ip]
        L.wcqp_ik_solve_device.argtypes = ik_args + [vp]
        L.wcqp_ik_solve_host.argtypes = ik_args
        L.wcqp_hull_from_feet_device.argtypes = [C.c_int32] + [C.c_void_p] * 8
        L.wcqp_hull_from_feet_host.argtypes = [C.c_int32] + [C.c_void_p] * 7
        L.wcqp_kin_create.argtypes = [C.POINTER(KinParams), C.POINTER(C.c_void_p)]
        L.wcqp_kin_destroy.argtypes = [C.c_void_p]
        L.wcqp_kin_jacobians_device.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8
        L.wcqp_kin_jacobians_host.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
        L.wcqp_tick_create.argtypes = [C.POINTER(TickParams), C.POINTER(C.c_void_p)]
        L.wcqp_tick_destroy.argtypes = [C.c_void_p]
        L.wcqp_tick_upload.argtypes = [C.c_void_p, C.POINTER(TickInputs)]
        L.wcqp_tick_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.wcqp_tick_download.argtypes = [C.c_void_p, C.POINTER(TickOutputs)]
        L.wcqp_tick_splice_reference.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.wcqp_tick_set_feedback_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wcqp_tick_set_feedback_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wcqp_qp_enqueue_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(QpStep), C.POINTER(C.c_int32)]
        L.wcqp_qp_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(QpStep), C.c_int32, C.POINTER(C.c_void_p)]
        L.wcqp_qp_plan_enqueue.argtypes = [C.c_void_p, C.c_void_p]
        L.wcqp_qp_plan_destroy.argtypes = [C.c_void_p]
        L.wcqp_slab_layout_for.argtypes = [C.c_int32, C.c_int32, C.POINTER(SlabLayout)]
        L.wcqp_qp_step_from_slabs.argtypes = [C.POINTER(SlabLayout), C.c_void_p, C.c_void_p, C.POINTER(QpStep)]
        _lib = L
    return _lib


def check(rc: int, what: str = "wcqp call") -> None:
    if rc != WCQP_OK:
        raise WcqpError(f"{what} failed: {lib().wcqp_strerror(rc).decode()} ({rc})")


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------------------
class MpcSolver:
    """Handle over wcqp_mpc_* — the batched stand-in for WalkingController + MPCSolver."""

    def __init__(self, horizon=50, sampling_time=0.01, com_height=0.53, gravity=9.81,
                 Q=None, R=None, convex_hull_tolerance=0.05, feas_tol=0.0):
        Q = 7500.0 * np.eye(2) if Q is None else np.asarray(Q, float)
        R = 9.0e6 * np.eye(2) if R is None else np.asarray(R, float)
        self.params = MpcParams(horizon, sampling_time, com_height, gravity,
                                (C.c_double * 4)(*Q.reshape(-1)), (C.c_double * 4)(*R.reshape(-1)),
                                convex_hull_tolerance, feas_tol)
        self.N = int(horizon)
        self._h = C.c_void_p()
        check(lib().wcqp_mpc_create(C.byref(self.params), C.byref(self._h)), "wcqp_mpc_create")

    def close(self):
        if self._h:
            lib().wcqp_mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def condensed(self):
        Gr = np.zeros((self.N + 1, 2, 2)); Gx = np.zeros((2, 2)); Gu = np.zeros((2, 2)); S0 = np.zeros((2, 2))
        check(lib().wcqp_mpc_get_condensed(self._h, _p(Gr), _p(Gx), _p(Gu), _p(S0)))
        return Gr, Gx, Gu, S0

    def matrices(self):
        n, nx, nu = 4 * self.N + 2, 2 * self.N + 2, 2 * self.N
        P = np.zeros((n, n)); A = np.zeros((nx, n)); G = np.zeros((nu, 2))
        check(lib().wcqp_mpc_get_matrices(self._h, _p(P), _p(A), _p(G)))
        return P, A, G

    def solve_host(self, x0, ref, u_prev, hull_A, hull_b, hull_nc):
        x0, ref, u_prev, hull_A, hull_b = map(_f64, (x0, ref, u_prev, hull_A, hull_b))
        hull_nc = np.ascontiguousarray(hull_nc, dtype=np.int32)
        B = x0.shape[0]
        ref_len = ref.shape[1]
        u0 = np.zeros((B, 2)); status = np.zeros(B, np.int32)
        active = np.zeros(B, np.uint32); margin = np.zeros(B)
        check(lib().wcqp_mpc_solve_host(self._h, B, _p(x0), _p(ref), ref_len, _p(u_prev), _p(hull_A), _p(hull_b),
                                        _p(hull_nc), _p(u0), _p(status), _p(active), _p(margin)),
              "wcqp_mpc_solve_host")
        return dict(u0=u0, status=status, active=active, margin=margin)

    def solve_device(self, batch, x0, ref, ref_len, u_prev, hull_A, hull_b, hull_nc,
                     u0, status, active=0, margin=0, stream=0):
        """All arguments are raw device addresses (ints), e.g. torch.Tensor.data_ptr()."""
        check(lib().wcqp_mpc_solve_device(self._h, batch, x0, ref, ref_len, u_prev, hull_A, hull_b, hull_nc,
                                          u0, status, active or None, margin or None, stream or None),
              "wcqp_mpc_solve_device")


class IkSolver:
    """Handle over wcqp_ik_* — the batched stand-in for WalkingQPIK_{osqp,qpOASES}."""

    def __init__(self, form=IK_FORM_QPOASES, dof=23, use_com_as_constraint=True,
                 com_weight=None, neck_weight=None, joint_reg_weights=None, joint_reg_gains=None,
                 joint_reg_rad=None, v_min=None, v_max=None,
                 k_pos_com=1.0, k_pos_foot=4.0, k_att_foot=2.0, k_neck=1.0,
                 rho=0.0, tol=0.0, max_iter=0, algorithm=0, jacobian_structure=0):
        from .synth import ICUB_JOINT_REG_DEG
        com_weight = 100.0 * np.eye(3) if com_weight is None else np.asarray(com_weight, float)
        neck_weight = 5.0 * np.eye(3) if neck_weight is None else np.asarray(neck_weight, float)
        if joint_reg_weights is None:
            joint_reg_weights = np.array([1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2] + [1] * 12, float)
        joint_reg_gains = 5.0 * np.ones(dof) if joint_reg_gains is None else joint_reg_gains
        joint_reg_rad = np.deg2rad(ICUB_JOINT_REG_DEG) if joint_reg_rad is None else joint_reg_rad
        v_max = np.ones(dof) if v_max is None else np.broadcast_to(np.asarray(v_max, float), (dof,))
        v_min = -v_max if v_min is None else np.broadcast_to(np.asarray(v_min, float), (dof,))

        def pad(a):
            out = np.zeros(MAX_DOF)
            out[:dof] = np.asarray(a, float)[:dof]
            return (C.c_double * MAX_DOF)(*out)

        self.params = IkParams(dof, int(bool(use_com_as_constraint)), int(form), int(max_iter),
                               (C.c_double * 9)(*com_weight.reshape(-1)), (C.c_double * 9)(*neck_weight.reshape(-1)),
                               pad(joint_reg_weights), pad(joint_reg_gains), pad(joint_reg_rad),
                               pad(v_min), pad(v_max), k_pos_com, k_pos_foot, k_att_foot, k_neck, rho, tol, int(algorithm),
                               int(jacobian_structure))
        self.dof = dof
        self._h = C.c_void_p()
        check(lib().wcqp_ik_create(C.byref(self.params), C.byref(self._h)), "wcqp_ik_create")

    def close(self):
        if self._h:
            lib().wcqp_ik_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_posture(self, joint_reg_rad):
        """WalkingQPIK::setDesiredJointPosition: a new regularisation posture (rad) for every later solve."""
        a = _f64(np.asarray(joint_reg_rad, float).reshape(self.dof))
        check(lib().wcqp_ik_set_posture(self._h, _p(a)), "wcqp_ik_set_posture")

    def solve_host(self, J_left, J_right, J_neck, J_com, q, state, want_foot_err=True):
        J_left, J_right, J_neck, J_com, q, state = map(_f64, (J_left, J_right, J_neck, J_com, q, state))
        B = q.shape[0]
        dq = np.zeros((B, self.dof)); status = np.zeros(B, np.int32)
        lo = np.zeros(B, np.uint32); up = np.zeros(B, np.uint32)
        ferr = np.zeros((B, 12)) if want_foot_err else None
        iters = np.zeros(B, np.int32)
        check(lib().wcqp_ik_solve_host(self._h, B, _p(J_left), _p(J_right), _p(J_neck), _p(J_com), _p(q), _p(state),
                                       _p(dq), _p(status), _p(lo), _p(up), _p(ferr), _p(iters)),
              "wcqp_ik_solve_host")
        return dict(dq=dq, status=status, active_lower=lo, active_upper=up, foot_err=ferr, iters=iters)

    def solve_device(self, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                     active_lower=0, active_upper=0, foot_err=0, iters=0, stream=0):
        """All arguments are raw device addresses (ints)."""
        check(lib().wcqp_ik_solve_device(self._h, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                         active_lower or None, active_upper or None, foot_err or None,
                                         iters or None, stream or None),
              "wcqp_ik_solve_device")


def hull_from_feet_host(foot_rect, left_T, right_T, contact):
    """Batch analogue of WalkingController::setConvexHullConstraint: foot poses -> hull rows."""
    foot_rect, left_T, right_T = _f64(foot_rect).reshape(8), _f64(left_T), _f64(right_T)
    contact = np.ascontiguousarray(contact, dtype=np.uint8)
    B = contact.shape[0]
    A = np.zeros((B, HULL_ROWS, 2)); b = np.zeros((B, HULL_ROWS)); nc = np.zeros(B, np.int32)
    check(lib().wcqp_hull_from_feet_host(B, _p(foot_rect), _p(left_T), _p(right_T), _p(contact), _p(A), _p(b), _p(nc)),
          "wcqp_hull_from_feet_host")
    return A, b, nc


class KinModel:
    """Handle over wcqp_kin_* - batched forward kinematics + MIXED free-floating Jacobians.
    `model` is a table as returned by `synth.icub_like_model()`."""

    def __init__(self, model: dict):
        n = int(model["dof"])
        p = KinParams()
        p.dof = n
        for j in range(n):
            p.parent[j] = int(model["parent"][j])
            for k in range(9):
                p.R0[j][k] = float(np.asarray(model["R0"][j]).reshape(9)[k])
            for k in range(3):
                p.p0[j][k] = float(model["p0"][j][k]); p.axis[j][k] = float(model["axis"][j][k]); p.com[j][k] = float(model["com"][j][k])
            p.mass[j] = float(model["mass"][j])
        p.root_mass = float(model["root_mass"])
        for k in range(3):
            p.root_com[k] = float(model["root_com"][k])
        for f in range(3):
            p.frame_joint[f] = int(model["frame_joint"][f])
            for k in range(9):
                p.frame_R[f][k] = float(np.asarray(model["frame_R"][f]).reshape(9)[k])
            for k in range(3):
                p.frame_p[f][k] = float(model["frame_p"][f][k])
        self.params, self.dof = p, n
        self._h = C.c_void_p()
        check(lib().wcqp_kin_create(C.byref(p), C.byref(self._h)), "wcqp_kin_create")

    def close(self):
        if self._h:
            lib().wcqp_kin_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def jacobians_host(self, base, q, state=None):
        base, q = _f64(base), _f64(q)
        B, nc = q.shape[0], 6 + self.dof
        JL = np.zeros((B, 6, nc)); JR = np.zeros((B, 6, nc)); JN = np.zeros((B, 3, nc)); JC = np.zeros((B, 3, nc))
        st = None if state is None else _f64(state).copy()
        check(lib().wcqp_kin_jacobians_host(self._h, B, _p(base), _p(q), _p(JL), _p(JR), _p(JN), _p(JC), _p(st)), "wcqp_kin_jacobians_host")
        return dict(J_left=JL, J_right=JR, J_neck=JN, J_com=JC, state=st)

    def jacobians_device(self, batch, base, q, J_left, J_right, J_neck, J_com, state=0, stream=0):
        check(lib().wcqp_kin_jacobians_device(self._h, int(batch), base, q, J_left, J_right, J_neck, J_com, state or None, stream or None),
              "wcqp_kin_jacobians_device")


class TickPipeline:
    """Handle over wcqp_tick_* — the device-resident MPC -> glue -> IK tick (configs 4/5)."""

    def __init__(self, batch, max_ticks, mpc: MpcSolver, ik: IkSolver, first=0, log_ticks=0,
                 step_ticks=180, ds_ticks=110, k_com=9.0, k_zmp=3.0, noise=1e-4, seed=99,
                 kin: "Optional[KinModel]" = None, foot_rect=None, ik_hot_start: bool = True, kin_handoff: int = 0,
                 ticks_per_launch: int = 0, logger_ticks: int = 0, external_feedback: bool = False):
        """kin: a KinModel -> per-tick kinematics (Jacobians, actual poses and hull rows rebuilt every tick from the
        integrated joint state with the base anchored at the stance foot; upload() then ignores J_* / hull_tab_*)."""
        self.batch, self.max_ticks, self.log_ticks, self.dof = batch, max_ticks, log_ticks, ik.dof
        self.logger_ticks = int(logger_ticks)
        self.use_kin = kin is not None
        if foot_rect is None:
            from .synth import FOOT_RECT
            foot_rect = FOOT_RECT
        self.params = TickParams(batch, first, max_ticks, log_ticks, step_ticks, ds_ticks, k_com, k_zmp, noise, seed,
                                 mpc.params, ik.params, int(not ik_hot_start), int(self.use_kin), kin.params if kin is not None else KinParams(),
                                 (C.c_double * 8)(*np.asarray(foot_rect, float).reshape(8)), int(kin_handoff), int(ticks_per_launch), int(logger_ticks),
                                 int(bool(external_feedback)))
        self._h = C.c_void_p()
        check(lib().wcqp_tick_create(C.byref(self.params), C.byref(self._h)), "wcqp_tick_create")
        self._keep = None

    def close(self):
        if self._h:
            lib().wcqp_tick_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, data: dict):
        f64 = ("ref_traj", "state0", "swing_twist", "q0", "dcm0", "com0", "u_init")
        f64 += () if self.use_kin else ("hull_tab_A", "hull_tab_b", "J_left", "J_right", "J_neck", "J_com")
        keep = {k: _f64(data[k]) for k in f64}
        if not self.use_kin:
            keep["hull_tab_nc"] = np.ascontiguousarray(data["hull_tab_nc"], dtype=np.int32)
        keep["phase0"] = np.ascontiguousarray(data["phase0"], dtype=np.int32)
        assert keep["ref_traj"].shape == (self.batch, self.max_ticks + self.params.mpc.horizon + 1, 2), keep["ref_traj"].shape
        ins = TickInputs(**{k: (keep[k].ctypes.data if k in keep else None) for k, _ in TickInputs._fields_})
        check(lib().wcqp_tick_upload(self._h, C.byref(ins)), "wcqp_tick_upload")

    def run(self, n_ticks: int, use_graph: bool = True, stream: int = 0):
        check(lib().wcqp_tick_run(self._h, int(n_ticks), int(bool(use_graph)), stream or None), "wcqp_tick_run")

    def set_feedback_device(self, dcm_meas: int, com_meas: int, zmp_meas: int, q_meas: int = 0, stream: int = 0):
        """External feedback (plant = EXTERNAL): raw DEVICE addresses of the measured DCM / CoM / ZMP [B][2] and, optionally, joint positions
        [B][dof] the next tick is to use; enqueue only."""
        check(lib().wcqp_tick_set_feedback_device(self._h, dcm_meas, com_meas, zmp_meas, q_meas or None, stream or None), "wcqp_tick_set_feedback_device")

    def set_feedback_host(self, dcm_meas, com_meas, zmp_meas, q_meas=None):
        """External feedback from host arrays ([B][2] each, q_meas [B][dof] or None): in place when the call returns (the run may name any stream)."""
        a = [_f64(x) for x in (dcm_meas, com_meas, zmp_meas)]
        q = None if q_meas is None else _f64(q_meas)
        assert all(x.shape == (self.batch, 2) for x in a) and (q is None or q.shape == (self.batch, self.dof))
        check(lib().wcqp_tick_set_feedback_host(self._h, _p(a[0]), _p(a[1]), _p(a[2]), _p(q)), "wcqp_tick_set_feedback_host")

    def splice_reference(self, from_tick: int, ref_tail, stream: int = 0):
        """Trajectory merge: stages [from_tick, from_tick + n) of every instance's DCM reference <- ref_tail[B][n][2]."""
        tail = _f64(ref_tail)
        assert tail.ndim == 3 and tail.shape[0] == self.batch and tail.shape[2] == 2, tail.shape
        # (the library stages the host rows before it returns: `tail` may go out of scope at once, whatever `stream` is still doing)
        check(lib().wcqp_tick_splice_reference(self._h, int(from_tick), tail.shape[1], _p(tail), stream or None), "wcqp_tick_splice_reference")

    def download(self):
        B, L, D = self.batch, self.log_ticks, self.dof
        o = dict(u0_log=np.zeros((L, B, 2)), dq_log=np.zeros((L, B, D)), q_des=np.zeros((B, D)), dcm=np.zeros((B, 2)),
                 com=np.zeros((B, 2)), mpc_fail=np.zeros(B, np.int64), ik_fail=np.zeros(B, np.int64),
                 hot_try=np.zeros(B, np.int64), hot_hit=np.zeros(B, np.int64), tick=np.zeros(1, np.int32),
                 active_lower=np.zeros(B, np.uint32), active_upper=np.zeros(B, np.uint32))
        if self.logger_ticks > 0:
            o["logger"] = np.zeros((self.logger_ticks, B, 53))
        outs = TickOutputs(**{k: (o[k].ctypes.data if k in o else None) for k, _ in TickOutputs._fields_})
        check(lib().wcqp_tick_download(self._h, C.byref(outs)), "wcqp_tick_download")
        o["tick"] = int(o["tick"][0])
        return o


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.cpp): what a measurement that is kept in the repository
    (profiles/traffic.json) is stamped with, so that it is not quoted for kernels that have changed since."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def device_count() -> int:
    return int(lib().wcqp_device_count())


def stream_create() -> int:
    """A HIP stream (raw handle as an int) from the library's own runtime - for callers without torch."""
    s = C.c_void_p()
    check(lib().wcqp_stream_create(C.byref(s)), "wcqp_stream_create")
    return s.value


def stream_destroy(stream: int) -> None:
    check(lib().wcqp_stream_destroy(C.c_void_p(stream)), "wcqp_stream_destroy")


def stream_synchronize(stream: int = 0) -> None:
    check(lib().wcqp_stream_synchronize(C.c_void_p(stream) if stream else None), "wcqp_stream_synchronize")
