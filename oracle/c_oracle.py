"""
ctypes loader for oracle/_build/libwc_oracle.so — TEST INFRASTRUCTURE ONLY
(checker in tests/, smoke(), and the cpu_baseline leg of bench.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libwc_oracle.so")


class OrcMpcParams(C.Structure):
    _fields_ = [("N", C.c_int), ("dT", C.c_double), ("com_height", C.c_double), ("gravity", C.c_double),
                ("Q", C.c_double * 4), ("R", C.c_double * 4)]


class OrcIkParams(C.Structure):
    _fields_ = [("dof", C.c_int), ("use_com", C.c_int), ("form", C.c_int),
                ("Wc", C.c_double * 9), ("Wn", C.c_double * 9),
                ("w", C.c_double * 32), ("gains", C.c_double * 32), ("qreg", C.c_double * 32),
                ("vmin", C.c_double * 32), ("vmax", C.c_double * 32),
                ("k_pos_com", C.c_double), ("k_pos_foot", C.c_double), ("k_att_foot", C.c_double), ("k_neck", C.c_double)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def osqp_dense(P, q, A, l, u, eps_abs=0.0, eps_rel=0.0, max_iter=0):
    """OSQP-algorithm restatement on a dense problem; returns (x, iters, status)."""
    P, q, A, l, u = (np.ascontiguousarray(v, dtype=np.float64) for v in (P, q, A, l, u))
    n, m = P.shape[0], A.shape[0]
    x = np.zeros(n)
    it = C.c_int(0)
    rc = lib().orc_osqp_dense(n, m, _p(P), _p(q), _p(A), _p(l), _p(u), C.c_double(eps_abs), C.c_double(eps_rel),
                              int(max_iter), _p(x), C.byref(it))
    return x, it.value, rc


def mpc_batch_osqp(mp, batch, nthreads=1):
    """mp: oracle.qp_spec.MPCParams; batch: dict in ABI layout.  Returns u0, iters, status."""
    p = OrcMpcParams(mp.horizon, mp.sampling_time, mp.com_height, mp.gravity,
                     (C.c_double * 4)(*np.asarray(mp.Q, float).reshape(-1)),
                     (C.c_double * 4)(*np.asarray(mp.R, float).reshape(-1)))
    B = batch["x0"].shape[0]
    ref = np.ascontiguousarray(batch["ref"], dtype=np.float64)
    u0 = np.zeros((B, 2)); iters = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
    arrs = [np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("x0", "u_prev", "hull_A", "hull_b")]
    nc = np.ascontiguousarray(batch["hull_nc"], dtype=np.int32)
    lib().orc_mpc_batch_osqp(C.byref(p), B, _p(arrs[0]), _p(ref), ref.shape[1], _p(arrs[1]), _p(arrs[2]), _p(arrs[3]),
                             _p(nc), _p(u0), _p(iters), _p(status), int(nthreads))
    return u0, iters, status


def mpc_batch_osqp_warm(mp, batch, warm_ticks, nthreads=1):
    """Warm ticks through one persistent OSQP workspace per instance (update bounds + gradient, warm-started solve):
    batch["ref"] must hold horizon + 1 + warm_ticks stages.  Returns (u0 of the last tick, mean iterations per warm
    solve, summed thread-seconds spent in the warm solves, number of warm solves that did not converge)."""
    p = OrcMpcParams(mp.horizon, mp.sampling_time, mp.com_height, mp.gravity,
                     (C.c_double * 4)(*np.asarray(mp.Q, float).reshape(-1)),
                     (C.c_double * 4)(*np.asarray(mp.R, float).reshape(-1)))
    B = batch["x0"].shape[0]
    ref = np.ascontiguousarray(batch["ref"], dtype=np.float64)
    u0 = np.zeros((B, 2))
    arrs = [np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("x0", "u_prev", "hull_A", "hull_b")]
    nc = np.ascontiguousarray(batch["hull_nc"], dtype=np.int32)
    mi, ws = C.c_double(0.0), C.c_double(0.0)
    f = lib().orc_mpc_batch_osqp_warm
    f.restype = C.c_int
    rc = f(C.byref(p), B, int(warm_ticks), _p(arrs[0]), _p(ref), ref.shape[1], _p(arrs[1]), _p(arrs[2]), _p(arrs[3]),
           _p(nc), _p(u0), C.byref(mi), C.byref(ws), int(nthreads))
    if rc < 0:
        raise ValueError("reference window too short for the requested warm ticks")
    return u0, mi.value, ws.value, rc


def _ik_params(ip, form):
    def pad(a):
        out = np.zeros(32)
        out[:len(a)] = a
        return (C.c_double * 32)(*out)
    return OrcIkParams(ip.dof, int(ip.use_com_as_constraint), 0 if form == "qpoases" else 1,
                       (C.c_double * 9)(*np.asarray(ip.com_weight, float).reshape(-1)),
                       (C.c_double * 9)(*np.asarray(ip.neck_weight, float).reshape(-1)),
                       pad(ip.joint_reg_weights), pad(ip.joint_reg_gains), pad(ip.q_reg),
                       pad(-np.asarray(ip.v_max)), pad(ip.v_max),
                       ip.k_pos_com, ip.k_pos_foot, ip.k_att_foot, ip.k_neck)


def ik_batch(ip, batch, form, nthreads=1):
    """ip: oracle.qp_spec.IKParams; form 'qpoases' | 'osqp'.  Returns dq, status, lo, up, iters."""
    p = _ik_params(ip, form)
    B = batch["q"].shape[0]
    arrs = [np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")]
    dq = np.zeros((B, ip.dof)); status = np.zeros(B, np.int32)
    lo = np.zeros(B, np.uint32); up = np.zeros(B, np.uint32); iters = np.zeros(B, np.int32)
    lib().orc_ik_batch(C.byref(p), B, *[_p(a) for a in arrs], _p(dq), _p(status), _p(lo), _p(up), _p(iters), int(nthreads))
    return dq, status, lo, up, iters


def mpc_condensed_gains(mp):
    """Rows of the inverse of the constant equality KKT K = [P A_eq'; A_eq 0] that give u0 (oracle/qp_spec.py's own P, A_eq;
    DESIGN.md 2): u0_unc = sum_i Gr_i r_i + Gx x0 + Gu u_prev, Sigma0 = the u0 block of K^-1.  numpy, once per parameter set;
    the cost is scaled by 1 / |R| first (the optimum does not depend on it; the scaled system is ~1e10 better conditioned)."""
    from . import qp_spec as qs
    c = qs.mpc_constants(mp)
    s = 1.0 / np.abs(np.asarray(mp.R, float)).max()
    K = np.block([[s * c.P, c.A_eq.T], [c.A_eq, np.zeros((c.n_x, c.n_x))]])
    Ki = np.linalg.inv(K)
    rows = Ki[c.n_x:c.n_x + 2]                           # u0 = rows @ [-s q; beq]
    # q_x[i] = -Q r_i, q_u[0:2] = -R u_prev (qs.mpc_gradient); beq[0:2] = -x0 (qs.mpc_assemble), the rest 0
    Gr = np.stack([rows[:, 2 * i:2 * i + 2] @ (s * c.Q) for i in range(c.N + 1)])
    Gu = rows[:, c.n_x:c.n_x + 2] @ (s * c.R)
    Gx = -rows[:, c.n:c.n + 2]
    S0 = s * Ki[c.n_x:c.n_x + 2, c.n_x:c.n_x + 2]
    return np.ascontiguousarray(Gr), np.ascontiguousarray(Gx), np.ascontiguousarray(Gu), np.ascontiguousarray(S0)


def mpc_batch_condensed(mp, batch, gains=None, nthreads=1, feas_tol=1e-10):
    """The device's MPC algorithm in plain C (condensed gains + 2-D projection by enumeration).  Returns u0, active, status."""
    Gr, Gx, Gu, S0 = gains if gains is not None else mpc_condensed_gains(mp)
    B = batch["x0"].shape[0]
    ref = np.ascontiguousarray(batch["ref"], dtype=np.float64)
    arrs = [np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("x0", "u_prev", "hull_A", "hull_b")]
    nc = np.ascontiguousarray(batch["hull_nc"], dtype=np.int32)
    u0 = np.zeros((B, 2)); act = np.zeros(B, np.uint32); status = np.zeros(B, np.int32)
    lib().orc_mpc_batch_condensed(int(mp.horizon), _p(Gr), _p(Gx), _p(Gu), _p(S0), C.c_double(feas_tol), B, _p(arrs[0]), _p(ref), ref.shape[1],
                                  _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(nc), _p(u0), _p(act), _p(status), int(nthreads))
    return u0, act, status


def ik_batch_range_space(ip, batch, form, nthreads=1):
    """The device's IK algorithm in plain C (base elimination + range space + dual active set).  Returns dq, status, lo, up, iters."""
    p = _ik_params(ip, form)
    B = batch["q"].shape[0]
    arrs = [np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")]
    dq = np.zeros((B, ip.dof)); status = np.zeros(B, np.int32)
    lo = np.zeros(B, np.uint32); up = np.zeros(B, np.uint32); iters = np.zeros(B, np.int32)
    rc = lib().orc_ik_batch_range_space(C.byref(p), B, *[_p(a) for a in arrs], _p(dq), _p(status), _p(lo), _p(up), _p(iters), int(nthreads))
    if rc < 0:
        raise ValueError("range-space IK: needs 23 DoF, CoM as constraint, a positive definite neck weight")
    return dq, status, lo, up, iters


def num_threads() -> int:
    return int(lib().orc_num_threads())
