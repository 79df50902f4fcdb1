"""
ORACLE (test infrastructure only; never imported by the product path).

Support-polygon rows from foot poses: what WalkingController::setConvexHullConstraint / buildConvexHull
(WM/src/WalkingDCMModelPredictiveController.cpp:364-489) ask of iDynTree's ConvexHullHelper for the one shape they ever
pass - the four corners of `foot_size` per foot in contact, through the foot's world transform, projected on the
ground plane, hulled.  iDynTree's row order / normalisation is upstream and unpinned, so the
convention is this repository's own: CCW hull, unit outward normals, rows a.u <= b, padded to 8 rows with 0.u <= 1e30.
PARITY UNPINNED by the reference; this file is the checker of csrc/hull_device.h (hull.hip, kin.hip in tick mode).
"""
from __future__ import annotations

import numpy as np

HULL_ROWS = 8
HULL_PAD_B = 1e30


def convex_hull_ccw(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain; CCW vertices without collinear points."""
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
    """rows (A[8,2], b[8], nc) of the convex hull of 2-D points; nc = 0 for fewer than 3 points."""
    A = np.zeros((HULL_ROWS, 2))
    b = np.full(HULL_ROWS, HULL_PAD_B)
    if len(pts) < 3:
        return A, b, 0
    V = convex_hull_ccw(pts)
    nc = len(V)
    for k in range(nc):
        v0, v1 = V[k], V[(k + 1) % nc]
        d = v1 - v0
        nrm = np.array([d[1], -d[0]]) / np.hypot(d[0], d[1])
        A[k] = nrm
        b[k] = nrm @ v0
    return A, b, nc


def foot_corners(pos_xy: np.ndarray, yaw: float, foot_x=(-0.02, 0.05), foot_y=(-0.025, 0.025)) -> np.ndarray:
    """Corners of the foot rectangle (foot_size, CFG/controllerParams.ini:7) of a foot at pos_xy with the given yaw."""
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s], [s, c]])
    loc = np.array([[foot_x[1], foot_y[1]], [foot_x[1], foot_y[0]], [foot_x[0], foot_y[0]], [foot_x[0], foot_y[1]]])
    return loc @ R.T + pos_xy


def foot_points(rect: np.ndarray, T: np.ndarray) -> np.ndarray:
    """Corners (x, y) x 4 of the foot rectangle through the foot-to-world transform T = [p(3), R(9 row-major)],
    projected on the XY plane."""
    rect = np.asarray(rect, float).reshape(4, 2)
    p, R = np.asarray(T[:3], float), np.asarray(T[3:12], float).reshape(3, 3)
    return np.stack([R[0, 0] * rect[:, 0] + R[0, 1] * rect[:, 1] + p[0], R[1, 0] * rect[:, 0] + R[1, 1] * rect[:, 1] + p[1]], -1)


def hull_from_feet(rect, left_T, right_T, contact: int):
    """contact: bit 0 = left foot in contact, bit 1 = right foot in contact (include/wcqp.h)."""
    pts = []
    if contact & 1:
        pts.append(foot_points(rect, left_T))
    if contact & 2:
        pts.append(foot_points(rect, right_T))
    return hull_rows(np.vstack(pts) if pts else np.zeros((0, 2)))
