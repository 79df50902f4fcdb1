// Device side of the support-polygon builder, shared by hull.hip and the tick pipeline's
// kinematics kernel.  Internal, not ABI.
// Reference: WalkingController::setConvexHullConstraint / buildConvexHull
//            src/WalkingDCMModelPredictiveController.cpp:364-489 (iDynTree ConvexHullHelpers upstream).
#pragma once
#include "wcqp_internal.h"

namespace wcqp_hull {

#if defined(__HIPCC__)
// Corners of the foot rectangle `rect` (x, y) x 4 through the foot-to-world transform T (position 3, row-major
// rotation 9), projected on the XY plane; appended to (px, py).
__device__ __forceinline__ void foot_points(const double* rect, const double* T, double* px, double* py, int& np) {
    for (int k = 0; k < 4; ++k) {
        const double x = rect[2 * k], y = rect[2 * k + 1];
        px[np] = T[3] * x + T[4] * y + T[0];
        py[np] = T[6] * x + T[7] * y + T[1];
        ++np;
    }
}

// Rows A u <= b of the convex hull of np <= 8 points (CCW, unit outward normals, padded to 8 rows with
// 0.u <= 1e30); returns the row count (0 when fewer than 3 points).  One thread.
__device__ __forceinline__ int hull_rows(double* px, double* py, int np, double* A, double* b) {
    for (int k = 0; k < 8; ++k) { A[2 * k] = 0.0; A[2 * k + 1] = 0.0; b[k] = 1e30; }
    if (np < 3) return 0;
    // insertion sort by (x, y), then Andrew's monotone chain (collinear points dropped)
    for (int i = 1; i < np; ++i) {
        const double x = px[i], y = py[i];
        int j = i - 1;
        while (j >= 0 && (px[j] > x || (px[j] == x && py[j] > y))) { px[j + 1] = px[j]; py[j + 1] = py[j]; --j; }
        px[j + 1] = x; py[j + 1] = y;
    }
    double hx[16], hy[16];
    int k = 0;
    for (int i = 0; i < np; ++i) {                                  // lower hull
        while (k >= 2 && (hx[k - 1] - hx[k - 2]) * (py[i] - hy[k - 2]) - (hy[k - 1] - hy[k - 2]) * (px[i] - hx[k - 2]) <= 0) --k;
        hx[k] = px[i]; hy[k] = py[i]; ++k;
    }
    const int lower = k + 1;
    for (int i = np - 2; i >= 0; --i) {                             // upper hull
        while (k >= lower && (hx[k - 1] - hx[k - 2]) * (py[i] - hy[k - 2]) - (hy[k - 1] - hy[k - 2]) * (px[i] - hx[k - 2]) <= 0) --k;
        hx[k] = px[i]; hy[k] = py[i]; ++k;
    }
    const int nc = k - 1;                                           // last point == first point
    for (int e = 0; e < nc; ++e) {
        const double dx = hx[e + 1] - hx[e], dy = hy[e + 1] - hy[e];
        const double len = sqrt(dx * dx + dy * dy);
        const double ax = dy / len, ay = -dx / len;                 // outward normal of a CCW edge
        A[2 * e] = ax; A[2 * e + 1] = ay;
        b[e] = ax * hx[e] + ay * hy[e];
    }
    return nc;
}
#endif

}  // namespace wcqp_hull
