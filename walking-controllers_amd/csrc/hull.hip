// Support-polygon rows from foot poses.  One thread per robot instance: the
// hull of at most 8 points is a few dozen flops, the kernel is a pure streaming pass
// (2 x 96 B in, 200 B out per instance).
//
// Reference: WalkingController::setConvexHullConstraint / buildConvexHull
//            src/WalkingDCMModelPredictiveController.cpp:364-489 (iDynTree ConvexHullHelpers upstream).
#include <cmath>
#include "wcqp_internal.h"
#include "hull_device.h"

namespace {

__global__ void hull_from_feet_kernel(int batch, const double* __restrict__ rect,
                                      const double* __restrict__ left_T, const double* __restrict__ right_T,
                                      const unsigned char* __restrict__ contact,
                                      double* __restrict__ hull_A, double* __restrict__ hull_b, int* __restrict__ hull_nc) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= batch) return;
    double px[8], py[8];
    int np = 0;
    const unsigned c = contact[inst];
    for (int f = 0; f < 2; ++f) {
        if (!((c >> f) & 1u)) continue;
        // foot-frame corner (x, y, 0) -> world, projected on the XY plane through the origin
        wcqp_hull::foot_points(rect, (f == 0 ? left_T : right_T) + (size_t)inst * 12, px, py, np);
    }
    hull_nc[inst] = wcqp_hull::hull_rows(px, py, np, hull_A + (size_t)inst * 16, hull_b + (size_t)inst * 8);
}

// The tick pipeline with per-tick kinematics: the three row sets (left foot, right foot, both in contact) of every robot
// from the DESIRED foot poses in its pose block (entries 24..35 left, 36..47 right).  One thread per (robot, contact pair).
struct RectArg { double v[8]; };
__global__ void hull_tables_kernel(int batch, RectArg rect, const double* __restrict__ state, int state_len,
                                   double* __restrict__ tab_A, double* __restrict__ tab_b, int* __restrict__ tab_nc) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= batch * 3) return;
    const int inst = g / 3, code = g % 3;
    const double* sd = state + (size_t)inst * state_len;
    double px[8], py[8];
    int np = 0;
    if (code == 0 || code == 2) wcqp_hull::foot_points(rect.v, sd + 24, px, py, np);
    if (code == 1 || code == 2) wcqp_hull::foot_points(rect.v, sd + 36, px, py, np);
    tab_nc[g] = wcqp_hull::hull_rows(px, py, np, tab_A + (size_t)g * 16, tab_b + (size_t)g * 8);
}

}  // namespace

namespace wcqp {
int hull_tables_from_state(int batch, const double* foot_rect_host, const double* state_dev, int state_len,
                           double* tab_A, double* tab_b, int* tab_nc, hipStream_t stream) {
    if (batch < 1 || !foot_rect_host || !state_dev || !tab_A || !tab_b || !tab_nc) return WCQP_E_INVALID;
    RectArg r;
    for (int k = 0; k < 8; ++k) r.v[k] = foot_rect_host[k];
    hipLaunchKernelGGL(hull_tables_kernel, dim3((batch * 3 + 127) / 128), dim3(128), 0, stream, batch, r, state_dev, state_len, tab_A, tab_b, tab_nc);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}
}  // namespace wcqp

extern "C" {

int wcqp_hull_from_feet_device(int32_t batch, const double* foot_rect, const double* left_T, const double* right_T,
                               const uint8_t* contact, double* hull_A, double* hull_b, int32_t* hull_nc, void* stream) {
    if (batch < 0 || !foot_rect || !left_T || !right_T || !contact || !hull_A || !hull_b || !hull_nc) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    hipLaunchKernelGGL(hull_from_feet_kernel, dim3((batch + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       batch, foot_rect, left_T, right_T, contact, hull_A, hull_b, hull_nc);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int wcqp_hull_from_feet_host(int32_t batch, const double* foot_rect, const double* left_T, const double* right_T,
                             const uint8_t* contact, double* hull_A, double* hull_b, int32_t* hull_nc) {
    if (batch < 0 || !foot_rect || !left_T || !right_T || !contact || !hull_A || !hull_b || !hull_nc) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the hull builder has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    const size_t B = (size_t)batch;
    double *d_rect = nullptr, *d_l = nullptr, *d_r = nullptr, *d_A = nullptr, *d_b = nullptr;
    unsigned char* d_c = nullptr; int* d_n = nullptr;
    int rc = WCQP_OK;
    if (hipMalloc(&d_rect, 64) != hipSuccess || hipMalloc(&d_l, B * 96) != hipSuccess || hipMalloc(&d_r, B * 96) != hipSuccess ||
        hipMalloc(&d_A, B * 128) != hipSuccess || hipMalloc(&d_b, B * 64) != hipSuccess || hipMalloc(&d_c, B) != hipSuccess ||
        hipMalloc(&d_n, B * 4) != hipSuccess) rc = WCQP_E_NOMEM;
    if (rc == WCQP_OK) {
        (void)hipMemcpy(d_rect, foot_rect, 64, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_l, left_T, B * 96, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_r, right_T, B * 96, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_c, contact, B, hipMemcpyHostToDevice);
        rc = wcqp_hull_from_feet_device(batch, d_rect, d_l, d_r, d_c, d_A, d_b, d_n, nullptr);
        if (rc == WCQP_OK && hipDeviceSynchronize() != hipSuccess) rc = WCQP_E_HIP;
        if (rc == WCQP_OK) {
            (void)hipMemcpy(hull_A, d_A, B * 128, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hull_b, d_b, B * 64, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hull_nc, d_n, B * 4, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_rect); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_A); (void)hipFree(d_b); (void)hipFree(d_c); (void)hipFree(d_n);
    return rc;
}

}  // extern "C"
