// Definitions shared by the two IK kernels (ik.hip: sweep on M = H + rho A'A;
// ik2.hip: null-space formulation).  Internal, not part of the ABI.
#pragma once
#include "wcqp_internal.h"

namespace wcqp_ik {

constexpr int kDof = 23;
constexpr int kNV = kDof + 6;          // 29
constexpr int kStateLen = WCQP_IK_STATE_LEN;

struct IkDeviceParams {
    double lam[32];        // Lambda diagonal per variable (0 on the base)       base.cpp:64-67
    double kq[32];         // w_i * K_i per variable                              base.cpp:70-72,87-89
    double qreg[32];       // regularisation posture per variable (rad)
    double vlo[32], vhi[32];   // variable bounds; base = -/+ DBL_MAX            qp.cpp:39-49
    double Wn[9], Wc[9];
    double k_pos_com, k_pos_foot, k_att_foot, k_neck, kappa, rho, tol;
    int form, max_iter;
};

// rotation error component k of unskew(0.5 (E - E')), E = R Rd'     Utils.cpp:22-27
__device__ __forceinline__ double rot_err(const double* R, const double* Rd, int k) {
    const int a = (k + 2) % 3, b = (k + 1) % 3;
    const double eab = R[3 * a] * Rd[3 * b] + R[3 * a + 1] * Rd[3 * b + 1] + R[3 * a + 2] * Rd[3 * b + 2];
    const double eba = R[3 * b] * Rd[3 * a] + R[3 * b + 1] * Rd[3 * a + 1] + R[3 * b + 2] * Rd[3 * a + 2];
    return 0.5 * (eab - eba);
}

// Broadcast of lane `SRC` of each 32-lane group to every lane of that group: ds_swizzle in
// bit-mask mode (and = 0, or = SRC, xor = 0) moves data through the LDS crossbar without
// touching LDS memory, so no write -> read round trip is needed.
template <int SRC>
__device__ __forceinline__ double group_bcast(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (SRC & 31) << 5);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (SRC & 31) << 5);
    return __hiloint2double(hi, lo);
}


// launch of the null-space kernel (ik2.hip)
int ik2_launch(const IkDeviceParams* d_prm, bool use_com, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream);

}  // namespace wcqp_ik
